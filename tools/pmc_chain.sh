#!/bin/bash
# SQ counters of the segment path's kernels on the C5/8 shard (separate passes, --pmc alone with --kernel-trace).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcc_a $R/gpurun_out/pmcc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_a -o a -- python3 $R/tools/chain_prof.py > $R/gpurun_out/pmcc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_b -o b -- python3 $R/tools/chain_prof.py > $R/gpurun_out/pmcc_b.log 2>&1
for d in a b; do python3 $R/tools/pmc_mean.py $R/gpurun_out/pmcc_$d chain_; done > $R/gpurun_out/pmc_chain_summary.txt
cat $R/gpurun_out/pmc_chain_summary.txt
