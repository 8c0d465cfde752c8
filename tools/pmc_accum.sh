#!/bin/bash
# SQ counters of the accumulate kernels on the C2 batch (separate passes, --pmc alone with --kernel-trace).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmca_a $R/gpurun_out/pmca_b
ARGS="--steps 5 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-shapes --no-strong --no-boundary --no-one-shot"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/pmca_a -o a -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmca_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmca_b -o b -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmca_b.log 2>&1
for d in a b; do python3 $R/tools/pmc_mean.py $R/gpurun_out/pmca_$d index_hash; python3 $R/tools/pmc_mean.py $R/gpurun_out/pmca_$d frame_lean; python3 $R/tools/pmc_mean.py $R/gpurun_out/pmca_$d frame_kernel; done > $R/gpurun_out/pmc_accum_summary.txt
cat $R/gpurun_out/pmc_accum_summary.txt
