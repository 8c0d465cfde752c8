#!/bin/bash
# SQ counters of the stream packer's kernels on the C2 batch (separate passes; --pmc alone with
# --kernel-trace, as gpurun requires).  tools/pmc_mean.py condenses the output.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcp_a $R/gpurun_out/pmcp_b $R/gpurun_out/pmcp_c $R/gpurun_out/pmcp_d
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_a -o a -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > $R/gpurun_out/pmcp_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_b -o b -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > $R/gpurun_out/pmcp_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_c -o c -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > $R/gpurun_out/pmcp_c.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_d -o d -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > $R/gpurun_out/pmcp_d.log 2>&1
for d in a b c d; do python3 $R/tools/pmc_mean.py $R/gpurun_out/pmcp_$d pack_; done > $R/gpurun_out/pmc_pack_summary.txt
cat $R/gpurun_out/pmc_pack_summary.txt
