#!/bin/bash
# SQ counters of the sweep kernel on the C2 batch (two passes; --pmc alone with --kernel-trace, as
# gpurun requires).  tools/pmc_mean.py condenses the output.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_a -o a -- python3 $R/bench.py --steps 10 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-shapes --no-accumulate > $R/gpurun_out/pmc_a.json 2> $R/gpurun_out/pmc_a.err
echo "pass a done"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_b -o b -- python3 $R/bench.py --steps 10 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-shapes --no-accumulate > $R/gpurun_out/pmc_b.json 2> $R/gpurun_out/pmc_b.err
echo "pass b done"
