#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cexpb
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cexpb -o cp -- python3 $R/bench.py --no-cpu-baseline --no-strong --no-one-shot --no-accumulate > $R/gpurun_out/cexpb.json 2> $R/gpurun_out/cexpb.err
python3 $R/tools/prof_timeline.py $R/gpurun_out/cexpb/cp_results.db -3
