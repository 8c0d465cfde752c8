#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cexp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cexp -o cp -- python3 $R/tools/chain_prof.py > $R/gpurun_out/cexp.log 2>&1
python3 $R/tools/prof_db.py $R/gpurun_out/cexp/cp_results.db | grep -i "chain_\|sweep_kernel" | head -10 | cut -c1-160
python3 $R/tools/chain_time2.py 1
