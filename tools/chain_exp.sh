#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for cfg in 0 1 2 6 7; do
  export KLAT_CHAIN_EXP=$cfg
  rm -rf $R/gpurun_out/cexp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cexp -o cp -- python3 $R/tools/chain_prof.py > $R/gpurun_out/cexp.log 2>&1
  echo "== exp=$cfg"
  python3 $R/tools/prof_db.py $R/gpurun_out/cexp/cp_results.db | grep -i "chain_walk\|chain_comb"
done
