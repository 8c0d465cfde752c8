#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
export KLAT_CHAIN_DEBUG=$R/gpurun_out/chain_waves.bin
python3 $R/tools/chain_prof.py > $R/gpurun_out/cexp.log 2>&1
python3 $R/tools/chain_waves.py $R/gpurun_out/chain_waves.bin | grep -A3 sweep
unset KLAT_CHAIN_DEBUG
python3 $R/tools/chain_time2.py 1
