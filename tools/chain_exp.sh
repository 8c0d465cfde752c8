#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
for seg in 128; do
export KLAT_CHAIN_SEG=$seg
export KLAT_CHAIN_DEBUG=$R/gpurun_out/chain_waves.bin
echo "== seg $seg"
python3 $R/tools/chain_prof.py > $R/gpurun_out/cexp.log 2>&1
python3 $R/tools/chain_waves.py $R/gpurun_out/chain_waves.bin
done
