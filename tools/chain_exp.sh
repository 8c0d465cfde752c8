#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
export KLAT_PACK_PROFILE=1
python3 $R/tools/chain_prof.py 2>&1 | grep -i "host stages\|pack_count phases" | tail -3
