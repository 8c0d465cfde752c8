#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the lean fill pass per direction (KLAT_FILL_PAIR=0: a launch per direction) on the C2 batch.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcf_r $R/gpurun_out/pmcf_w
KLAT_FILL_PAIR=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_r -o r -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > /dev/null 2>&1
KLAT_FILL_PAIR=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_w -o w -- python3 $R/tools/pack_time.py 10000 2000 3 3 1 1 > /dev/null 2>&1
PMC_NAME_CHARS=90 python3 $R/tools/pmc_mean.py $R/gpurun_out/pmcf_r pack_fill
PMC_NAME_CHARS=90 python3 $R/tools/pmc_mean.py $R/gpurun_out/pmcf_w pack_fill
