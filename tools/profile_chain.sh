#!/bin/bash
# Profiles the segment path (klat_chain.hip) on the C5/8 shard (tools/chain_prof.py: 10 sweeps of the BASELINE
# configs[4] mix) for the files kept under profiles/: per-kernel times, and FETCH_SIZE / WRITE_SIZE in separate
# passes.  tools/prof_summarize.py <tag>_chain chainprof condenses the output afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/chainprof_stats $R/gpurun_out/chainprof_fetch $R/gpurun_out/chainprof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chainprof_stats -o stats -- python3 $R/tools/chain_prof.py > $R/gpurun_out/chainprof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/chainprof_fetch -o fetch -- python3 $R/tools/chain_prof.py > $R/gpurun_out/chainprof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/chainprof_write -o write -- python3 $R/tools/chain_prof.py > $R/gpurun_out/chainprof_write.log 2>&1
find $R/gpurun_out/chainprof_stats $R/gpurun_out/chainprof_fetch $R/gpurun_out/chainprof_write -name "*.csv" | head
