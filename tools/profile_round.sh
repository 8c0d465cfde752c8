#!/bin/bash
# Profiles bench.py on the GPU box for the files kept under profiles/ (run through gpurun from the
# repo root): per-kernel times (--kernel-trace --stats) and, in separate passes, the PMC counters
# FETCH_SIZE and WRITE_SIZE.  tools/prof_summarize.py <tag> condenses the output afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-shapes --no-strong --no-boundary > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -o fetch -- python3 $R/bench.py --steps 20 --warmup 5 --settle-seconds 0 --no-cpu-baseline --no-shapes --no-strong --no-boundary > $R/gpurun_out/bench_pmc_fetch.json 2> $R/gpurun_out/bench_pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -o write -- python3 $R/bench.py --steps 20 --warmup 5 --settle-seconds 0 --no-cpu-baseline --no-shapes --no-strong --no-boundary > $R/gpurun_out/bench_pmc_write.json 2> $R/gpurun_out/bench_pmc_write.err
echo "write pass done"
find $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write -name "*.csv" | head -20
