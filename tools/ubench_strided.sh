#!/bin/bash
# tools/ubench_strided.hip over its modes, then WRITE_SIZE / FETCH_SIZE of the interesting ones (separate passes).
R=${GRAFT_REPO_ROOT:-$PWD}
B=$R/tools/ubench_strided.bin
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2 3 4; do for xcd in 0 1; do for jit in 0 8; do
  if [ $mode = 0 ] && [ $xcd = 1 ]; then continue; fi
  $B $mode $xcd $jit 4 | tail -2
done; done; done
for cfg in "0 0 0" "1 1 0" "1 1 8" "1 0 8" "2 1 8"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $R/gpurun_out/ubs_w_$tag $R/gpurun_out/ubs_f_$tag
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ubs_w_$tag -o w -- $B $cfg 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ubs_f_$tag -o f -- $B $cfg 3 > /dev/null 2>&1
  echo "cfg $cfg:"; python3 $R/tools/pmc_mean.py $R/gpurun_out/ubs_w_$tag _major; python3 $R/tools/pmc_mean.py $R/gpurun_out/ubs_f_$tag _major
done
