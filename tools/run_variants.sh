#!/bin/bash
# kernel experiments: bench each variant library under build/variants (timings only, results may be wrong)
for f in kaldi-lattice-utils_amd/build/variants/libklat_*.so; do
  KLAT_EXPERIMENT=1 KLAT_LIB_OVERRIDE=$PWD/$f python bench.py --steps 200 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['kernel_ms'])"
done
