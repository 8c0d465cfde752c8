import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, ROOT)
import klat_amd
from bench import c5_sizes
ns, fo = c5_sizes(2500)
arrays = klat_amd.synth(20260004, ns, fo, host_threads=16)
ctx = klat_amd.Ctx(0)
scales = (1.0, 0.1, 0.0)
b = klat_amd.Batch(ctx, arrays, host_threads=16, fuse_scales=scales)
for _ in range(10):
    b.forward_backward(*scales)
ctx.sync()
print(b.info())
