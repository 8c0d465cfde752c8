"""Same-box A/B of two builds of the library on the C2 sweep: alternates processes is not possible
inside one interpreter (the library is loaded once), so this script measures ONE build (the one
KLAT_LIB_OVERRIDE names, else the tree's) and tools/ab_sweep.sh alternates them on one box."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
a = klat_amd.synth(20260001, np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32), host_threads=16)
b = klat_amd.Batch(ctx, a, host_threads=16, fuse_scales=(1.0, 0.1, 0.0) if os.environ.get("KLAT_AB_FUSE", "1") == "1" else None)
for _ in range(300):
    b.forward_backward(1.0, 0.1, 0.0)
ctx.sync()
h = ctx.timing_history(200)
print("%s: sweep median %.4f ms  min %.4f   posterior median %.4f ms   step median %.4f ms" % (os.environ.get("KLAT_LIB_OVERRIDE", "tree")[-20:],
      np.median([t["sweep_ms"] for t in h]), min(t["sweep_ms"] for t in h), np.median([t["posterior_ms"] for t in h]),
      np.median([t["total_ms"] for t in h])), flush=True)
