"""Runs the C2 batch against the cycle-counter variant of the sweep (tools/make_prof_variant.py,
KLAT_LIB_OVERRIDE=.../libklat_prof.so) and prints where producer and consumer waves spend time."""
import os, sys
import numpy as np
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd, ctypes as C
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
a = klat_amd.synth(20260001, ns, fo)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a, lanes_per_lattice=4, ring_slots=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
for _ in range(60):
    b.forward_backward(1.0, 0.1, 0.0)
ctx.sync()
print("U", U, "timings", b.timings())
n = min(4096, b.info()["num_jobs"])
buf = np.zeros(n * 8, np.uint64)
klat_amd.lib().klat_debug_sweep_profile(klat_amd._p(buf), C.c_int(n * 8))
p = buf.reshape(n, 8).astype(np.float64)
rounds = p[:, 4] * 4
for name, col in (("consumer total", 0), ("  barrier", 1), ("  LDS wait at chunk start", 2), ("  four rounds", 3)):
    x = p[:, col]
    print("%-26s mean %9.0f cyc  (%.1f per round)  min %9.0f max %9.0f" % (name, x.mean(), (x / rounds).mean(), x.min(), x.max()))
print("rounds per job: mean %.0f" % rounds.mean())
