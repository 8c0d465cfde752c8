"""BASELINE configs[3] whole (100 000 lattices x 5 000 states, 1.5e9 arcs) on ONE MI355X: does it fit the
288 GB, and at what rate does it run."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = klat_amd.Ctx(0)
t0 = time.perf_counter()
a = klat_amd.synth(20260003, np.full(U, 5000, np.int32), np.full(U, 3, np.int32), host_threads=16)
print("synth: %.1f s, %.1f GB of host arrays" % (time.perf_counter() - t0, sum(v.nbytes for v in a.values()) / 1e9), flush=True)
t0 = time.perf_counter()
b = klat_amd.Batch(ctx, a, host_threads=16)
ctx.sync()
print("klat_batch_create: %.2f s; info %s" % (time.perf_counter() - t0, b.info()), flush=True)
for _ in range(3):
    b.forward_backward(1.0, 0.1, 0.0)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    b.forward_backward(1.0, 0.1, 0.0)
ctx.sync()
dt = (time.perf_counter() - t0) / 5
h = ctx.timing_history(5)
total, tf, tb = b.totals()
ok = bool(np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-6)
print("step %.2f ms (sweep %.2f, posterior %.2f) -> %.1f G arcs/s; forward and backward totals agree: %s" % (
    dt * 1e3, np.median([t["sweep_ms"] for t in h]), np.median([t["posterior_ms"] for t in h]), b.A / dt / 1e9, ok), flush=True)
