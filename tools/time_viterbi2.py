"""Wall time per call of the tropical sweep (klat_viterbi) on the C2 batch, 200 calls back to back."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
a = klat_amd.synth(20260001, np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32), host_threads=16)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a, host_threads=16)
for name, f in (("viterbi", lambda: b.viterbi(1.0, 0.1, 0.0)), ("forward_backward", lambda: b.forward_backward(1.0, 0.1, 0.0))):
    for _ in range(100): f()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(200): f()
    ctx.sync()
    print("%s: %.3f ms per call" % (name, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
print("sweep_ms of forward_backward (events):", np.median([t["sweep_ms"] for t in ctx.timing_history(50)]))
