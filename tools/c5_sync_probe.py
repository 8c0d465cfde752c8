"""C5/8 (skewed mix, four layout classes on four streams): sweep time with a host sync after every
step against steps enqueued back to back."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
if os.environ.get("PROBE_TORCH"):
    import torch
    _x = torch.zeros(1024, device="cuda:0"); torch.cuda.synchronize()
ctx = klat_amd.Ctx(0)
rng = np.random.RandomState(20260004)
ns = np.exp(rng.uniform(np.log(200), np.log(20000), 2500)).astype(np.int32)
fo = rng.choice([2, 3, 4, 6], 2500).astype(np.int32)
a = klat_amd.synth(20260004, ns, fo, host_threads=16)
b = klat_amd.Batch(ctx, a, host_threads=16)
print(b.info(), flush=True)
for mode in ("sync", "back-to-back", "sync", "back-to-back"):
    for _ in range(10):
        b.forward_backward(1.0, 0.1, 0.0)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        b.forward_backward(1.0, 0.1, 0.0)
        if mode == "sync":
            ctx.sync()
    ctx.sync()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    h = ctx.timing_history(20)
    print("%-13s wall %.3f ms/step   sweep_ms median %.3f  min %.3f  max %.3f   total_ms median %.3f" % (
        mode, wall, np.median([t["sweep_ms"] for t in h]), min(t["sweep_ms"] for t in h), max(t["sweep_ms"] for t in h),
        np.median([t["total_ms"] for t in h])), flush=True)
