"""Sweep time of the stream layouts (lanes per lattice G, ring slots W) on a very large batch of small
lattices: which layout the automatic choice should take there."""
import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
U, S = 140000, 150
a = klat_amd.synth(5, np.full(U, S, np.int32), np.full(U, 3, np.int32))
for G, W in ((0, 0), (1, 16), (1, 32), (2, 32), (4, 32)):
    b = klat_amd.Batch(ctx, a, lanes_per_lattice=G, ring_slots=W)
    info = b.info()
    ts = []
    for it in range(40):
        b.forward_backward(1.0, 0.1, 0.0)
        if it >= 20:
            ctx.sync(); ts.append(b.timings()["sweep_ms"])
    t = float(np.median(ts))
    print("G %d W %d -> G %d W %d jobs %d rounds %d: sweep %.3f ms, %.1f G arcs/s (sweep only)" % (G, W, info["lanes_per_lattice"], info["ring_slots"], info["num_jobs"], info["max_rounds"], t, b.A / t / 1e6), flush=True)
    b.close()
