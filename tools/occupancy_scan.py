"""Sweep kernel time against the number of workgroups per CU (batch size), C2-shaped lattices."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for U in (16, 1024, 2048, 4096, 6144, 8192, 10000, 10240, 12288, 16384, 20480):
    ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
    a = klat_amd.synth(20260001, ns, fo)
    b = klat_amd.Batch(ctx, a, lanes_per_lattice=G, ring_slots=W)
    info = b.info()
    ts = []
    for it in range(80):
        b.forward_backward(1.0, 0.1, 0.0)
        if it >= 40:
            ctx.sync(); ts.append(b.timings()["sweep_ms"])
    t = float(np.median(ts))
    print("U %6d  jobs %5d (%.2f per CU)  rounds %d  sweep %.4f ms  -> %.0f cycles/round @2.33GHz, %.2f G arcs/s" % (
        U, info["num_jobs"], info["num_jobs"] / 256.0, info["max_rounds"], t, t * 1e-3 * 2.33e9 / info["max_rounds"], b.A / t / 1e6), flush=True)
    del b
