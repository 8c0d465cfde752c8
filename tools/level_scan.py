"""Sweep time of WIDE lattices (2000 states in levels of 32: depth 64) through the level path and
through the streams, against the number of lattices in the batch."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
for U in (1, 16, 64, 256, 1024, 4096):
    ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
    a = klat_amd.synth(20260005, ns, fo, wide=32)
    row = []
    for G in (0, 4):
        b = klat_amd.Batch(ctx, a, lanes_per_lattice=G)
        for _ in range(30): b.forward_backward(1.0, 0.1, 0.0)
        h = ctx.timing_history(20)
        row.append((b.info()["level_lattices"], float(np.median([t["sweep_ms"] for t in h]))))
        b.close()
    print("U=%5d  level path (%d lattices): %.3f ms   streams: %.3f ms" % (U, row[0][0], row[0][1], row[1][1]), flush=True)
