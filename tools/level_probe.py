"""Wall time of forward_backward and viterbi for a few WIDE lattices: level path against streams."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
for U, W in ((1, 32), (1, 64), (1, 200), (16, 32)):
    a = klat_amd.synth(20260005, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), wide=W)
    for G in (0, 4):
        b = klat_amd.Batch(ctx, a, lanes_per_lattice=G)
        out = []
        for f in (lambda: b.forward_backward(1.0, 0.1, 0.0), lambda: b.viterbi(1.0, 0.1, 0.0)):
            for _ in range(20): f()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(200): f()
            ctx.sync(); out.append((time.perf_counter() - t0) / 200 * 1e3)
        print("U=%d width=%d %s: forward_backward %.3f ms  viterbi %.3f ms" % (U, W, "level  " if b.info()["level_lattices"] else "streams", out[0], out[1]), flush=True)
        b.close()
