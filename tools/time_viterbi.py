"""Times the tropical twin (klat_viterbi + klat_prune_mask) on the C2 batch."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
a = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a)
for it in range(5):
    ctx.sync(); t0 = time.perf_counter()
    b.viterbi(1.0, 0.1, 0.0); ctx.sync()
    t1 = time.perf_counter()
    ka, kf, lb = b.prune_mask(5.0)
    t2 = time.perf_counter()
    print("viterbi sweep %.3f ms (%.1f G arcs/s); prune mask incl. D2H of %.0f MB: %.2f ms" % (1e3 * (t1 - t0), b.A / (t1 - t0) / 1e9, (ka.nbytes + kf.nbytes) / 1e6, 1e3 * (t2 - t1)), flush=True)
