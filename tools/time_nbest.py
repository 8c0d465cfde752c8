"""Times klat_nbest (k = 100) on C2-shaped lattices: the n-best leg of BASELINE configs[2]."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 6250   # configs[2] / 8 GPUs
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
a = klat_amd.synth(20260002, ns, fo, host_threads=16)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a)
rng = np.random.RandomState(0)
cost = rng.rand(b.A).astype(np.float32)
cost[a["olabel"] == 0] = 0.0
for it in range(2):
    t0 = time.perf_counter()
    npaths, pc, pl, lab = b.nbest(cost, k, 2048)
    dt = time.perf_counter() - t0
    print("klat_nbest k=%d on %d lattices x 2000 states (%d arcs): %.1f ms incl. upload of costs and D2H of %.0f MB of paths -> %.2f G arcs/s, %.1f k lattices/s"
          % (k, U, b.A, 1e3 * dt, (pc.nbytes + pl.nbytes + lab.nbytes) / 1e6, b.A / dt / 1e9, U / dt / 1e3), flush=True)
assert np.all(npaths == k) and np.all(np.diff(pc, axis=1) >= 0)
