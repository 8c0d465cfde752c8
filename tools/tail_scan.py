"""Does a batch whose jobs just exceed the resident capacity (6 workgroups per CU x 256 CUs = 1536
at W = 32) pay a second wave of workgroups?  Sweep time against the number of 5000-state lattices."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for U in (11776, 12288, 12304, 12500, 13312, 14336):
    ns = np.full(U, S, np.int32); fo = np.full(U, 3, np.int32)
    a = klat_amd.synth(20260003, ns, fo, host_threads=16)
    b = klat_amd.Batch(ctx, a, host_threads=16)
    for _ in range(25): b.forward_backward(1.0, 0.1, 0.0)
    h = ctx.timing_history(15)
    sw = float(np.median([t["sweep_ms"] for t in h]))
    info = b.info()
    print("U=%d jobs=%d (%.2f/CU) sweep %.3f ms  %.1f ns per lattice-state" % (U, info["num_jobs"], info["num_jobs"] / 256.0, sw, sw * 1e6 / (U * S)), flush=True)
    b.close(); del a
