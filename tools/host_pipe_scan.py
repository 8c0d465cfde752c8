"""klat_forward_backward_host on the C2 batch: seconds per call against the number of pieces (page-locked buffers)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260001, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=16)
compact = klat_amd.compact_arrays(arrays)
pin = {k: (klat_amd.pinned_copy(v) if k not in ("state_off", "arc_off") else v) for k, v in compact.items()}
St, A = int(arrays["state_off"][-1]), int(arrays["arc_off"][-1])
out = {"alpha": klat_amd.pinned_empty(St, np.float64), "beta": klat_amd.pinned_empty(St, np.float64),
       "total": klat_amd.pinned_empty(U, np.float64), "post": klat_amd.pinned_empty(A, np.float64)}
scales = (1.0, 0.1, 0.0)
for fuse in (True, False):
    for pieces in (1, 2, 3, 4, 6, 8, 12, 16, 0):
        t = []
        for _ in range(4):
            t0 = time.perf_counter()
            klat_amd.forward_backward_host(ctx, pin, scales, pieces=pieces, fuse=fuse, out=out, host_threads=16)
            t.append(time.perf_counter() - t0)
        print("fuse=%d pieces=%2d: %.1f ms (min), %.1f (median of the last three)" % (fuse, pieces, min(t) * 1e3, sorted(t[1:])[1] * 1e3), flush=True)
