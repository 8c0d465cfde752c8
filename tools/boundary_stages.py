"""Per-stage wall times of the host-to-host path on one context: create (H2D + packing), sweep, fetch."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U, S = 10000, 2000
ns = np.full(U, S, np.int32); fo = np.full(U, 3, np.int32)
h = klat_amd.synth(20260001, ns, fo, host_threads=16)
for pinned in (False, True):
    hh = {k: klat_amd.pinned_copy(v) for k, v in h.items()} if pinned else h
    alloc = klat_amd.pinned_empty if pinned else (lambda shape, dt: np.empty(shape, dt))
    A, St = int(h["arc_off"][-1]), int(h["state_off"][-1])
    post, al, be = alloc(A, np.float64), alloc(St, np.float64), alloc(St, np.float64)
    ctx = klat_amd.Ctx(0)
    for it in range(4):
        t0 = time.perf_counter(); b = klat_amd.Batch(ctx, hh, host_threads=8); t1 = time.perf_counter()
        b.forward_backward(1.0, 0.1, 0.0); ctx.sync(); t2 = time.perf_counter()
        b.arc_logpost(out=post); t3 = time.perf_counter(); b.alpha_beta(out=(al, be)); t4 = time.perf_counter()
        b.close(); t5 = time.perf_counter()
        print("%s it %d: create %.1f ms  sweep %.1f  logpost D2H %.1f (%.1f GB/s)  alpha/beta D2H %.1f  destroy %.1f" % (
            "pinned  " if pinned else "pageable", it, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), post.nbytes/(t3-t2)/1e9, 1e3*(t4-t3), 1e3*(t5-t4)), flush=True)
    ctx.close()
