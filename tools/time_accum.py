"""Times the accumulate stage (segment index, frame posteriors) on the C2 workload."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd, ctypes as C
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ns = np.full(U, S, np.int32); fo = np.full(U, 3, np.int32)
a = klat_amd.synth(20260001, ns, fo)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a)
b.forward_backward(1.0, 0.1, 0.0); ctx.sync()
f, keep = klat_amd.Batch._filter(None, None)
off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
for it in range(3):
    t0 = time.perf_counter()
    klat_amd._check(klat_amd.lib().klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
    t1 = time.perf_counter()
    print("segment index (compute+count, incl. small D2H): %.2f ms, %d records, %.2f G arcs/s" % (1e3 * (t1 - t0), n.value, b.A / (t1 - t0) / 1e9))
rec = np.zeros(n.value, klat_amd.SEGMENT_DTYPE)
t0 = time.perf_counter()
klat_amd._check(klat_amd.lib().klat_segment_index(b.h, C.byref(f), klat_amd._p(off), klat_amd._p(rec), C.c_int64(len(rec)), C.byref(n)))
print("gather + D2H of %.1f MB: %.2f ms" % (rec.nbytes / 1e6, 1e3 * (time.perf_counter() - t0)))
if U <= 4000:
    nf = a["state_time"][a["state_off"][1:] - 1].astype(np.int32)
    foff = np.zeros(int(nf.sum()) + 1, np.int64)
    for it in range(3):
        ne = C.c_int64(0)
        t0 = time.perf_counter()
        klat_amd._check(klat_amd.lib().klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
        t1 = time.perf_counter()
        print("frame post compute (count + bucket/fold/sort kernels + scratch allocation): %.2f ms, %d entries, %.2f G arcs/s" % (1e3 * (t1 - t0), ne.value, b.A / (t1 - t0) / 1e9))
    t0 = time.perf_counter(); o, l, p = b.frame_post(nf); print("frame post total incl. gather + D2H of %.0f MB: %.1f ms" % ((l.nbytes + p.nbytes + o.nbytes) / 1e6, 1e3 * (time.perf_counter() - t0)))
