"""Segment index and frame posteriors on a configs[4]-shaped mix (sizes log-uniform in 200 .. 20 000 states, fanout 2 / 3 / 4 / 6):
device time of the sizing calls, and how the lattices split over the index kernel's size classes."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
rng = np.random.RandomState(20260004)
ns = np.exp(rng.uniform(np.log(200), np.log(20000), U)).astype(np.int32)
fo = rng.choice([2, 3, 4, 6], U).astype(np.int32)
arrays = klat_amd.synth(20260004, ns, fo, host_threads=16)
arcs = np.diff(arrays["arc_off"])
edges = [0, 1536, 3072, 6144, 8192, 15360, 16384, 1 << 30]
print("lattices per class (arcs <= %s): %s; arcs per class: %s" % (edges[1:], np.histogram(arcs, edges)[0].tolist(),
      [int(arcs[(arcs > lo) & (arcs <= hi)].sum()) for lo, hi in zip(edges[:-1], edges[1:])]))
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, arrays, host_threads=16)
b.forward_backward(1.0, 0.1, 0.0)
L = klat_amd.lib()
f, keep = klat_amd.Batch._filter(None, None)
ms = C.c_float(0)
for _ in range(3):
    off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
    klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
    klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
    print("segment index: %.3f ms, %d records, %.1f G arcs/s" % (ms.value, n.value, b.A / ms.value / 1e6), flush=True)
nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
foff = np.zeros(int(nf.sum()) + 1, np.int64); ne = C.c_int64(0)
for _ in range(3):
    klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
    klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(ms)))
    print("frame posteriors: %.3f ms, %d entries, %.1f G arcs/s" % (ms.value, ne.value, b.A / ms.value / 1e6), flush=True)
import time
for _ in range(2):
    t0 = time.perf_counter(); off, rec = b.segment_index(); t1 = time.perf_counter()
    print("segment index incl. the fetch (gather / merge of the pieces' runs + %.0f MB to the host): %.1f ms" % (rec.nbytes / 1e6, (t1 - t0) * 1e3), flush=True)
