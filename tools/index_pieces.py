"""Segment index on a batch of U lattices x S states (fanout 3) with the long lattices whole (KLAT_INDEX_PIECES=big) and cut
in pieces for the 512-thread class (small): device time of the sizing call, wall time of the device-side gather that merges
the pieces' runs (klat_index_device_records), and that the two give the same records.  usage: index_pieces.py [U] [S]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
CHECK = U * S <= 4000 * 5000
ctx = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260003, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=16)
L = klat_amd.lib()
f, keep = klat_amd.Batch._filter(None, None)
got = {}
for knob in ("big", "small"):
    os.environ["KLAT_INDEX_PIECES"] = knob
    b = klat_amd.Batch(ctx, arrays, host_threads=16, fuse_scales=(1.0, 0.1, 0.0))
    b.forward_backward(1.0, 0.1, 0.0, arc_posteriors=False)
    ms, t, g = C.c_float(0), [], []
    for _ in range(4):
        off = np.zeros(U + 1, np.int64)
        n = C.c_int64(0)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
        t.append(ms.value)
        dev, roff, nl, nr = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_int64(0)
        ctx.sync()
        t0 = time.perf_counter()
        klat_amd._check(L.klat_index_device_records(b.h, C.byref(dev), C.byref(roff), C.byref(nl), C.byref(nr)))
        ctx.sync()
        g.append((time.perf_counter() - t0) * 1e3)
    print("%-6s index %.3f ms (min; median %.3f)  gather %.3f ms wall (min)  %d records" % (knob, min(t), sorted(t)[1], min(g[1:]), n.value), flush=True)
    if CHECK:
        rec = np.zeros(n.value, klat_amd.SEGMENT_DTYPE)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), klat_amd._p(rec), C.c_int64(n.value), C.byref(n)))
        got[knob] = (off.copy(), rec)
    b.close()
if CHECK:
    same = np.array_equal(got["big"][0], got["small"][0]) and got["big"][1].tobytes() == got["small"][1].tobytes()
    print("records equal" if same else "RECORDS DIFFER")
    if not same:
        x, y = got["big"][1], got["small"][1]
        print("offsets equal:", np.array_equal(got["big"][0], got["small"][0]))
        for name in x.dtype.names:
            d = np.nonzero(x[name] != y[name])[0]
            print(name, len(d), d[:5], x[name][d[:5]], y[name][d[:5]])
