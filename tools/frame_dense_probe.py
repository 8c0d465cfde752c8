"""Frame posteriors against the arcs a frame holds: lattices of 1 000 states with fanout 3 .. 16 (synth gives every arc up to
eight states' span: a frame holds about 4.5 x fanout arcs).  python tools/frame_dense_probe.py"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
L = klat_amd.lib()
for fan in [int(x) for x in sys.argv[1:]] or (3, 6, 8, 12, 16, 24, 32):
    U = 24000 // fan
    arrays = klat_amd.synth(20260061, np.full(U, 1000, np.int32), np.full(U, fan, np.int32), host_threads=16)
    b = klat_amd.Batch(ctx, arrays, host_threads=16)
    b.forward_backward(1.0, 0.1, 0.0)
    nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
    foff = np.zeros(int(nf.sum()) + 1, np.int64); ne = C.c_int64(0); ms = C.c_float(0)
    t = []
    for _ in range(3):
        klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
        klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(ms)))
        t.append(ms.value)
    print("fanout %2d: %d lattices, %.1f M arcs, %.1f entries a frame: %.2f ms, %.1f G arcs/s" %
          (fan, U, b.A / 1e6, ne.value / max(int(nf.sum()), 1), min(t), b.A / min(t) / 1e6), flush=True)
    b.close()
