"""Device time of the frame posteriors on the C2 batch: the lean kernel + the general one over what it
left (default) against the general kernel alone (KLAT_FRAME_GENERAL=1); KLAT_ACCUM_PROFILE=1 adds the
per-phase shader cycles."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260001, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=16)
b = klat_amd.Batch(ctx, arrays, host_threads=16)
b.forward_backward(1.0, 0.1, 0.0)
L = klat_amd.lib()
nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
foff = np.zeros(int(nf.sum()) + 1, np.int64); ne = C.c_int64(0); ms = C.c_float(0)
for mode in ("general", "lean", "general", "lean"):
    if mode == "general":
        os.environ["KLAT_FRAME_GENERAL"] = "1"
    else:
        os.environ.pop("KLAT_FRAME_GENERAL", None)
    t = []
    for _ in range(4):
        klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
        klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(ms)))
        t.append(round(ms.value, 3))
    print(mode, t, ne.value, flush=True)
