"""Where the accumulate kernels spend their cycles on the C2 batch (KLAT_ACCUM_PROFILE=1 makes the
library print the per-phase shader cycles of a workgroup) and their device times."""
import ctypes as C
import os, sys
import numpy as np
os.environ["KLAT_ACCUM_PROFILE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
b = klat_amd.Batch(ctx, arrays, host_threads=16)
b.forward_backward(1.0, 0.1, 0.0)
L = klat_amd.lib()
f, keep = klat_amd.Batch._filter(None, None)
off = np.zeros(U + 1, np.int64); n = C.c_int64(0); ms = C.c_float(0)
for _ in range(2):
    klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
    klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
    print("segment index: %.3f ms, %d records" % (ms.value, n.value), flush=True)
nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
foff = np.zeros(int(nf.sum()) + 1, np.int64); ne = C.c_int64(0)
for _ in range(2):
    klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
    klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(ms)))
    print("frame posteriors: %.3f ms, %d entries, %d frames" % (ms.value, ne.value, int(nf.sum())), flush=True)
