"""Device time of the segment index's sizing call on a batch of U lattices x S states (fanout 3); environment decides
the kernel's knobs.  python tools/index_time.py [U] [S]"""
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
f, keep = klat_amd.Batch._filter(None, None)
for knob in sys.argv[3:] or [""]:
    for kv in knob.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            os.environ[k] = v
    ms = C.c_float(0); t = []
    for _ in range(7):
        off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
        t.append(ms.value)
    print("%-40s %.3f ms (min), %.3f (median)  %d records" % (knob, min(t), sorted(t)[3], n.value), flush=True)
