"""The segment index on C2-sized lattices with records that tie: a share of the arcs dead (infinite cost: their records
tie at log-zero and crowd the last bucket).  python tools/index_edge_probe.py"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = 4000
ctx = klat_amd.Ctx(0)
L = klat_amd.lib()
f, keep = klat_amd.Batch._filter(None, None)
base = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
rng = np.random.RandomState(1)
for share in (0.0, 0.001, 0.01, 0.05, 0.2):
    arrays = {k: v.copy() for k, v in base.items()}
    dead = rng.rand(len(arrays["graph"])) < share
    dead &= arrays["dst"] != arrays["src"] + 1
    arrays["graph"][dead] = np.inf
    arrays["acoustic"][dead] = np.inf
    b = klat_amd.Batch(ctx, arrays, host_threads=16)
    b.forward_backward(1.0, 0.1, 0.0)
    ms = C.c_float(0); t = []
    for _ in range(3):
        off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
        t.append(ms.value)
    print("dead arcs %.1f %%: %.3f ms, %.1f G arcs/s" % (100 * share, min(t), b.A / min(t) / 1e6), flush=True)
    b.close()
