"""The accumulators on C2-sized lattices away from the synthetic defaults: all costs equal (records that tie on finite
values), a large exclude list, few distinct words (every frame folds), long arcs.  python tools/accum_edge_probe.py"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = 4000
ctx = klat_amd.Ctx(0)
L = klat_amd.lib()
base = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
rng = np.random.RandomState(1)
def run(name, arrays, include=None, exclude=None):
    b = klat_amd.Batch(ctx, arrays, host_threads=16)
    b.forward_backward(1.0, 0.1, 0.0)
    f, keep = klat_amd.Batch._filter(include, exclude)
    ms = C.c_float(0); t = []
    for _ in range(3):
        off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
        t.append(ms.value)
    nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
    foff = np.zeros(int(nf.sum()) + 1, np.int64); ne = C.c_int64(0); t2 = []
    for _ in range(3):
        klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
        klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(ms)))
        t2.append(ms.value)
    print("%-40s segment index %.3f ms (%d records), frame posteriors %.3f ms (%d entries)" % (name, min(t), n.value, min(t2), ne.value), flush=True)
    b.close()
run("synthetic defaults", base)
a = {k: v.copy() for k, v in base.items()}; a["graph"][:] = 1.0; a["acoustic"][:] = 2.0
run("all costs equal", a)
run("exclude list of 300 words", base, exclude=[int(x) for x in rng.randint(1, 1000, size=300)])
run("include list of 100 words", base, include=[int(x) for x in rng.randint(1, 1000, size=100)])
a = {k: v.copy() for k, v in base.items()}; a["olabel"] = (a["olabel"] % 4).astype(np.int32); a["ilabel"] = a["olabel"].copy()
run("four distinct words", a)
a = {k: v.copy() for k, v in base.items()}
S_of = np.repeat(np.full(U, 2000, np.int32), np.diff(a["arc_off"]))
far = (np.arange(len(a["dst"])) % 10 == 3) & (a["dst"] != a["src"] + 1)
a["dst"][far] = np.minimum(a["src"][far] + 100, S_of[far] - 1)
run("a tenth of the arcs 100 states long", a)
