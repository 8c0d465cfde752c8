"""A/B of the two forms of the segment / position index kernel (csrc/klat_accum_hash.hip) on one box: device time of
the sizing call (HIP events, klat_get_accum_timings), records compared bit for bit.
  python tools/index_ab.py [U] [S] [--dups] [--position] [--profile]"""
import ctypes as C
import os, sys
import numpy as np
args = [x for x in sys.argv[1:] if not x.startswith("--")]
flags = [x for x in sys.argv[1:] if x.startswith("--")]
if "--profile" in flags:
    os.environ["KLAT_ACCUM_PROFILE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(args[0]) if len(args) > 0 else 10000
S = int(args[1]) if len(args) > 1 else 2000
ctx = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260001, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=16)
if "--dups" in flags:  # what real word lattices look like: few words, many states per time, most keys repeated
    arrays["olabel"] = (arrays["olabel"] % 16).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    arrays["state_time"] = (arrays["state_time"] // 8).astype(np.int32)
b = klat_amd.Batch(ctx, arrays, host_threads=16)
b.forward_backward(1.0, 0.1, 0.0)
L = klat_amd.lib()
f, keep = klat_amd.Batch._filter(None, None)
position = "--position" in flags
state_len = (arrays["src"][np.searchsorted(arrays["src"] * 0, 0)] * 0) if False else None
if position:
    # any per-state integer will do for the kernel: the state's number modulo 97
    state_len = np.concatenate([np.arange(int(n), dtype=np.int32) % 97 for n in np.diff(arrays["state_off"])])
res = {}
for name, env in (("v2", None), ("v1", "1"), ("v2", None), ("v1", "1")):
    if env is None:
        os.environ.pop("KLAT_INDEX_V1", None)
    else:
        os.environ["KLAT_INDEX_V1"] = env
    ms = C.c_float(0)
    best = 1e9
    for _ in range(5):
        if position:
            off, rec = b.position_index(state_len)
        else:
            off = np.zeros(U + 1, np.int64); n = C.c_int64(0)
            klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ms), None))
        best = min(best, ms.value)
    if not position:
        off, rec = b.segment_index()
    res.setdefault(name, (off.copy(), rec.tobytes()))
    print("%s: %.3f ms (best of 5), %d records, %.1f G arcs/s" % (name, best, len(rec), b.A / best / 1e6), flush=True)
same = np.array_equal(res["v1"][0], res["v2"][0]) and res["v1"][1] == res["v2"][1]
print("records identical:", same)
sys.exit(0 if same else 1)
