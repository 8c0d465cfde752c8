"""klat_batch_create on a batch whose arrays are already on the device: the device passes of the
stream packer (klat_batch_get_pack_timings) and the wall time of the call, then one forward-backward
-- the one-shot path every tool runs.  usage: pack_time.py [lattices] [states] [fanout] [reps] [fuse: 1|0] [borrow: 0|1]
(KLAT_PACK_PROFILE=1 adds the count pass's phase cycles and the host stages of klat_batch_create on stderr)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
FO = int(sys.argv[3]) if len(sys.argv) > 3 else 3
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
FUSE = (1.0, 0.1, 0.0) if (len(sys.argv) <= 5 or sys.argv[5] != "0") else None
BORROW = len(sys.argv) > 6 and sys.argv[6] == "1"
if NS == 0:  # the configs[4] mix: `lattices` sizes log-uniform in 200 .. 20 000 states, fanout in {2, 3, 4, 6}
    rng = np.random.RandomState(20260004)
    a = klat_amd.synth(20260004, np.exp(rng.uniform(np.log(200), np.log(20000), U)).astype(np.int32), rng.choice([2, 3, 4, 6], U).astype(np.int32))
else:
    a = klat_amd.synth(20260001, np.full(U, NS, np.int32), np.full(U, FO, np.int32))
hip = C.CDLL("libamdhip64.so")
dev, ptrs = dict(a), []
for name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time"):
    h = np.ascontiguousarray(a[name])
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(max(h.nbytes, 4))) == 0
    assert hip.hipMemcpy(p, C.c_void_p(h.ctypes.data), C.c_size_t(h.nbytes), C.c_int(1)) == 0
    ptrs.append(p)
    dev[name] = int(p.value)
ctx = klat_amd.Ctx(0)
A = int(a["arc_off"][-1])
for rep in range(REPS):
    t0 = time.perf_counter()
    b = klat_amd.Batch(ctx, dev, fuse_scales=FUSE, borrow=BORROW)
    t1 = time.perf_counter()
    b.forward_backward(1.0, 0.1, 0.0, arc_posteriors=False)
    ctx.sync()
    t2 = time.perf_counter()
    pt, ft = b.pack_timings(), b.timings()
    dev_ms = pt["device_ms"] + ft["total_ms"]
    print("rep %d: create %.2f ms wall (device passes %.3f: count %.3f level %.3f rounds %.3f fill %.3f prep %.3f) | fb %.3f ms device, %.2f wall | one-shot device %.3f ms = %.1f G arcs/s, wall %.2f ms = %.1f G arcs/s"
          % (rep, (t1 - t0) * 1e3, pt["device_ms"], pt["count_ms"], pt["level_plan_ms"], pt["rounds_ms"], pt["fill_ms"],
             pt["posterior_prep_ms"], ft["total_ms"], (t2 - t1) * 1e3, dev_ms, A / dev_ms / 1e6, (t2 - t0) * 1e3, A / (t2 - t0) / 1e9), flush=True)
    b.close()
