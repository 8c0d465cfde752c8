"""The C2 sweep against SweepArgs::role_mask (KLAT_SWEEP_ROLE_MASK): which of a workgroup's two waves walks the chain, by the
workgroup's turn on its CU ((blockIdx >> 8) & 7).  One process per mask would load the library anew; the mask is read at
every sweep, so one batch serves.  python tools/sweep_roles.py [lattices] [states] [number of masks | masks ...]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
masks = [int(x, 0) for x in sys.argv[3:]] or list(range(32))
if len(masks) == 1 and masks[0] > 31:
    masks = list(range(masks[0]))
ctx = klat_amd.Ctx(0)
a = klat_amd.synth(20260001, np.full(U, NS, np.int32), np.full(U, 3, np.int32), host_threads=16)
b = klat_amd.Batch(ctx, a, host_threads=16, fuse_scales=(1.0, 0.1, 0.0))
res = []
for rep in range(2):
    for m in masks:
        os.environ["KLAT_SWEEP_ROLE_MASK"] = str(m)
        for _ in range(24):
            b.forward_backward(1.0, 0.1, 0.0)
        ctx.sync()
        h = ctx.timing_history(16)
        res.append((m, float(np.median([t["sweep_ms"] for t in h]))))
best = {}
for m, t in res:
    best.setdefault(m, []).append(t)
order = sorted(best, key=lambda m: min(best[m]))
for m in (order if len(order) <= 16 else order[:12] + order[-4:] + [0]):
    print("mask 0x%02x: sweep %s ms" % (m, " / ".join("%.4f" % t for t in best[m])), flush=True)
