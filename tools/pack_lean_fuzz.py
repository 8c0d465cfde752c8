"""Differential fuzz of the two stream packers: random batches (sizes, fanouts, arc lengths up to and beyond the lean
passes' 64 states, parallel arcs, many or no final states, empty lattices, fixed or automatic lanes, plain or fused
costs) through KLAT_PACK_LEAN=1 and =0; the streams must agree bit for bit and the sweep's totals forward == backward.
usage: pack_lean_fuzz.py [iterations] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = klat_amd.Ctx(0)
bad = 0
for it in range(N):
    U = int(rng.choice([1, 3, 17, 40, 130]))
    hi = int(rng.choice([5, 60, 400, 3000]))
    ns = rng.randint(1, hi + 1, size=U).astype(np.int32)
    if rng.rand() < 0.3:
        ns[rng.randint(U)] = int(rng.choice([6000, 9000]))
    fo = rng.choice([1, 2, 3, 4, 6, 9], size=U).astype(np.int32)
    a = klat_amd.synth(1000 + it, ns, fo)
    lat = np.repeat(np.arange(U), np.diff(a["arc_off"]))
    if len(a["dst"]) and rng.rand() < 0.7:  # stretch some arcs
        far = int(rng.choice([10, 50, 64, 65, 100, 300]))
        src, dst = a["src"], a["dst"].copy()
        st = (rng.rand(len(dst)) < rng.choice([0.02, 0.3])) & (dst != src + 1)
        dst[st] = np.minimum(src[st] + rng.randint(2, far + 1, size=int(st.sum())), ns[lat][st] - 1)
        a["dst"] = np.maximum(dst, src + 1).astype(np.int32)
    if rng.rand() < 0.5:  # final states anywhere
        fin = rng.rand(len(a["final_graph"])) < rng.choice([0.01, 0.3])
        a["final_graph"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
        a["final_acoustic"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
    if len(a["graph"]) and rng.rand() < 0.2:  # dead arcs
        dead = rng.rand(len(a["graph"])) < 0.01
        a["graph"][dead] = np.inf
        a["acoustic"][dead] = np.inf
    kw = {}
    if rng.rand() < 0.6:
        G = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
        kw = {"lanes_per_lattice": G, "ring_slots": int(rng.choice([w for w in (8, 16, 32, 64) if w * (64 // G) <= 4032]))}
    if rng.rand() < 0.5:
        kw["fuse_scales"] = (1.0, float(rng.choice([0.1, 1.0])), float(rng.choice([0.0, 0.5])))
    got = []
    for lean in ("0", "1"):
        os.environ["KLAT_PACK_LEAN"] = lean
        b = klat_amd.Batch(ctx, a, **kw)
        info = b.info()
        st = b.fused_streams() if "fuse_scales" in kw else b.streams()
        sc = kw.get("fuse_scales", (1.0, 0.1, 0.0))
        b.forward_backward(*sc)
        total, tf, tb = b.totals()
        b.close()
        got.append((info, st, total, tf, tb))
    os.environ.pop("KLAT_PACK_LEAN", None)
    (ig, sg, tg, _, _), (il, sl, tl, tf, tb) = got
    same = all(ig[k] == il[k] for k in ("num_jobs", "fwd_slots", "bwd_slots", "far_terms", "max_rounds")) and all(x.tobytes() == y.tobytes() for x, y in zip(sg, sl))
    fin_ok = np.array_equal(np.isfinite(tg), np.isfinite(tl)) and (not np.any(np.isfinite(tl)) or float(np.max(np.abs((tf - tb)[np.isfinite(tl)]))) < 1e-6)
    if not (same and fin_ok):
        bad += 1
        print("iteration %d: MISMATCH streams %s totals %s U=%d kw=%s lean jobs %d of %d" % (it, same, fin_ok, U, kw, il["lean_jobs"], il["num_jobs"]), flush=True)
    elif it % 20 == 0:
        print("iteration %d ok: %d lattices, %d arcs, lean jobs %d of %d, %s" % (it, U, len(a["dst"]), il["lean_jobs"], il["num_jobs"], kw), flush=True)
ctx.close()
print("FAILED: %d of %d" % (bad, N) if bad else "all %d equal" % N)
sys.exit(1 if bad else 0)
