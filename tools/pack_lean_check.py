"""The lean stream packer (csrc/klat_pack_lean.hip) against the general one (KLAT_PACK_LEAN=0): the same streams, bit
for bit, on batches that mix both kinds of lattices; plain and fused costs.  usage: pack_lean_check.py [big]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402


def build(ctx, a, lean, **kw):
    os.environ["KLAT_PACK_LEAN"] = "1" if lean else "0"  # (1: also where a small batch would keep the general passes for the segment path)
    b = klat_amd.Batch(ctx, a, **kw)
    info = b.info()
    st = b.fused_streams() if kw.get("fuse_scales") else b.streams()
    b.close()
    os.environ.pop("KLAT_PACK_LEAN", None)
    return info, st


def stretch(a, ns, rng, frac, far):
    lat = np.repeat(np.arange(len(ns)), np.diff(a["arc_off"]))
    src, dst = a["src"], a["dst"].copy()
    st = (rng.rand(len(dst)) < frac) & (dst != src + 1)
    dst[st] = np.minimum(src[st] + rng.randint(2, far, size=int(st.sum())), ns[lat][st] - 1)
    a["dst"] = np.maximum(dst, src + 1).astype(np.int32)


def cases(big):
    rng = np.random.RandomState(7)
    # 1: what the tests pack: small lattices, every fanout
    ns = np.concatenate([[1, 2, 3], rng.randint(2, 400, size=45), [1500]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4, 7], size=len(ns)).astype(np.int32)
    yield "small mix", klat_amd.synth(11, ns, fo), {}
    a = klat_amd.synth(12, ns, fo)
    stretch(a, ns, rng, 0.3, 60)
    fin = rng.rand(len(a["final_graph"])) < 0.15
    a["final_graph"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
    a["final_acoustic"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
    yield "arcs up to 60 states, many finals", a, {}
    a = klat_amd.synth(13, ns, fo)
    stretch(a, ns, rng, 0.05, 150)
    yield "a few arcs beyond the lean span (mixed jobs)", a, {}
    for G, W in ((1, 16), (2, 8), (4, 32), (8, 64), (64, 16)):
        a = klat_amd.synth(14 + G, ns, fo)
        stretch(a, ns, rng, 0.2, 40)
        yield "fixed lanes %d ring %d" % (G, W), a, {"lanes_per_lattice": G, "ring_slots": W}
    ns = np.full(300, 700, np.int32)
    yield "uniform 300 x 700 x 3", klat_amd.synth(21, ns, np.full(300, 3, np.int32)), {}
    ns = np.exp(rng.uniform(np.log(50), np.log(6000), size=200)).astype(np.int32)
    yield "skewed 200", klat_amd.synth(22, ns, rng.choice([2, 3, 4, 6], size=200).astype(np.int32)), {}
    if big:
        yield "C2 / 10", klat_amd.synth(20260001, np.full(1000, 2000, np.int32), np.full(1000, 3, np.int32)), {}


def main():
    big = len(sys.argv) > 1
    ctx = klat_amd.Ctx(0)
    bad = 0
    for name, a, kw in cases(big):
        for fuse in (None, (1.0, 0.1, 0.5)):
            k = dict(kw)
            if fuse:
                k["fuse_scales"] = fuse
            ig, sg_ = build(ctx, a, False, **k)
            il, sl_ = build(ctx, a, True, **k)
            same_info = all(ig[x] == il[x] for x in ("lanes_per_lattice", "ring_slots", "num_jobs", "max_rounds", "fwd_slots", "bwd_slots", "far_terms", "num_layouts"))
            diffs = [int(np.sum(x.view(np.uint8 if x.dtype.itemsize == 1 else (np.uint16 if x.dtype.itemsize == 2 else np.uint32)) !=
                                y.view(np.uint8 if y.dtype.itemsize == 1 else (np.uint16 if y.dtype.itemsize == 2 else np.uint32)))) if x.shape == y.shape else -1
                     for x, y in zip(sg_, sl_)]
            ok = same_info and all(d == 0 for d in diffs)
            bad += 0 if ok else 1
            print("%-50s %-6s %s  slots %d  diffs %s%s" % (name, "fused" if fuse else "plain", "ok" if ok else "MISMATCH", ig["fwd_slots"] + ig["bwd_slots"], diffs,
                                                            "" if same_info else "  info %s vs %s" % (ig, il)))
            if not ok and diffs[0] > 0:
                w = np.nonzero(sg_[0] != sl_[0])[0]
                print("   first differing slots:", w[:8], [hex(int(x) & 0xFFFFFFFF) for x in sg_[0][w[:8]]], [hex(int(x) & 0xFFFFFFFF) for x in sl_[0][w[:8]]])
    ctx.close()
    print("FAILED" if bad else "all equal")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
