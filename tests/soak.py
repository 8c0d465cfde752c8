"""(A checker script, not a pytest module: it lives with the tests because it calls the oracle.)
Randomised parity soak: random batch shapes, layouts, scales, final-state patterns, infinite
costs and long-span arcs through forward-backward, the index accumulators, the Viterbi twin and
the utterance scores, all against the oracle.  usage: tests/soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
from oracle import oracle as orc
from test_gpu_parity import _oracle_batch, _check
from test_gpu_accum import _split, _check_sorted_records

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = klat_amd.Ctx(0)
t_end = time.time() + budget
it = 0
n_level = 0
while time.time() < t_end:
    seed = seed0 * 100003 + it
    rng = np.random.RandomState(seed)
    U = int(rng.choice([1, 3, 17, 64, 200]))
    ns = np.maximum(1, np.exp(rng.uniform(0, np.log(rng.choice([30, 300, 1500])), U))).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4, 6, 9], U).astype(np.int32)
    wide = int(rng.choice([0, 0, 6, 12, 40]))  # wide > 0: levels of that many states (the level path takes them)
    if wide:
        ns = np.maximum(ns, 8)
    a = klat_amd.synth(seed, ns, fo, wide=wide)
    if rng.rand() < 0.4:  # state times apart: frames without an arc boundary share their records
        a["state_time"] = (a["state_time"] * int(rng.choice([2, 5]))).astype(np.int32)
    lat = np.repeat(np.arange(U), np.diff(a["arc_off"]))
    if not wide and rng.rand() < 0.5 and len(a["dst"]):  # long-span arcs
        src, dst = a["src"], a["dst"].copy()
        st = (rng.rand(len(dst)) < 0.3) & (dst != src + 1)
        dst[st] = np.minimum(src[st] + rng.randint(2, 200, size=int(st.sum())), ns[lat][st] - 1)
        a["dst"] = dst.astype(np.int32)
    if rng.rand() < 0.5:  # extra final states
        fin = rng.rand(len(a["final_graph"])) < 0.1
        a["final_graph"][fin] = (rng.rand(int(fin.sum())) * 4).astype(np.float32)
        a["final_acoustic"][fin] = (rng.rand(int(fin.sum())) * 4).astype(np.float32)
    if rng.rand() < 0.3 and len(a["graph"]):  # dead arcs
        dead = rng.rand(len(a["graph"])) < 0.05
        a["graph"][dead] = np.inf
        a["acoustic"][dead] = np.inf
    if rng.rand() < 0.3:  # few labels: keys collide
        a["olabel"] = (a["olabel"] % 4).astype(np.int32)
        a["ilabel"] = a["olabel"].copy()
    G = int(rng.choice([0, 1, 2, 4, 8, 16]))
    W = int(rng.choice([0, 8, 16, 32, 64])) if G else 0
    if G and W and W * (64 // G) > 4032:
        W = 16
    scales = [(1.0, 1.0, 0.0), (1.0, 0.1, 0.0), (0.5, 0.2, 0.7), (0.0, 1.0, 0.0)][rng.randint(4)]
    b = klat_amd.Batch(ctx, a, lanes_per_lattice=G, ring_slots=W)
    n_level += b.info()["level_lattices"]
    b.forward_backward(*scales)
    alpha, beta = b.alpha_beta(); total, tf, tb = b.totals(); post = b.arc_logpost()
    wa, wb, wt, wpost = _oracle_batch(a, *scales)
    ok = np.isfinite(wt)
    _check(alpha, wa, 1e-7); _check(beta, wb, 1e-7); _check(total[ok], wt[ok], 1e-7)
    okarc = np.repeat(ok, np.diff(a["arc_off"]))
    _check(post[okarc], wpost[okarc], 1e-7)
    lats = _split(a)
    off, rec = b.segment_index()
    for u in rng.choice(U, min(U, 6), replace=False):
        if not ok[u]:
            continue
        sl = orc.scale_and_penalize(lats[u], *scales)
        al, be, to, _, _ = orc.alphas_betas(sl)
        t = a["state_time"][a["state_off"][u]:a["state_off"][u + 1]]
        want = orc.segment_index(sl, t, al, be, to)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec[off[u]:off[u + 1]]]
        _check_sorted_records(got, want, 3)
        uw = orc.utterance_index(sl)
        if uw:
            words = np.array(sorted(w for w, _ in uw), np.int32)
            wo = np.zeros(U + 1, np.int64); wo[u + 1:] = len(words)
            lp = b.utterance_scores(wo, words)
            wd = dict(uw)
            for w, p in zip(words, lp):
                assert abs(p - wd[int(w)]) < 1e-6 * max(1.0, abs(p)) or (np.isinf(p) and np.isinf(wd[int(w)])), (seed, u, w, p, wd[int(w)])
    # frame posteriors and position index on a few small lattices
    small = [u for u in range(U) if ok[u] and 0 < len(lats[u].dst) <= 400][:3]
    if small:
        nf = np.zeros(U, np.int32)
        for u in small:
            t = a["state_time"][a["state_off"][u]:a["state_off"][u + 1]]
            nf[u] = int(t.max()) if len(t) else 0
        foff, flab, flp = b.frame_post(nf)
        base = np.concatenate([[0], np.cumsum(nf)])
        for u in small:
            sl = orc.scale_and_penalize(lats[u], *scales)
            al, be, to, _, _ = orc.alphas_betas(sl)
            t = a["state_time"][a["state_off"][u]:a["state_off"][u + 1]]
            want, _ = orc.frame_post(sl, t, int(nf[u]), al, be, to)
            for k in range(int(nf[u])):
                g = sorted(zip(flab[foff[base[u] + k]:foff[base[u] + k + 1]].tolist(), flp[foff[base[u] + k]:foff[base[u] + k + 1]].tolist()))
                w = sorted((l, float(p)) for l, p in want[k])
                assert [x[0] for x in g] == [x[0] for x in w], (seed, u, k)
                assert all(abs(x[1] - y[1]) < 1e-4 * max(1.0, abs(y[1])) or x[1] == y[1] for x, y in zip(g, w)), (seed, u, k, g, w)
    b.viterbi(*scales)
    fw, bw, best = b.viterbi_costs()
    for u in rng.choice(U, min(U, 6), replace=False):
        sl = orc.scale_and_penalize(lats[u], *scales)
        wbeam, wfw, wbw = orc.lattice_beam(sl)
        s0, s1 = int(a["state_off"][u]), int(a["state_off"][u + 1])
        assert np.allclose(fw[s0:s1], wfw, rtol=1e-12, atol=1e-9, equal_nan=True), (seed, u)
        assert np.allclose(bw[s0:s1], wbw, rtol=1e-12, atol=1e-9, equal_nan=True), (seed, u)
    # PruneLattice decisions on a few lattices
    beam = float(rng.choice([0.5, 3.0, 12.0]))
    ka, kf, lb = b.prune_mask(beam)
    for u in rng.choice(U, min(U, 4), replace=False):
        sl = orc.scale_and_penalize(lats[u], *scales)
        if sl.S == 0:
            continue
        wbeam, wfw, wbw = orc.lattice_beam(sl)
        if not np.isfinite(wbw[0]):
            continue
        pr = sl.copy()
        oka = np.zeros(max(sl.A, 1), np.uint8); oks = np.zeros(max(sl.S, 1), np.uint8)
        orc.lib().kor_prune_lattice(orc.C.c_int32(sl.S), orc._p(pr.arc_begin), orc._p(pr.dst), orc._p(pr.graph), orc._p(pr.acoustic),
                                    orc._p(pr.final_graph), orc._p(pr.final_acoustic), orc.C.c_double(beam), orc._p(oka), orc._p(oks))
        a0, a1 = int(a["arc_off"][u]), int(a["arc_off"][u + 1])
        with np.errstate(invalid="ignore"):
            through = wfw[sl.src()] + (sl.graph.astype(np.float64) + sl.acoustic.astype(np.float64)) + wbw[sl.dst]
            sure = ~(np.abs(through - (wbw[0] + beam)) <= 1e-9)
        assert np.array_equal(ka[a0:a1][sure], oka[:sl.A][sure]), (seed, u)
        assert lb[u] == wbeam or abs(lb[u] - wbeam) < 1e-9 * max(1.0, abs(wbeam)), (seed, u, lb[u], wbeam)
    info = b.info()
    b.close()
    it += 1
    if it % 20 == 0:
        print("iteration %d ok (last: U=%d arcs=%d G=%d W=%d far=%d scales=%s)" % (it, U, len(a["dst"]), info["lanes_per_lattice"], info["ring_slots"], info["far_terms"], scales), flush=True)
print("soak ok: %d random batches, %d lattices through the level path" % (it, n_level))
