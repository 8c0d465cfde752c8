"""Long lattices swept in segments (csrc/klat_chain.hip): ComputeLatticeAlphasAndBetas (SURVEY.md A.1;
call sites latbin/lattice-to-word-frame-post.cc:88-90, kwsbin2/lattice-word-index-segment.cc:63) is linear
in the (+, x) semiring, so a lattice is cut where few states reach across, the cuts' transfer matrices are
combined serially and every segment is swept with its boundary known.  Checked against the oracle, against
the lockstep streams on the same lattices, for the rule that picks the path, and with the callers on top
(arc posteriors, segment index, klat_viterbi on the streams the lattices keep)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402
from test_gpu_parity import _check, _oracle_batch  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = klat_amd.Ctx(0)
    yield c
    c.close()


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _lattices(seed, ns, fo, finals_anywhere=False):
    a = klat_amd.synth(seed, np.asarray(ns, np.int32), np.asarray(fo, np.int32))
    rng = np.random.RandomState(seed)
    # more final states, with weights -- near the end, where a lattice has them, or anywhere; a few arcs removed
    # (infinite cost)
    fin = rng.rand(len(a["final_graph"])) < (0.01 if finals_anywhere else 0.3)
    if not finals_anywhere:
        near_end = np.zeros(len(fin), bool)
        for u in range(len(ns)):
            near_end[max(a["state_off"][u], a["state_off"][u + 1] - 12):a["state_off"][u + 1]] = True
        fin &= near_end
    a["final_graph"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 3
    a["final_acoustic"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 5
    dead = rng.rand(len(a["graph"])) < 0.005
    for u in range(len(ns)):  # (not the arc to the next state: every lattice keeps a path to its end)
        a0, a1 = a["arc_off"][u], a["arc_off"][u + 1]
        dead[a0:a1] &= (np.arange(a1 - a0) % fo[u]) != 0
    a["graph"][dead] = np.inf
    return a


def _run(ctx, arrays, scales, **env):
    with _env(**env):
        b = klat_amd.Batch(ctx, arrays, fuse_scales=scales)
    b.forward_backward(*scales)
    alpha, beta = b.alpha_beta()
    total, tf, tb = b.totals()
    post = b.arc_logpost()
    return b, (alpha, beta, total, tf, tb, post)


@pytest.mark.parametrize("seg", [64, 256])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_segments_match_the_oracle_and_the_streams(ctx, seg, scales):
    ns = [3000, 700, 5000, 40, 1300, 2 * seg + 2, 2 * seg - 2, 4097, 1]
    fo = [3, 2, 4, 3, 6, 3, 3, 1, 1]
    arrays = _lattices(20260005, ns, fo)
    b, got = _run(ctx, arrays, scales, KLAT_CHAIN=1, KLAT_CHAIN_SEG=seg)
    info = b.info()
    # a lattice's segments hold ~4 x seg records each: seg states at fanout 3, fewer at a larger fanout
    def seg_len(s, f):
        if s < 2:
            return 1 << 30
        return max(32, int(4.0 * seg / (f * (s - 1) / s + 1.0)))
    seg_states = [seg_len(s, f) for s, f in zip(ns, fo)]
    want_chain = sum(1 for s, l in zip(ns, seg_states) if s >= 2 * l)
    assert info["chain_lattices"] == want_chain and want_chain >= 4 and not b.chain_fallback(), info
    assert info["chain_segments"] == sum(s // l for s, l in zip(ns, seg_states) if s >= 2 * l), info
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    alpha, beta, total, tf, tb, post = got
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(total, wt, 1e-8), _check(post, wpost, 1e-8)
    assert np.max(np.abs(tf - tb)[np.isfinite(tf)], initial=0.0) < 1e-8
    b.close()
    b0, ref = _run(ctx, arrays, scales, KLAT_CHAIN=0)
    assert b0.info()["chain_lattices"] == 0
    for g, r in zip(got, ref):
        _check(g, r, 1e-9)
    b0.close()


def test_arcs_of_every_length_a_cut_allows(ctx):
    """Arcs up to 16 states long (the most a cut allows), in- and out-degrees up to 12, unreachable states,
    states that reach no final state; a lattice with a longer arc stays on the streams."""
    rng = np.random.RandomState(7)
    lats = []
    for S, span in ((1500, 16), (900, 5), (2000, 2), (1100, 17)):
        src, dst = [], []
        for s in range(S - 1):
            if rng.rand() < 0.02 and s > 0 and span >= 5:
                continue  # no out-arc: beta = log-zero unless final (the others still reach the end around it)
            for _ in range(1 + rng.randint(0, 4) + (8 if rng.rand() < 0.01 else 0)):
                d = s + 1 + rng.randint(0, min(span, S - 1 - s))
                src.append(s), dst.append(d)
        lats.append((S, np.array(src, np.int32), np.array(dst, np.int32)))
    so, ao = np.zeros(len(lats) + 1, np.int64), np.zeros(len(lats) + 1, np.int64)
    for i, (S, s_, d_) in enumerate(lats):
        so[i + 1], ao[i + 1] = so[i] + S, ao[i] + len(s_)
    A, St = int(ao[-1]), int(so[-1])
    arrays = {"state_off": so, "arc_off": ao, "src": np.concatenate([l[1] for l in lats]), "dst": np.concatenate([l[2] for l in lats]),
              "ilabel": rng.randint(0, 50, A).astype(np.int32), "graph": (rng.rand(A) * 8).astype(np.float32),
              "acoustic": (rng.rand(A) * 60 - 10).astype(np.float32), "final_graph": np.full(St, np.inf, np.float32),
              "final_acoustic": np.full(St, np.inf, np.float32), "state_time": np.zeros(St, np.int32)}
    arrays["olabel"] = arrays["ilabel"].copy()
    for i in range(len(lats)):
        arrays["final_graph"][so[i + 1] - 1] = 0.5
        arrays["final_acoustic"][so[i + 1] - 1] = 0.25
        more = so[i + 1] - 1 - rng.randint(0, 10, 4)
        arrays["final_graph"][more], arrays["final_acoustic"][more] = 1.5, 2.0
    scales = (1.0, 0.3, 0.0)
    b, got = _run(ctx, arrays, scales, KLAT_CHAIN=1, KLAT_CHAIN_SEG=128)
    assert b.info()["chain_lattices"] == 3 and not b.chain_fallback(), b.info()
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    alpha, beta, total, tf, tb, post = got
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(total, wt, 1e-8), _check(post, wpost, 1e-8)
    b.close()


def test_the_path_is_taken_when_the_batch_waits_for_its_longest_lattices(ctx):
    """A scaled-down BASELINE configs[4] mix (sizes log-uniform over two decades): the long lattices go; a
    uniform batch that fills the chip keeps its streams; fused scales or not."""
    rng = np.random.RandomState(3)
    ns = np.exp(rng.uniform(np.log(200), np.log(20000), 300)).astype(np.int32)
    fo = rng.choice([2, 3, 4, 6], 300).astype(np.int32)
    arrays = klat_amd.synth(11, ns, fo)
    scales = (1.0, 0.1, 0.0)
    b, got = _run(ctx, arrays, scales, KLAT_CHAIN=None, KLAT_CHAIN_SEG=None)
    info = b.info()
    assert info["chain_lattices"] >= 20 and info["chain_lattices"] < 300, info
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    alpha, beta, total, tf, tb, post = got
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(total, wt, 1e-8), _check(post, wpost, 1e-8)
    # the callers on top: the segment index reads alpha / beta, klat_viterbi the streams those lattices keep
    off, rec = b.segment_index()
    b.viterbi(*scales)
    fw, bw, best = b.viterbi_costs()
    b.close()
    b0, ref = _run(ctx, arrays, scales, KLAT_CHAIN=0)
    off0, rec0 = b0.segment_index()
    b0.viterbi(*scales)
    fw0, bw0, best0 = b0.viterbi_costs()
    assert np.array_equal(off, off0)
    for u in range(len(ns)):  # (records of nearly equal posterior may change places: compare them by key)
        r, r0 = rec[off[u]:off[u + 1]], rec0[off[u]:off[u + 1]]
        r, r0 = r[np.lexsort((r["t1"], r["t0"], r["label"]))], r0[np.lexsort((r0["t1"], r0["t0"], r0["label"]))]
        for f in ("label", "t0", "t1"):
            assert np.array_equal(r[f], r0[f])
        assert np.max(np.abs(r["logp"] - r0["logp"]), initial=0.0) < 1e-9
    # (the Viterbi twin of the segment path adds a path's costs in another order, from costs kept to 43 fraction bits)
    for g, w in ((fw, fw0), (bw, bw0), (best, best0)):
        assert np.array_equal(np.isinf(g), np.isinf(w)) and np.allclose(g, w, rtol=1e-12, atol=1e-9)
    b0.close()
    uniform = klat_amd.synth(12, np.full(3000, 2000, np.int32), np.full(3000, 3, np.int32))
    with _env(KLAT_CHAIN=None):
        bu = klat_amd.Batch(ctx, uniform, fuse_scales=scales)
        assert bu.info()["chain_lattices"] == 0
        bu.close()
        # without fused scales the segment path's records are built for the scales of the call, again when they change
        bn = klat_amd.Batch(ctx, arrays)
        assert bn.info()["chain_lattices"] == info["chain_lattices"]
        for sc in ((1.0, 0.1, 0.0), (0.7, 0.3, 0.5), (1.0, 0.1, 0.0)):
            bn.forward_backward(*sc)
            assert not bn.chain_fallback()
            wa2, wb2, wt2, _ = _oracle_batch(arrays, *sc)
            a2, b2 = bn.alpha_beta()
            _check(a2, wa2, 1e-8), _check(b2, wb2, 1e-8), _check(bn.totals()[0], wt2, 1e-8)
        bn.close()


def test_values_the_segments_cannot_hold_fall_back_to_the_streams(ctx):
    """Two interleaved chains, one of cheap arcs and one of dear ones: a few states on, neighbouring states are
    hundreds of nats apart -- more than a lane of the segment path keeps between two rescalings.  It says so, and
    the streams these lattices keep sweep them inside the same call; a tame lattice beside them is not disturbed."""
    S = 1200
    src, dst, cost = [0, 0], [1, 2], [1.0, 0.5]
    for s in range(1, S - 2):
        src.append(s), dst.append(s + 2), cost.append(0.1 if s % 2 == 0 else 30.0)
        if s % 100 == 0:
            src.append(s), dst.append(s + 1), cost.append(2.0)
    src.append(S - 2), dst.append(S - 1), cost.append(0.3)
    order = np.argsort(np.array(src), kind="stable")
    src, dst, cost = np.array(src, np.int32)[order], np.array(dst, np.int32)[order], np.array(cost, np.float32)[order]
    tame = klat_amd.synth(5, np.array([900], np.int32), np.array([3], np.int32))
    A = len(src)
    hard = {"state_off": np.array([0, S], np.int64), "arc_off": np.array([0, A], np.int64), "src": src, "dst": dst,
            "ilabel": np.ones(A, np.int32), "olabel": np.ones(A, np.int32), "graph": cost, "acoustic": np.zeros(A, np.float32),
            "final_graph": np.full(S, np.inf, np.float32), "final_acoustic": np.full(S, np.inf, np.float32),
            "state_time": np.arange(S, dtype=np.int32)}
    hard["final_graph"][S - 1] = hard["final_acoustic"][S - 1] = 0.0
    arrays = {}
    for k in hard:
        if k in ("state_off", "arc_off"):
            arrays[k] = np.concatenate([hard[k], tame[k][1:] + hard[k][-1]])
        else:
            arrays[k] = np.concatenate([hard[k], tame[k]])
    scales = (1.0, 1.0, 0.0)
    b, got = _run(ctx, arrays, scales, KLAT_CHAIN=1, KLAT_CHAIN_SEG=128)
    assert b.info()["chain_lattices"] == 2 and b.chain_fallback(), b.info()
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    alpha, beta, total, tf, tb, post = got
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(total, wt, 1e-8), _check(post, wpost, 1e-8)
    b.forward_backward(*scales)  # (and again: the flag is cleared and raised anew)
    assert b.chain_fallback()
    a2, b2 = b.alpha_beta()
    assert np.array_equal(a2, alpha) and np.array_equal(b2, beta)
    b.close()


def test_final_states_in_the_middle_of_a_long_lattice(ctx):
    """A final state a thousand states before the end has a beta hundreds of orders above its neighbours': more than
    one lane scale holds.  The segment path says so and the streams take over; the results are the oracle's."""
    ns, fo = [3000, 1300, 4097], [3, 6, 1]
    arrays = _lattices(20260006, ns, fo, finals_anywhere=True)
    scales = (1.0, 1.0, 0.0)
    b, got = _run(ctx, arrays, scales, KLAT_CHAIN=1, KLAT_CHAIN_SEG=64)
    assert b.info()["chain_lattices"] == 3 and b.chain_fallback(), b.info()
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    alpha, beta, total, tf, tb, post = got
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(total, wt, 1e-8), _check(post, wpost, 1e-8)
    b.close()


@pytest.mark.parametrize("seg", [64, 256])
def test_viterbi_twin_in_segments(ctx, seg):
    """klat_viterbi on lattices cut into segments: the (min,+) twin of the three kernels over cost records, against the
    streams on the same lattices (themselves checked against the oracle in test_gpu_viterbi.py), costs and keep masks;
    final states anywhere are no trouble here (no scale to leave)."""
    ns = [3000, 700, 5000, 40, 1300, 2 * seg + 2, 4097, 1]
    fo = [3, 2, 4, 3, 6, 3, 1, 1]
    arrays = _lattices(20260007, ns, fo, finals_anywhere=True)
    scales = (0.7, 0.1, 0.5)
    res = []
    for chain in (1, 0):
        with _env(KLAT_CHAIN=chain, KLAT_CHAIN_SEG=seg):
            b = klat_amd.Batch(ctx, arrays, fuse_scales=scales)
        assert (b.info()["chain_lattices"] >= 5) == (chain == 1)
        b.viterbi(*scales)
        fw, bw, best = b.viterbi_costs()
        ka, kf, lb = b.prune_mask(8.0)
        res.append((fw, bw, best, ka, kf, lb))
        b.close()
    (fw, bw, best, ka, kf, lb), (fw0, bw0, best0, ka0, kf0, lb0) = res
    for g, w in ((fw, fw0), (bw, bw0), (best, best0), (lb, lb0)):
        assert np.array_equal(np.isinf(g), np.isinf(w)) and np.allclose(g, w, rtol=1e-12, atol=1e-9)
    # decisions within 1e-7 of the cutoff may differ (sums in another order); the others may not
    so, ao = arrays["state_off"], arrays["arc_off"]
    lat_of_arc = np.repeat(np.arange(len(ns)), np.diff(ao))
    gsrc = arrays["src"] + so[lat_of_arc]
    gdst = arrays["dst"] + so[lat_of_arc]
    g = np.where(np.isinf(arrays["graph"]), np.inf, np.float32(scales[0]) * arrays["graph"]).astype(np.float32)
    g = np.where(arrays["ilabel"] != 0, g + np.float32(scales[2]), g).astype(np.float32)
    ac = (np.float32(scales[1]) * arrays["acoustic"]).astype(np.float32)
    through = fw0[gsrc] + (g.astype(np.float64) + ac.astype(np.float64)) + bw0[gdst]
    sure = ~(np.abs(through - (best0[lat_of_arc] + 8.0)) <= 1e-7)
    assert np.array_equal(ka[sure], ka0[sure]) and np.mean(ka != ka0) < 1e-4
    assert np.mean(kf != kf0) < 1e-3


def test_costs_the_records_cannot_hold_leave_the_lattices_to_their_streams(ctx):
    """exp(-cost) as a double holds |cost| up to ~700: with larger costs among a cut lattice's arcs (acoustic scale 1 on
    costs of hundreds) the records are not used and the streams sweep, silently and correctly."""
    ns, fo = [2000, 900], [3, 2]
    arrays = klat_amd.synth(9, np.asarray(ns, np.int32), np.asarray(fo, np.int32))
    arrays["acoustic"] *= 25.0  # up to 1000
    with _env(KLAT_CHAIN=1, KLAT_CHAIN_SEG=128):
        b = klat_amd.Batch(ctx, arrays)
    assert b.info()["chain_lattices"] == 2
    for scales in ((1.0, 1.0, 0.0), (1.0, 0.1, 0.0)):  # the first cannot use the records, the second can
        b.forward_backward(*scales)
        wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
        alpha, beta = b.alpha_beta()
        _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(b.totals()[0], wt, 1e-8), _check(b.arc_logpost(), wpost, 1e-8)
    b.close()
