"""The level-synchronous path (csrc/klat_level.hip): lattices whose topological depth is far below
their states are walked level by level, one wave per lattice, every state folding its arcs into a
log-sum-exp (what ComputeLatticeAlphasAndBetas computes with LogAdd; call site
latbin/lattice-to-word-frame-post.cc:88-90) -- checked against the oracle, against the stream
path on the same lattices, in batches that mix both kinds, and for the rule that picks the path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
from oracle import oracle as orc  # noqa: E402
import klat_amd  # noqa: E402
from test_gpu_parity import _check, _oracle_batch  # noqa: E402
from test_gpu_accum import _split  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = klat_amd.Ctx(0)
    yield c
    c.close()


def _wide(seed, ns, fo, width):
    a = klat_amd.synth(seed, np.asarray(ns, np.int32), np.asarray(fo, np.int32), wide=width)
    rng = np.random.RandomState(seed)
    # a few more final states with non-trivial weights, and a few removed arcs (infinite cost)
    fin = rng.rand(len(a["final_graph"])) < 0.02
    a["final_graph"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 3
    a["final_acoustic"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 5
    a["graph"][rng.rand(len(a["graph"])) < 0.01] = np.inf
    return a


@pytest.mark.parametrize("width", [12, 32, 200])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_wide_lattices_take_the_level_path_and_match_the_oracle(ctx, width, scales):
    ns = [64, 100, 257, 1000, 2000, 10 * width + 1]
    fo = [3, 2, 4, 3, 3, 6]
    arrays = _wide(77, ns, fo, width)
    b = klat_amd.Batch(ctx, arrays)
    info = b.info()
    depth = b.topological_depth()
    want_level = sum(1 for s, d in zip(ns, depth) if 64 <= s <= 8192 and d <= s // 8)
    assert info["level_lattices"] == want_level and want_level >= 3, (info, depth)
    b.forward_backward(*scales)
    alpha, beta = b.alpha_beta()
    total, tf, tb = b.totals()
    post = b.arc_logpost()
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    _check(alpha, wa, 1e-9), _check(beta, wb, 1e-9), _check(total, wt, 1e-9), _check(post, wpost, 1e-9)
    assert np.max(np.abs(tf - tb)[np.isfinite(tf)], initial=0.0) < 1e-9
    # the accumulators run on top as on any batch
    off, rec = b.segment_index()
    for u, lat in enumerate(_split(arrays)):
        sl = orc.scale_and_penalize(lat, *scales)
        a_, b_, tot, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want = orc.segment_index(sl, t, a_, b_, tot)
        got = rec[off[u]:off[u + 1]]
        assert sorted((int(r["label"]), int(r["t0"]), int(r["t1"])) for r in got) == sorted(w[:3] for w in want)
    b.close()


def test_level_path_and_stream_path_agree_and_mix(ctx):
    """The same wide lattices through the streams (a caller that fixes the lanes asks for them) and
    through the levels; and one batch that holds wide lattices and time-ordered chains."""
    wide = _wide(5, [500, 1500, 90], [3, 3, 2], 16)
    chain = klat_amd.synth(6, np.array([300, 40, 1200], np.int32), np.array([3, 2, 4], np.int32))
    scales = (1.0, 0.1, 0.0)
    res = []
    for G in (0, 4):
        b = klat_amd.Batch(ctx, wide, lanes_per_lattice=G)
        assert b.info()["level_lattices"] == (3 if G == 0 else 0)
        b.forward_backward(*scales)
        res.append(b.alpha_beta() + (b.arc_logpost(),))
        b.close()
    for x, y in zip(*res):
        fin = np.isfinite(y)
        assert np.array_equal(np.isfinite(x), fin) and np.max(np.abs(x[fin] - y[fin]), initial=0.0) < 1e-8
    # mixed batch: interleave the two kinds
    lats = []
    for a in (wide, chain):
        for u, l in enumerate(_split(a)):
            lats.append((l, a["state_time"][a["state_off"][u]:a["state_off"][u + 1]]))
    order = [0, 3, 1, 4, 2, 5]
    mixed = klat_amd.concat_lattices([lats[i][0] for i in order], times=[lats[i][1] for i in order])
    b = klat_amd.Batch(ctx, mixed)
    info = b.info()
    assert info["level_lattices"] == 3 and info["num_jobs"] >= 2
    b.forward_backward(*scales)
    alpha, beta = b.alpha_beta()
    wa, wb, wt, wpost = _oracle_batch(mixed, *scales)
    _check(alpha, wa, 1e-8), _check(beta, wb, 1e-8), _check(b.arc_logpost(), wpost, 1e-8)
    b.close()


def test_viterbi_twin_on_the_level_path(ctx):
    arrays = _wide(9, [400, 1000, 64], [3, 4, 2], 25)
    scales = (0.5, 0.1, 0.75)
    b = klat_amd.Batch(ctx, arrays)
    assert b.info()["level_lattices"] == 3
    b.viterbi(*scales)
    fw, bw, best = b.viterbi_costs()
    keep_arc, keep_final, lbeam = b.prune_mask(5.0)
    for u, lat in enumerate(_split(arrays)):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        wbeam, wfw, wbw = orc.lattice_beam(orc.scale_and_penalize(lat, *scales))
        assert np.allclose(fw[s0:s1], wfw, rtol=1e-12, atol=1e-9, equal_nan=True)
        assert np.allclose(bw[s0:s1], wbw, rtol=1e-12, atol=1e-9, equal_nan=True)
        assert abs(best[u] - wbw[0]) < 1e-9
        if np.isfinite(wbeam):
            assert abs(lbeam[u] - wbeam) < 1e-9
    assert keep_arc.sum() > 0
    b.close()


def test_the_level_path_takes_what_fits_one_round_of_waves(ctx, monkeypatch):
    """By default the level path takes the lattices that qualify only when all of them are resident
    at once; KLAT_LEVEL=1 sends them there whatever their number, KLAT_LEVEL=0 never.
    Whichever path runs, the results are the same."""
    U = 1500  # above 256 CUs x 2 waves
    arrays = _wide(11, [96] * U, [3] * U, 12)
    scales = (1.0, 0.3, 0.0)
    res = {}
    for mode in (None, "1", "0"):
        if mode is None:
            monkeypatch.delenv("KLAT_LEVEL", raising=False)
        else:
            monkeypatch.setenv("KLAT_LEVEL", mode)
        b = klat_amd.Batch(ctx, arrays)
        assert b.info()["level_lattices"] == (U if mode == "1" else 0)
        b.forward_backward(*scales)
        res[mode] = (b.alpha_beta(), b.totals()[0], b.arc_logpost())
        b.close()
    monkeypatch.delenv("KLAT_LEVEL", raising=False)
    few = _wide(11, [96] * 8, [3] * 8, 12)
    b = klat_amd.Batch(ctx, few)
    assert b.info()["level_lattices"] == 8
    b.close()
    for mode in ("1", "0"):
        _check(res[mode][0][0], res[None][0][0], 1e-9), _check(res[mode][0][1], res[None][0][1], 1e-9)
        _check(res[mode][1], res[None][1], 1e-9), _check(res[mode][2], res[None][2], 1e-9)


def test_level_path_at_its_largest_lattice(ctx):
    """8 192 states is the bound of the level path (8 B of LDS per state beside the 64-KiB ring: a
    128-KiB workgroup); just below it, and one lattice just above it that must stay with the streams."""
    arrays = _wide(3, [8100, 8300], [3, 3], 500)
    b = klat_amd.Batch(ctx, arrays)
    assert b.info()["level_lattices"] == 1
    scales = (1.0, 0.2, 0.0)
    b.forward_backward(*scales)
    alpha, beta = b.alpha_beta()
    total, tf, tb = b.totals()
    wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
    _check(alpha, wa, 1e-9), _check(beta, wb, 1e-9), _check(total, wt, 1e-9), _check(b.arc_logpost(), wpost, 1e-9)
    b.close()
