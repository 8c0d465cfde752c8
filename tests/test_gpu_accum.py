"""GPU parity of the accumulate stage (segment / position index, frame posteriors) through the
C ABI, against the oracle and the reference's README vectors.  Comparison rule of SURVEY.md
Appendix C.3: integer fields bit-exact, |delta logp| <= tol, order must be a valid sort under the
reference comparator (records whose logp differ by <= tol may swap)."""
import os
import sys

import numpy as np
import pytest

import latfmt
from oracle import oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def ctx():
    assert klat_amd.device_count() >= 1
    c = klat_amd.Ctx(0)
    yield c
    c.close()


@pytest.fixture
def ctx_fresh():
    c = klat_amd.Ctx(0)
    yield c
    c.close()


def _fmt7(x):
    s = "%.7g" % x
    return "0" if float(s) == 0 else s


def _split(arrays):
    """klat_batch_desc arrays -> list of orc.Lat (for the oracle)."""
    lats = []
    for u in range(len(arrays["state_off"]) - 1):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        a0, a1 = int(arrays["arc_off"][u]), int(arrays["arc_off"][u + 1])
        S = s1 - s0
        ab = np.concatenate([[0], np.cumsum(np.bincount(arrays["src"][a0:a1], minlength=S))]).astype(np.int64)
        z = np.zeros(a1 - a0, np.int32)
        lats.append(orc.Lat(ab, arrays["dst"][a0:a1].copy(), arrays["ilabel"][a0:a1].copy(),
                            arrays["olabel"][a0:a1].copy(), arrays["graph"][a0:a1].copy(),
                            arrays["acoustic"][a0:a1].copy(), z, arrays["final_graph"][s0:s1].copy(),
                            arrays["final_acoustic"][s0:s1].copy(), np.zeros(S, np.int32)))
    return lats


def _check_sorted_records(got, want, nint, tol=TOL):
    """got/want: lists of tuples (ints..., logp) for ONE lattice."""
    assert len(got) == len(want)
    gk = {g[:nint]: g[nint] for g in got}
    assert len(gk) == len(got), "duplicate keys in GPU output"
    for w in want:
        assert w[:nint] in gk, "missing record %r" % (w[:nint],)
        assert gk[w[:nint]] == w[nint] or abs(gk[w[:nint]] - w[nint]) <= tol, (w, gk[w[:nint]])  # == : both log-zero
    # valid order: logp non-increasing up to tol, ties (within tol) may permute
    for x, y in zip(got[:-1], got[1:]):
        assert x[nint] >= y[nint] - tol or x[nint] == y[nint]
        if x[nint] == y[nint]:
            assert x[:nint] < y[:nint]


def test_segment_index_readme_vector(ctx, golden_dir):
    (lat,) = latfmt.read_text_archive(os.path.join(golden_dir, "egs_word_lattice.ark.txt"))
    lat = latfmt.prepare(lat)
    times, _, _ = orc.state_times(lat)
    arrays = klat_amd.concat_lattices([lat], [times])
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward()
    off, rec = b.segment_index()
    b.close()
    want = None
    with open(os.path.join(golden_dir, "readme_expected.txt")) as f:
        for line in f:
            if line.startswith("segment "):
                _, want = latfmt.parse_tuple_line(line.split(None, 1)[1], 3)
    assert off.tolist() == [0, 10] and len(rec) == 10
    # the 10 README records: integer fields exact, 7-digit text identical per key; records whose
    # logp agree to rounding noise (the three P=0.8 arcs, ...) may swap
    got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec]
    _check_sorted_records(got, want, 3, tol=5e-7)
    gk = {g[:3]: g[3] for g in got}
    for w in want:
        assert _fmt7(gk[w[:3]]) == _fmt7(w[3]) or abs(gk[w[:3]] - w[3]) < 5e-8


def test_position_index_readme_vector(ctx, golden_dir):
    (lat,) = latfmt.read_text_archive(os.path.join(golden_dir, "egs_word_lattice.ark.txt"))
    lat = latfmt.prepare(lat)
    dlat, state_len, _ = orc.disambiguate_length(lat)  # host pre-pass a14 (oracle side in this test)
    times, _, _ = orc.state_times(dlat)
    arrays = klat_amd.concat_lattices([dlat], [times])
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward()
    off, rec = b.position_index(state_len)
    b.close()
    with open(os.path.join(golden_dir, "readme_expected.txt")) as f:
        for line in f:
            if line.startswith("position "):
                _, want = latfmt.parse_tuple_line(line.split(None, 1)[1], 4)
    assert len(rec) == 9
    got = sorted([(int(r["label"]), int(r["pos"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec])
    for g, w in zip(got, sorted(want)):
        assert g[:4] == w[:4]
        assert abs(g[4] - w[4]) < 5e-7


@pytest.mark.parametrize("G", [1, 4])
@pytest.mark.parametrize("flt", [{}, {"exclude": [3, 7, 500]}, {"include": list(range(1, 200))}])
@pytest.mark.parametrize("largest", [2700, 5000, 6000])
def test_segment_index_synthetic(ctx, G, flt, largest):
    """Every lattice goes to the kernel its arc count asks for.  largest = 2700 states (8 097
    arcs): all take the radix-sort kernel with 8 items per thread (<= 8 192 arcs); 5000 states
    (14 997 arcs): that one takes the 16-item instantiation; 6000 states (17 997 arcs): the general
    kernel, whose sort leaves the LDS tile."""
    rng = np.random.RandomState(11)
    ns = np.concatenate([[1, 2], rng.randint(3, 400, size=60), [largest]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4], size=len(ns)).astype(np.int32)
    fo[-1] = 3
    arrays = klat_amd.synth(20260004, ns, fo)
    # few distinct labels so that keys collide and runs get folded
    arrays["olabel"] = (arrays["olabel"] % 7).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    b = klat_amd.Batch(ctx, arrays, lanes_per_lattice=G)
    b.forward_backward(1.0, 0.1, 0.0)
    off, rec = b.segment_index(**flt)
    b.close()
    for u, lat in enumerate(_split(arrays)):
        sl = orc.scale_and_penalize(lat, 1.0, 0.1, 0.0)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want = orc.segment_index(sl, t, alpha, beta, total, **flt)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec[off[u]:off[u + 1]]]
        _check_sorted_records(got, want, 3)


@pytest.mark.parametrize("kernel", ["radix", "general"])
def test_segment_index_orders_near_ties_exactly(ctx, kernel, monkeypatch):
    """Records whose log-posteriors agree to 1e-6 and better: the radix kernel sorts on the upper
    half of the double and repairs runs of equal upper halves afterwards (short runs: insertion
    sort; long unsorted runs: the remaining radix passes).  The output must be ordered on ALL bits
    of its own logp, ties by (label, t0, t1) -- kwsbin2/lattice-word-index-segment.cc:98-126."""
    if kernel == "general":
        monkeypatch.setenv("KLAT_INDEX_BITONIC", "1")
    rng = np.random.RandomState(3)
    lats, times = [], []
    for n in (3000, 100, 7):
        ab = np.array([0, n, n + 1, n + 1], np.int64)
        dst = np.concatenate([np.ones(n, np.int32), [2]]).astype(np.int32)
        lab = np.concatenate([rng.permutation(n) + 1, [5000]]).astype(np.int32)
        graph = np.ones(n + 1, np.float32)
        ac = np.concatenate([rng.permutation(n) * 1e-9, [0.5]]).astype(np.float32)
        ac[:n][rng.rand(n) < 0.2] = np.float32(1e-7)  # exact ties as well
        fg, fa = np.full(3, np.inf, np.float32), np.full(3, np.inf, np.float32)
        fg[2] = fa[2] = 0.0
        lats.append(orc.Lat(ab, dst, lab, lab.copy(), graph, ac, np.zeros(n + 1, np.int32), fg, fa, np.zeros(3, np.int32)))
        times.append(np.array([0, 1, 2], np.int32))
    arrays = klat_amd.concat_lattices(lats, times)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 1.0, 0.0)
    off, rec = b.segment_index()
    b.close()
    for u, lat in enumerate(lats):
        alpha, beta, total, _, _ = orc.alphas_betas(lat)
        want = orc.segment_index(lat, times[u], alpha, beta, total)
        r = rec[off[u]:off[u + 1]]
        got = [(int(x["label"]), int(x["t0"]), int(x["t1"]), float(x["logp"])) for x in r]
        _check_sorted_records(got, want, 3)
        keys = [(-g[3],) + g[:3] for g in got]
        assert keys == sorted(keys), "lattice %d: not ordered on all bits of logp" % u
        assert len({g[3] for g in got}) > 0.7 * len(got)  # mostly distinct values, 1e-9 apart


@pytest.mark.parametrize("kernel", ["radix", "general"])
def test_position_index_synthetic(ctx, kernel, monkeypatch):
    """Both index kernels: the radix-sort one these small lattices qualify for, and the general
    (bitonic) one, forced through the library's test hook."""
    if kernel == "general":
        monkeypatch.setenv("KLAT_INDEX_BITONIC", "1")
    rng = np.random.RandomState(5)
    ns = rng.randint(2, 60, size=40).astype(np.int32)
    fo = rng.choice([1, 2, 3], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(99, ns, fo)
    arrays["olabel"] = (arrays["olabel"] % 5).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    # disambiguate every lattice on the host (oracle side), rebuild the batch
    dl, sls, tms = [], [], []
    for u, lat in enumerate(_split(arrays)):
        lat.dur[:] = (arrays["state_time"][arrays["state_off"][u] + lat.dst] -
                      arrays["state_time"][arrays["state_off"][u] + lat.src()])
        d, sl, _ = orc.disambiguate_length(lat)
        t, _, _ = orc.state_times(d)
        dl.append(d), sls.append(sl), tms.append(t)
    darr = klat_amd.concat_lattices(dl, tms)
    b = klat_amd.Batch(ctx, darr)
    b.forward_backward(1.0, 0.1, 0.0)
    off, rec = b.position_index(np.concatenate(sls))
    b.close()
    for u, d in enumerate(dl):
        sl = orc.scale_and_penalize(d, 1.0, 0.1, 0.0)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.position_index(sl, tms[u], sls[u], alpha, beta, total)
        got = [(int(r["label"]), int(r["pos"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec[off[u]:off[u + 1]]]
        # (t0,t1) is payload: it must match exactly unless two arcs tie for the maximum within tol
        assert len(got) == len(want)
        gk = {g[:2]: g for g in got}
        for w in want:
            g = gk[w[:2]]
            assert abs(g[4] - w[4]) <= TOL
            assert g[2:4] == w[2:4]
        for x, y in zip(got[:-1], got[1:]):
            assert x[4] >= y[4] - TOL


@pytest.mark.parametrize("width", [3000, 6500, 9000])
def test_position_index_wide_lattices(ctx, width):
    """Position index on lattices of 2 * width arcs (two bundles of parallel arcs, so the length
    disambiguation leaves them as they are): 6 000 arcs take the radix kernel with 8 items per
    thread, 13 000 the 16-item instantiation, 18 000 the general kernel; a small lattice rides along."""
    rng = np.random.RandomState(width)
    lats, sls, tms = [], [], []
    for n in (width, 40):
        ab = np.array([0, n, 2 * n, 2 * n], np.int64)
        dst = np.concatenate([np.ones(n, np.int32), np.full(n, 2, np.int32)]).astype(np.int32)
        lab = rng.randint(1, max(3, n // 3), size=2 * n).astype(np.int32)
        graph = (rng.rand(2 * n) * 8).astype(np.float32)
        ac = (rng.rand(2 * n) * 4).astype(np.float32)
        fg, fa = np.full(3, np.inf, np.float32), np.full(3, np.inf, np.float32)
        fg[2] = fa[2] = 0.0
        lats.append(orc.Lat(ab, dst, lab, lab.copy(), graph, ac, np.ones(2 * n, np.int32), fg, fa, np.zeros(3, np.int32)))
        sls.append(np.array([0, 1, 2], np.int32))  # words before each state
        tms.append(np.array([0, 1, 2], np.int32))
    arrays = klat_amd.concat_lattices(lats, tms)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.5, 0.0)
    off, rec = b.position_index(np.concatenate(sls))
    b.close()
    for u, lat in enumerate(lats):
        sl = orc.scale_and_penalize(lat, 1.0, 0.5, 0.0)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.position_index(sl, tms[u], sls[u], alpha, beta, total)
        got = [(int(r["label"]), int(r["pos"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec[off[u]:off[u + 1]]]
        assert len(got) == len(want)
        gk = {g[:2]: g for g in got}
        assert len(gk) == len(got)
        for w in want:
            g = gk[w[:2]]
            assert abs(g[4] - w[4]) <= TOL and g[2:4] == w[2:4]
        for x, y in zip(got[:-1], got[1:]):
            assert x[4] > y[4] or (x[4] == y[4] and x[:2] < y[:2])


def test_index_needs_state_time(ctx):
    arrays = klat_amd.synth(1, np.array([5], np.int32), np.array([2], np.int32))
    del arrays["state_time"]
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward()
    with pytest.raises(klat_amd.KlatError) as e:
        b.segment_index()
    assert e.value.code == klat_amd.ERR_INVALID_ARG
    b.close()


def _check_frames(off, labels, logps, fb, want_frames, tol=2e-6):
    for k, want in enumerate(want_frames):
        lo, hi = int(off[fb + k]), int(off[fb + k + 1])
        got = [(int(labels[i]), float(logps[i])) for i in range(lo, hi)]
        assert sorted(g[0] for g in got) == sorted(w[0] for w in want), (k, got, want)
        gk = dict(got)
        for lab, lp in want:
            assert gk[lab] == float(lp) or abs(gk[lab] - float(lp)) <= tol * max(1.0, abs(float(lp)))  # == : log-zero
        for x, y in zip(got[:-1], got[1:]):  # float logp descending, then label ascending
            assert x[1] > y[1] or (x[1] == y[1] and x[0] < y[0])


def test_frame_post_word_lattice(ctx, golden_dir):
    """No reference known-answer exists for lattice-to-word-frame-post ("parity unpinned");
    checked against the oracle and, through it, the analytic table of SURVEY.md App. C.2."""
    (lat,) = latfmt.read_text_archive(os.path.join(golden_dir, "egs_word_lattice.ark.txt"))
    lat = latfmt.prepare(lat)
    times, T, _ = orc.state_times(lat)
    arrays = klat_amd.concat_lattices([lat], [times])
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward()
    off, labels, logps = b.frame_post([T])
    b.close()
    alpha, beta, total, _, _ = orc.alphas_betas(lat)
    want, _ = orc.frame_post(lat, times, T, alpha, beta, total)
    assert len(off) == T + 1 and T == 33
    _check_frames(off, labels, logps, 0, want)
    assert [int(x) for x in labels[off[0]:off[1]]] == [2, 1]
    assert [int(x) for x in labels[off[9]:off[10]]] == [5]


@pytest.mark.parametrize("shape", ["plain", "repeated_keys", "crowded_keys", "dead_arcs", "filtered"])
def test_index_kernel_shapes(ctx, shape):
    """The hash kernel's size classes and rare paths against the oracle, segment and position index: lattices of
    1 ... 7 000 arcs (the 128-, 256- and 512-thread classes, 12 and 16 arcs a thread), keys that repeat in most arcs
    (few words, many states a time: what real word lattices look like -- the counting sort by inserter, one thread
    folding a key's arcs in arc order), hundreds of arcs a key (the fold's bitonic path), arcs of infinite cost
    (records that tie at log-zero: the out-of-line exact comparisons and a crowded last bucket) and a label filter.
    Keys and order exact (kwsbin2/lattice-word-index-segment.cc:98-126, -position.cc:102-127), logp to 1e-9."""
    rng = np.random.RandomState(21)
    ns = np.concatenate([[1, 2, 3], rng.randint(4, 700, size=14), [2000, 2300]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4], size=len(ns)).astype(np.int32)
    fo[-1] = 3
    arrays = klat_amd.synth(20260021, ns, fo)
    flt = {}
    if shape == "repeated_keys":
        arrays["olabel"] = (arrays["olabel"] % 5).astype(np.int32)
        arrays["ilabel"] = arrays["olabel"].copy()
        arrays["state_time"] = (arrays["state_time"] // 6).astype(np.int32)
    if shape == "crowded_keys":
        arrays["olabel"] = (arrays["olabel"] % 2).astype(np.int32)
        arrays["ilabel"] = arrays["olabel"].copy()
        arrays["state_time"] = (arrays["state_time"] // 400).astype(np.int32)
    if shape == "dead_arcs":
        dead = rng.rand(len(arrays["graph"])) < 0.05
        dead &= arrays["dst"] != arrays["src"] + 1
        arrays["graph"][dead] = np.inf
        arrays["acoustic"][dead] = np.inf
    if shape == "filtered":
        flt = {"exclude": [int(x) for x in rng.randint(1, 1000, size=300)]}
    # (any per-state integer serves as a length here; the state's own number keeps the maxima of a key's arcs apart:
    # arcs of one word thirteen states apart on the best path carry the same likelihood to the last bits)
    state_len = np.concatenate([np.arange(int(n), dtype=np.int32) for n in ns])
    scales = (1.0, 0.3, 0.25)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(*scales)
    (soff, srec), (poff, prec) = b.segment_index(**flt), b.position_index(state_len, **flt)
    b.close()
    assert soff[-1] > 0
    for u, lat in enumerate(_split(arrays)):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        sl = orc.scale_and_penalize(lat, *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.segment_index(sl, arrays["state_time"][s0:s1], alpha, beta, total, **flt)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in srec[soff[u]:soff[u + 1]]]
        _check_sorted_records(got, want, 3, tol=1e-9)
        want = orc.position_index(sl, arrays["state_time"][s0:s1], state_len[s0:s1], alpha, beta, total, **flt)
        got = [(int(r["label"]), int(r["pos"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in prec[poff[u]:poff[u + 1]]]
        assert len(got) == len(want)
        gk = {g[:2]: g for g in got}
        assert len(gk) == len(got)
        for w in want:  # (t0, t1) is payload: the first arc attaining the maximum, strict '>' (-position.cc:178)
            g = gk[w[:2]]
            assert g[4] == w[4] or abs(g[4] - w[4]) <= 1e-9
            assert g[2:4] == w[2:4]
        for x, y in zip(got[:-1], got[1:]):
            assert x[4] >= y[4] - 1e-9 or x[4] == y[4]
            if x[4] == y[4]:
                assert x[:2] < y[:2]


@pytest.mark.parametrize("shape", ["plain", "repeated_keys", "long_ties", "descending", "disorder"])
def test_index_of_long_lattices(ctx, shape):
    """Lattices of more than 16 384 arcs: the hash kernel takes them in pieces cut where the source states' time (the
    length, for the position index) changes -- no key repeats across such a cut -- and the gather merges the pieces'
    sorted runs.  20 000 and 60 000 arcs beside short lattices, against the oracle; keys that repeat (several states a
    time: the cuts must find the changes); a stretch of 3 000 arcs of one time (no cut inside a window: that lattice takes
    the general kernel), two neighbouring states out of order (still cut: every arc before a cut is earlier than every arc
    behind it) and a late state early in the lattice (no cut for 9 000 states: the general kernel): the oracle's records."""
    ns = np.array([150, 6700, 40, 10000, 900], np.int32)
    fo = np.array([3, 3, 2, 6, 4], np.int32)
    arrays = klat_amd.synth(20260031, ns, fo)
    if shape == "repeated_keys":
        arrays["olabel"] = (arrays["olabel"] % 5).astype(np.int32)
        arrays["ilabel"] = arrays["olabel"].copy()
        arrays["state_time"] = (arrays["state_time"] // 6).astype(np.int32)
    if shape == "long_ties":
        t = arrays["state_time"].copy()
        s0 = int(arrays["state_off"][3])
        t[s0 + 2000:s0 + 2600] = t[s0 + 2000]  # 600 states x 6 arcs of one time around the first cut of the 60 000-arc lattice
        arrays["state_time"] = t
    if shape == "descending":  # two neighbours out of order: a cut is still good wherever all before lie below all behind
        t = arrays["state_time"].copy()
        s0 = int(arrays["state_off"][1])
        t[s0 + 3000], t[s0 + 3001] = t[s0 + 3001], t[s0 + 3000]
        arrays["state_time"] = t
    if shape == "disorder":  # a late state early on: nothing before state 9 000 of the 60 000-arc lattice can be cut
        t = arrays["state_time"].copy()
        s0 = int(arrays["state_off"][3])
        t[s0 + 10] = t[s0 + 9000]
        arrays["state_time"] = t
    state_len = np.concatenate([np.arange(int(n), dtype=np.int32) // 3 for n in ns])
    scales = (1.0, 0.3, 0.25)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(*scales)
    (soff, srec), (poff, prec) = b.segment_index(), b.position_index(state_len)
    b.close()
    for u, lat in enumerate(_split(arrays)):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        sl = orc.scale_and_penalize(lat, *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.segment_index(sl, arrays["state_time"][s0:s1], alpha, beta, total)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in srec[soff[u]:soff[u + 1]]]
        _check_sorted_records(got, want, 3, tol=1e-9)
        want = orc.position_index(sl, arrays["state_time"][s0:s1], state_len[s0:s1], alpha, beta, total)
        got = [(int(r["label"]), int(r["pos"]), float(r["logp"])) for r in prec[poff[u]:poff[u + 1]]]
        _check_sorted_records(got, [(w[0], w[1], w[4]) for w in want], 2, tol=1e-9)


def test_index_of_lattices_of_many_pieces(ctx):
    """300 000 arcs: 23 pieces through the gather's banded merge; 450 000 arcs: 34 pieces, more runs than that merge keeps
    in LDS -- its plain merge (a search of every other run per record); against the oracle, beside a lattice of five pieces
    and a short one."""
    ns = np.array([30000, 200, 45000, 10000], np.int32)
    fo = np.array([10, 3, 10, 6], np.int32)
    arrays = klat_amd.synth(20260032, ns, fo)
    arrays["olabel"] = (1 + arrays["olabel"] % 300).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    scales = (1.0, 0.3, 0.0)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(*scales)
    soff, srec = b.segment_index()
    b.close()
    for u, lat in enumerate(_split(arrays)):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        sl = orc.scale_and_penalize(lat, *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.segment_index(sl, arrays["state_time"][s0:s1], alpha, beta, total)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in srec[soff[u]:soff[u + 1]]]
        _check_sorted_records(got, want, 3, tol=1e-9)


@pytest.mark.parametrize("how", ["generic", "tools"])
@pytest.mark.parametrize("shape", ["C2", "C4"])
def test_segment_index_full_size_properties(ctx, shape, how):
    """The segment index at BASELINE's sizes -- configs[1] whole (10 000 lattices x 2 000 states: the 512 x 12 class)
    and 2 000 lattices of configs[3]'s shape (5 000 states, 14 997 arcs: the 1 024 x 15 class) -- through what does
    not depend on the size: one record per distinct (lattice, label, t0, t1) of the word arcs, no more and no fewer;
    every lattice's records in descending order of logp, equal values by ascending key
    (kwsbin2/lattice-word-index-segment.cc:98-126); and folding keeps the mass: the records' exp(logp) add up to the word
    arcs' posteriors, lattice by lattice (LogAdd of a key's arcs, :165-171).  `tools` is the instantiation the tools
    run (host/bin/sharded-index.h): scales fused into the streams, sweep without the arc-posterior pass -- the index forms
    the through-likelihoods from alpha / beta itself; the first lattices are checked against the oracle record by record."""
    U, S = (10000, 2000) if shape == "C2" else (2000, 5000)
    arrays = klat_amd.synth(20260001 if shape == "C2" else 20260003, np.full(U, S, np.int32), np.full(U, 3, np.int32))
    scales = (1.0, 0.1, 0.0)
    b = klat_amd.Batch(ctx, arrays, fuse_scales=scales if how == "tools" else None)
    b.forward_backward(*scales, arc_posteriors=how != "tools")
    off, rec = b.segment_index()
    post = b.arc_logpost()  # (tools: the posterior pass runs now, after the index, for the mass check below)
    b.close()
    for u in range(4 if shape == "C2" else 2):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        a0, a1 = int(arrays["arc_off"][u]), int(arrays["arc_off"][u + 1])
        one = {k: (arrays[k][a0:a1] if len(arrays[k]) == len(arrays["src"]) else arrays[k][s0:s1]) for k in arrays if k not in ("state_off", "arc_off")}
        one["state_off"], one["arc_off"] = np.array([0, s1 - s0], np.int64), np.array([0, a1 - a0], np.int64)
        sl = orc.scale_and_penalize(_split(one)[0], *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.segment_index(sl, arrays["state_time"][s0:s1], alpha, beta, total)
        got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in rec[off[u]:off[u + 1]]]
        _check_sorted_records(got, want, 3, tol=1e-8)
    lat_of_arc = np.repeat(np.arange(U, dtype=np.int64), np.diff(arrays["arc_off"]))
    word = arrays["olabel"] != 0
    gsrc = arrays["src"].astype(np.int64) + arrays["state_off"][lat_of_arc]
    gdst = arrays["dst"].astype(np.int64) + arrays["state_off"][lat_of_arc]
    t0, t1 = arrays["state_time"][gsrc].astype(np.int64), arrays["state_time"][gdst].astype(np.int64)
    assert int(t1.max()) < (1 << 15) and int(arrays["olabel"].max()) < (1 << 11) and U < (1 << 14)
    key = (((lat_of_arc << 11 | arrays["olabel"].astype(np.int64)) << 15 | t0) << 15 | t1)[word]
    want_keys = np.unique(key)
    lat_of_rec = np.repeat(np.arange(U, dtype=np.int64), np.diff(off))
    got_keys = ((lat_of_rec << 11 | rec["label"].astype(np.int64)) << 15 | rec["t0"].astype(np.int64)) << 15 | rec["t1"].astype(np.int64)
    assert len(got_keys) == len(want_keys) and np.array_equal(np.sort(got_keys), want_keys)
    lp = rec["logp"]
    inside = lat_of_rec[1:] == lat_of_rec[:-1]
    assert np.all(lp[:-1][inside] >= lp[1:][inside])
    tie = inside & (lp[:-1] == lp[1:])
    assert np.all(got_keys[:-1][tie] < got_keys[1:][tie])
    mass_arcs = np.bincount(lat_of_arc[word], weights=np.exp(post[word]), minlength=U)
    mass_recs = np.bincount(lat_of_rec, weights=np.exp(lp), minlength=U)
    assert np.max(np.abs(mass_arcs - mass_recs) / np.maximum(mass_arcs, 1e-300)) < 1e-9


# the frame kernels: the lean one with the general one behind it (default), the general one alone, and a record
# pool that holds the first few lattices only (the others go through the count pass and the general kernel)
FRAME_MODES = {"lean": {}, "general": {"KLAT_FRAME_GENERAL": "1"}, "short_pool": {"KLAT_FRAME_POOL_RATIO": "0.2"}}


@pytest.fixture(params=sorted(FRAME_MODES))
def frame_mode(request, monkeypatch):
    for k, v in FRAME_MODES[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (1.0, 0.1, 0.5)])
def test_frame_post_synthetic(ctx, scales, frame_mode):
    rng = np.random.RandomState(3)
    ns = np.concatenate([[1, 2], rng.randint(3, 300, size=30), [1500]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(20260004, ns, fo)
    arrays["olabel"] = (arrays["olabel"] % 9).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    dead = rng.rand(len(arrays["graph"])) < 0.03  # arcs of infinite cost: log-zero entries, not NaN
    dead &= arrays["dst"] != arrays["src"] + 1    # (keep every lattice connected)
    arrays["graph"][dead] = np.inf
    arrays["acoustic"][dead] = np.inf
    lats = _split(arrays)
    nframes = [int(arrays["state_time"][arrays["state_off"][u + 1] - 1]) for u in range(len(ns))]
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(*scales)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb = 0
    for u, lat in enumerate(lats):
        sl = orc.scale_and_penalize(lat, *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, mis = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


@pytest.mark.parametrize("words", [7, 400])
def test_frame_post_of_dense_lattices(ctx, words):
    """Lattices whose frames hold from a few to two hundred arcs (fanout 1 ... 40), of few distinct words (runs to fold in
    every frame, across a wave's lanes and registers) or of many: the thread-per-rank path with 8 / 12 / 16 registers, the
    dense pass with 16 / 24 / 32, the wave path with one, two and four entries a lane, and tiles of fewer than 128 ranks --
    every lattice against the oracle (latbin/lattice-to-word-frame-post.cc:93-135)."""
    rng = np.random.RandomState(words)
    fo = np.array([1, 2, 3, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 3, 6, 12, 24, 40], np.int32)
    ns = np.concatenate([rng.randint(40, 200, size=13), [900, 700, 500, 300, 250]]).astype(np.int32)
    arrays = klat_amd.synth(20260070 + words, ns, fo)
    arrays["olabel"] = (1 + arrays["olabel"] % words).astype(np.int32)
    arrays["olabel"][rng.rand(len(arrays["olabel"])) < 0.04] = 0  # epsilon arcs cross no frame
    arrays["ilabel"] = arrays["olabel"].copy()
    dead = rng.rand(len(arrays["graph"])) < 0.02
    dead &= arrays["dst"] != arrays["src"] + 1  # (keep every lattice connected)
    arrays["graph"][dead] = np.inf
    arrays["acoustic"][dead] = np.inf
    lats = _split(arrays)
    nframes = [int(arrays["state_time"][arrays["state_off"][u + 1] - 1]) for u in range(len(ns))]
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.3, 0.1)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb, densest = 0, 0
    for u, lat in enumerate(lats):
        sl = orc.scale_and_penalize(lat, 1.0, 0.3, 0.1)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, mis = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


def test_frame_post_of_many_dense_lattices(ctx):
    """A batch whose dense pass has more than 2 048 lattices takes that pass in workgroups of 256 threads (two lattices a
    CU; a smaller batch: one workgroup of 512): every fortieth lattice against the oracle."""
    rng = np.random.RandomState(11)
    U = 2400
    ns = rng.randint(30, 70, size=U).astype(np.int32)
    fo = rng.choice([6, 8, 10], size=U).astype(np.int32)
    arrays = klat_amd.synth(20260081, ns, fo)
    arrays["olabel"] = (1 + arrays["olabel"] % 50).astype(np.int32)
    arrays["ilabel"] = arrays["olabel"].copy()
    lats = _split(arrays)
    nframes = [int(arrays["state_time"][arrays["state_off"][u + 1] - 1]) for u in range(U)]
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.2, 0.0)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    first = np.concatenate([[0], np.cumsum(nframes)])
    for u in range(0, U, 40):
        sl = orc.scale_and_penalize(lats[u], 1.0, 0.2, 0.0)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, mis = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, int(first[u]), want)
    assert off[-1] == len(labels)


def test_frame_post_pool_follows_the_batches(ctx_fresh):
    """The lean kernel's record pool is sized by what the context's earlier batches asked for: a batch of long arcs
    (thirty ranks an arc: far beyond the four records per arc a new context starts with) goes to the general
    kernel the first time and through the enlarged pool the second time, and an ordinary batch afterwards brings
    the pool back down; all four calls give the oracle's posteriors."""
    rng = np.random.RandomState(5)
    ns = np.concatenate([[1, 2], rng.randint(100, 500, size=12)]).astype(np.int32)
    fo = np.full(len(ns), 3, np.int32)
    plain = klat_amd.synth(20260009, ns, fo)
    plain["olabel"] = (plain["olabel"] % 7).astype(np.int32)
    plain["ilabel"] = plain["olabel"].copy()
    long_arcs = {k: v.copy() for k, v in plain.items()}
    S_of = np.repeat(ns, np.diff(plain["arc_off"]))
    src, dst = long_arcs["src"], long_arcs["dst"]
    far = dst != src + 1  # (the arcs to the next state keep every lattice connected)
    dst[far] = np.minimum(src[far] + 60, S_of[far] - 1)
    for arrays in (long_arcs, long_arcs, plain, plain):
        nframes = [int(arrays["state_time"][arrays["state_off"][u + 1] - 1]) for u in range(len(ns))]
        b = klat_amd.Batch(ctx_fresh, arrays)
        b.forward_backward(1.0, 0.2, 0.0)
        off, labels, logps = b.frame_post(nframes)
        b.close()
        fb = 0
        for u, lat in enumerate(_split(arrays)):
            sl = orc.scale_and_penalize(lat, 1.0, 0.2, 0.0)
            alpha, beta, total, _, _ = orc.alphas_betas(sl)
            t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
            want, _ = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
            _check_frames(off, labels, logps, fb, want)
            fb += nframes[u]
        assert off[-1] == len(labels)


@pytest.mark.parametrize("shape", ["plain", "few_words", "descending", "disorder", "long_arc"])
def test_frame_post_of_long_lattices(ctx, shape):
    """Lattices beyond the lean kernel's 8 192 states or 65 535 arcs are cut into pieces of ~6 144 states on the device --
    a piece owns the frames between two cuts and takes every arc that can cross them, its times clipped to the window --
    and the pieces run through the lean kernel's second instantiation: 9 000 states x 3 (27 000 arcs) and 12 000 states x 6
    (72 000 arcs) beside short lattices, against the oracle frame by frame; few distinct words (every frame folds); times
    that descend somewhere and an arc longer than a piece's margin (those lattices take the general kernel)."""
    ns = np.array([150, 9000, 40, 12000, 900], np.int32)
    fo = np.array([3, 3, 2, 6, 4], np.int32)
    arrays = klat_amd.synth(20260041, ns, fo)
    if shape == "few_words":
        arrays["olabel"] = (arrays["olabel"] % 4).astype(np.int32)
        arrays["ilabel"] = arrays["olabel"].copy()
    if shape == "descending":
        t = arrays["state_time"].copy()
        s0 = int(arrays["state_off"][1])
        t[s0 + 7000], t[s0 + 7001] = t[s0 + 7001], t[s0 + 7000]
        arrays["state_time"] = t
    if shape == "disorder":  # a late state early on: no cut before state 11 000 of the 12 000-state lattice
        t = arrays["state_time"].copy()
        s0 = int(arrays["state_off"][3])
        t[s0 + 10] = t[s0 + 11000]
        arrays["state_time"] = t
    if shape == "long_arc":
        a0 = int(arrays["arc_off"][3])
        k = a0 + int(np.argmax((arrays["src"][a0:] == 5000) & (arrays["dst"][a0:] != 5001)))
        arrays["dst"][k] = 7000  # spans 2 000 states: more than a piece's margin of 1 024
    scales = (1.0, 0.2, 0.0)
    nframes = [int(arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]].max()) for u in range(len(ns))]
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(*scales)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb = 0
    for u, lat in enumerate(_split(arrays)):
        sl = orc.scale_and_penalize(lat, *scales)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, _ = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


@pytest.mark.parametrize("times", ["ties", "clipped", "descending", "jitter"])
def test_frame_post_state_times(ctx, times, frame_mode):
    """The lean kernel works over the distinct state times: states that share a time share a rank, times beyond
    num_frames (and below 0) are clipped as latbin/lattice-to-word-frame-post.cc:96-99 clips an arc's frames, and
    a lattice whose times descend somewhere gets its ranks from a bit per frame (and looks at all arcs per tile)."""
    rng = np.random.RandomState(11)
    ns = np.concatenate([[1, 2, 3], rng.randint(4, 400, size=20)]).astype(np.int32)
    fo = rng.choice([1, 2, 3], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(77, ns, fo)
    arrays["ilabel"] = (arrays["ilabel"] % 40).astype(np.int32)
    arrays["olabel"] = arrays["ilabel"].copy()
    t = arrays["state_time"].copy()
    nframes = []
    for u in range(len(ns)):
        lo, hi = arrays["state_off"][u], arrays["state_off"][u + 1]
        if times == "ties":  # runs of states at one time, the first states among them
            t[lo:hi] = (t[lo:hi] // 5) * 5
            t[lo:hi] -= t[lo]
        elif times == "clipped":  # some times below 0, the last states beyond num_frames
            t[lo:hi] -= 3
        elif times == "descending" and hi - lo > 3:
            t[lo + 2] = t[lo + 1] - 1 if u % 2 else t[lo + 2]
        elif times == "jitter":  # any order: a top-sorted lattice's times need not ascend with the state number
            t[lo:hi] = np.maximum(t[lo:hi] + rng.randint(-6, 7, size=hi - lo), 0)
        last = int(t[hi - 1])
        nframes.append(max(0, last - (4 if times == "clipped" else 0)))
    arrays["state_time"] = t.astype(np.int32)
    lats = _split(arrays)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward()
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb = 0
    for u, lat in enumerate(lats):
        alpha, beta, total, _, _ = orc.alphas_betas(lat)
        tt = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, _ = orc.frame_post(lat, tt, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


def test_frame_post_kernels_agree_on_a_mix(ctx, monkeypatch):
    """The two instantiations of the lean kernel and the general kernel against one another on a batch too large
    for the CPU oracle in a test: lattices of 2 .. 9 000 states (the larger ones keep their spans in L2), jittered
    and tied state times, few labels (every frame folds), frames of 17 .. 40 arcs (a wave each).  Same frames, same
    labels in the same order up to ties of the float posterior, posteriors within the tolerance of the oracle tests."""
    rng = np.random.RandomState(5)
    ns = np.concatenate([rng.randint(2, 600, size=60), rng.randint(2500, 9000, size=12), [1, 2]]).astype(np.int32)
    fo = rng.choice([2, 3, 4], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(4242, ns, fo)
    arrays["ilabel"] = (arrays["ilabel"] % 13).astype(np.int32)
    arrays["olabel"] = arrays["ilabel"].copy()
    t = arrays["state_time"].copy()
    for u in range(len(ns)):
        lo, hi = arrays["state_off"][u], arrays["state_off"][u + 1]
        if u % 3 == 1:
            t[lo:hi] = np.maximum(t[lo:hi] + rng.randint(-5, 6, size=hi - lo), 0)
        elif u % 3 == 2:
            t[lo:hi] = (t[lo:hi] // 3) * 3
    arrays["state_time"] = t.astype(np.int32)
    nframes = [max(0, int(t[arrays["state_off"][u + 1] - 1])) for u in range(len(ns))]
    got = {}
    for mode in ("lean", "general"):
        for k, v in FRAME_MODES[mode].items():
            monkeypatch.setenv(k, v)
        b = klat_amd.Batch(ctx, arrays)
        b.forward_backward(1.0, 0.3, 0.0)
        got[mode] = b.frame_post(nframes)
        b.close()
        for k in FRAME_MODES[mode]:
            monkeypatch.delenv(k)
    (o0, l0, p0), (o1, l1, p1) = got["lean"], got["general"]
    assert np.array_equal(o0, o1) and len(l0) == len(l1) and len(l0) > 100000
    close = np.isclose(p0, p1, rtol=2e-6, atol=2e-6) | (p0 == p1)
    assert close.all(), int((~close).sum())
    diff = np.nonzero(l0 != l1)[0]  # labels may swap only where the float posteriors tie or nearly tie inside a frame
    for i in diff[:200]:
        f = np.searchsorted(o0, i, side="right") - 1
        assert sorted(l0[o0[f]:o0[f + 1]].tolist()) == sorted(l1[o1[f]:o1[f + 1]].tolist()), (int(i), int(f))


@pytest.mark.parametrize("width", [40, 100, 128, 135, 200, 256, 270, 700, 5000])
def test_frame_post_wide_frames(ctx, width):
    """Frames crossed by many arcs.  The lean kernel: up to 16 (dense pass: 32) entries in a thread's registers, up to 64
    one a lane of a wave, up to 128 two and up to 256 four a lane (100 ... 270: runs of one word cross from lane 63's
    register into lane 0's next).  Beyond that the general kernel: up to 256 in place, above that the whole workgroup sorts the
    frame's bucket (700: through the LDS tile; 5000: beyond it).  Labels repeat inside a frame, so runs get folded; some
    arcs are dead."""
    rng = np.random.RandomState(width)
    lats, times, nframes = [], [], []
    for n, dur in ((width, 3), (width // 3 + 1, 2), (5, 4)):
        ab = np.array([0, n, n + 1, n + 1], np.int64)
        dst = np.concatenate([np.ones(n, np.int32), [2]]).astype(np.int32)
        lab = np.concatenate([rng.randint(1, max(2, n // 2), size=n), [7]]).astype(np.int32)
        lab[rng.rand(n + 1) < 0.05] = 0  # epsilon arcs are skipped
        graph = np.concatenate([rng.rand(n) * 6, [0.25]]).astype(np.float32)
        ac = np.concatenate([rng.rand(n) * 3, [0.5]]).astype(np.float32)
        dead = rng.rand(n + 1) < 0.03
        dead[0] = dead[n] = False
        graph[dead] = np.inf
        ac[dead] = np.inf
        fg, fa = np.full(3, np.inf, np.float32), np.full(3, np.inf, np.float32)
        fg[2] = fa[2] = 0.0
        lats.append(orc.Lat(ab, dst, lab, lab.copy(), graph, ac, np.zeros(n + 1, np.int32), fg, fa, np.zeros(3, np.int32)))
        times.append(np.array([0, dur, dur + 2], np.int32))
        nframes.append(dur + 2)
    arrays = klat_amd.concat_lattices(lats, times)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.5, 0.25)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb = 0
    for u, lat in enumerate(lats):
        sl = orc.scale_and_penalize(lat, 1.0, 0.5, 0.25)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want, mis = orc.frame_post(sl, times[u], nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


def test_frames_without_an_arc_boundary_share_their_records(ctx):
    """latbin/lattice-to-word-frame-post.cc:93-117 adds an arc to every frame of its span, so two
    consecutive frames with no arc starting or ending between them hold the same posteriors: the
    kernel stores them once (klat_frame_stored_records) and the fetch expands them.  Stretching
    the state times by 3 must give every frame's entries three times over and (nearly) the same
    stored records; the dense output is checked against the oracle either way."""
    ns = np.array([40, 300, 1], np.int32)
    fo = np.array([3, 2, 1], np.int32)
    arrays = klat_amd.synth(31, ns, fo)
    res = []
    for stretch in (1, 3):
        arr = dict(arrays)
        arr["state_time"] = (arrays["state_time"] * stretch).astype(np.int32)
        nframes = [int(arr["state_time"][arr["state_off"][u + 1] - 1]) for u in range(len(ns))]
        b = klat_amd.Batch(ctx, arr)
        b.forward_backward()
        off, labels, logps = b.frame_post(nframes)
        res.append((len(labels), b.frame_stored_records()))
        b.close()
        fb = 0
        for u, lat in enumerate(_split(arr)):
            alpha, beta, total, _, _ = orc.alphas_betas(lat)
            t = arr["state_time"][arr["state_off"][u]:arr["state_off"][u + 1]]
            want, _ = orc.frame_post(lat, t, nframes[u], alpha, beta, total)
            _check_frames(off, labels, logps, fb, want)
            fb += nframes[u]
    # (a tile of 1 920 frames starts with a stored frame: the stretched run has a few more of those)
    assert res[1][0] == 3 * res[0][0] and res[0][1] <= res[1][1] <= res[0][1] + 64 and res[0][1] <= res[0][0]


@pytest.mark.parametrize("case", ["many_arcs", "many_frames"])
def test_frame_post_beyond_the_lds_spans(ctx, case):
    """The frame kernel keeps an arc's clipped span in LDS as two 16-bit frame numbers when the
    lattice has at most 6 144 arcs and 65 535 frames; beyond either bound the spans stay in the
    arc's scratch row.  Both bounds are crossed here, next to a small lattice in the same batch."""
    if case == "many_arcs":
        ns, fo, stretch = np.array([2600, 50], np.int32), np.array([3, 2], np.int32), 1
    else:
        ns, fo, stretch = np.array([300, 50], np.int32), np.array([3, 2], np.int32), 100
    arrays = klat_amd.synth(47, ns, fo)
    arrays["state_time"] = (arrays["state_time"] * stretch).astype(np.int32)
    nframes = [int(arrays["state_time"][arrays["state_off"][u + 1] - 1]) for u in range(len(ns))]
    assert (case == "many_arcs" and np.diff(arrays["arc_off"])[0] > 6144) or (case == "many_frames" and nframes[0] > 65535)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.2, 0.0)
    off, labels, logps = b.frame_post(nframes)
    b.close()
    fb = 0
    for u, lat in enumerate(_split(arrays)):
        sl = orc.scale_and_penalize(lat, 1.0, 0.2, 0.0)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        want, _ = orc.frame_post(sl, t, nframes[u], alpha, beta, total)
        _check_frames(off, labels, logps, fb, want)
        fb += nframes[u]
    assert off[-1] == len(labels)


def test_accumulators_need_no_posterior_pass(ctx):
    """The tools sweep with KLAT_FB_NO_ARC_POSTERIORS: the accumulators form fw[src] + arc_lkh + bw[dst]
    from alpha / beta themselves (kwsbin2/lattice-word-index-segment.cc:160-163) and subtract the total at
    record time (:98-108).  Same records, bit for bit, as after a sweep with the posterior pass; and
    klat_get_arc_logpost still answers afterwards (the pass runs on demand)."""
    rng = np.random.RandomState(21)
    ns = rng.randint(2, 300, size=30).astype(np.int32)
    fo = rng.choice([1, 2, 3, 5], size=30).astype(np.int32)
    arrays = klat_amd.synth(99, ns, fo)
    nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
    out = []
    for arc_posteriors in (True, False):
        b = klat_amd.Batch(ctx, arrays)
        b.forward_backward(1.0, 0.1, 0.5, arc_posteriors=arc_posteriors)
        off, rec = b.segment_index()
        foff, lab, lp = b.frame_post(nf)
        post = b.arc_logpost()
        tot = b.totals()
        out.append((off, rec.tobytes(), foff, lab, lp.tobytes(), post, tot))
        b.close()
    (o0, r0, f0, l0, p0, a0, t0), (o1, r1, f1, l1, p1, a1, t1) = out
    assert np.array_equal(o0, o1) and r0 == r1
    assert np.array_equal(f0, f1) and np.array_equal(l0, l1) and p0 == p1
    assert np.array_equal(a0, a1) and all(np.array_equal(x, y) for x, y in zip(t0, t1))


def _log_add(x, y):
    if x < y:
        x, y = y, x
    d = y - x
    return x + np.log1p(np.exp(d)) if d >= np.log(np.finfo(np.float64).eps) else x


@pytest.mark.parametrize("source", ["egs", "synthetic"])
def test_frame_post_is_the_segment_index_summed_over_frames(ctx, golden_dir, source):
    """lattice-to-word-frame-post has no known-answer vector in the reference; the segment index has
    (kwsbin2/README.md:75).  The two accumulate the same quantity: an arc (w, t0, t1) adds its likelihood
    to every frame t0 <= k < t1 (latbin/lattice-to-word-frame-post.cc:110-116) and to the record (w, t0, t1)
    (kwsbin2/lattice-word-index-segment.cc:165-171).  So the frame posterior of (k, w) must be the LogAdd
    of the segment records (w, t0, t1) with t0 <= k < t1, narrowed to float -- the unpinned output checked
    against the family of outputs the README pins.  (The frame tool forms the arc cost as a float sum and
    LogAdds in arc order, the index as doubles per segment: a few float32 ulps.)"""
    if source == "egs":  # every shipped example lattice, the README's among them
        lats, times = [], []
        for name in sorted(os.listdir(golden_dir)):
            if not name.startswith("egs_") or not name.endswith(".txt"):
                continue
            for lat in latfmt.read_text_archive(os.path.join(golden_dir, name)):
                lat = latfmt.prepare(lat)
                try:
                    t = orc.state_times(lat)[0]
                except RuntimeError:  # (the character lattices carry no consistent frame times: not inputs of these tools)
                    continue
                lats.append(lat)
                times.append(t)
        assert len(lats) >= 2
        arrays = klat_amd.concat_lattices(lats, times)
    else:
        rng = np.random.RandomState(33)
        ns = rng.randint(2, 120, size=20).astype(np.int32)
        fo = rng.choice([1, 2, 3, 4], size=20).astype(np.int32)
        arrays = klat_amd.synth(1234, ns, fo)
    U = len(arrays["state_off"]) - 1
    nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.25, 0.0, arc_posteriors=False)
    off, rec = b.segment_index()
    foff, lab, lp = b.frame_post(nf)
    b.close()
    fb = np.concatenate([[0], np.cumsum(nf)])
    checked = 0
    for u in range(U):
        want = [dict() for _ in range(int(nf[u]))]
        for r in rec[off[u]:off[u + 1]]:
            for k in range(max(int(r["t0"]), 0), min(int(r["t1"]), int(nf[u]))):
                w = int(r["label"])
                want[k][w] = _log_add(want[k][w], float(r["logp"])) if w in want[k] else float(r["logp"])
        for k in range(int(nf[u])):
            e0, e1 = int(foff[fb[u] + k]), int(foff[fb[u] + k + 1])
            got = {int(lab[i]): float(lp[i]) for i in range(e0, e1)}
            assert sorted(got) == sorted(want[k]), (u, k)
            for w, v in got.items():
                assert abs(v - want[k][w]) <= 2e-5 * max(1.0, abs(v)), (u, k, w, v, want[k][w])
                checked += 1
            vals = [float(lp[i]) for i in range(e0, e1)]
            assert all(vals[i] >= vals[i + 1] for i in range(len(vals) - 1))
    assert checked > 0
