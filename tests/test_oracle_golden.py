"""Pins the CPU oracle to the reference's own known-answer vectors
before anything else is allowed to trust it."""
import math
import os

import numpy as np
import pytest

import latfmt
from oracle import oracle as orc


def _expected(golden_dir):
    exp = {}
    with open(os.path.join(golden_dir, "readme_expected.txt")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            kind, rest = line.split(None, 1)
            exp[kind] = rest.strip()
    return exp


def _fmt7(x):
    """Kaldi text output: os.precision(7) default float formatting (%g-like)."""
    s = "%.7g" % x
    return "0" if float(s) == 0 else s


def _word_lattice(golden_dir):
    (lat,) = latfmt.read_text_archive(os.path.join(golden_dir, "egs_word_lattice.ark.txt"))
    return latfmt.prepare(lat)


def test_state_times_match_readme(golden_dir):
    lat = _word_lattice(golden_dir)
    times, utt_len, warn = orc.state_times(lat)
    want = [int(v) for v in _expected(golden_dir)["state_times"].split()]
    assert times.tolist() == want
    assert utt_len == 33 and warn == 0


def test_segment_index_matches_readme(golden_dir):
    lat = _word_lattice(golden_dir)
    times, _, _ = orc.state_times(lat)
    alpha, beta, total, tf, tb = orc.alphas_betas(lat)
    assert abs(tf - tb) < 1e-12
    got = orc.segment_index(lat, times, alpha, beta, total)
    key, want = latfmt.parse_tuple_line(_expected(golden_dir)["segment"], 3)
    assert key == lat.key == "lat1"
    assert len(got) == len(want) == 10
    # exact order and 7-significant-digit text identity with the README
    for g, w in zip(got, want):
        assert g[:3] == w[:3]
        assert _fmt7(g[3]) == _fmt7(w[3]), (g, w)


def test_position_index_matches_readme(golden_dir):
    lat = _word_lattice(golden_dir)
    dlat, state_len, max_len = orc.disambiguate_length(lat)
    assert max_len == 7
    alpha, beta, total, _, _ = orc.alphas_betas(dlat)
    times, _, _ = orc.state_times(dlat)
    got = orc.position_index(dlat, times, state_len, alpha, beta, total)
    _, want = latfmt.parse_tuple_line(_expected(golden_dir)["position"], 4)
    assert len(got) == len(want) == 9
    for g, w in zip(got, want):
        assert g[:4] == w[:4]
        assert _fmt7(g[4]) == _fmt7(w[4]), (g, w)


def test_position_post_agrees_with_the_readme_position_index(golden_dir):
    """latbin/lattice-to-word-position-post.cc accumulates the same (word, position) sums as the
    position index, whose output kwsbin2/README.md:122 pins: entry n-1 of the posterior must hold
    exactly the README records with pos == n, as float32, best first."""
    lat = _word_lattice(golden_dir)
    dlat, state_len, max_len = orc.disambiguate_length(lat)
    alpha, beta, total, _, _ = orc.alphas_betas(dlat)
    post = orc.position_post(dlat, state_len, max_len, alpha, beta, total)
    _, want = latfmt.parse_tuple_line(_expected(golden_dir)["position"], 4)
    assert len(post) == max_len == 7
    for n in range(1, max_len + 1):
        ref = sorted([(w[0], w[4]) for w in want if w[1] == n], key=lambda r: (-r[1], r[0]))
        got = post[n - 1]
        assert [g[0] for g in got] == [r[0] for r in ref], (n, got, ref)
        for g, r in zip(got, ref):
            assert abs(float(g[1]) - r[1]) < 1e-6, (n, g, r)


def test_length_dist_equals_path_enumeration(golden_dir):
    """latbin/lattice-to-transcript-length-dist.cc has no known-answer vector in the reference;
    the oracle's restatement is checked against brute-force path enumeration (the reference's own
    test idiom) on a lattice with epsilon arcs and final states at several depths."""
    lat = latfmt.prepare(latfmt.read_text_archive(os.path.join(golden_dir, "multi_final_lattice.ark.txt"))[0])
    dlat, state_len, max_len = orc.disambiguate_length(lat)
    alpha, beta, total, _, _ = orc.alphas_betas(dlat)
    got = orc.length_dist(dlat, state_len, alpha, total)
    by_len = {}
    for arcs, ll in latfmt.brute_force_paths(lat):
        n = sum(1 for i in arcs if lat.olabel[i] != 0)
        by_len.setdefault(n, []).append(ll)
    tot = latfmt.logsumexp(x for v in by_len.values() for x in v)
    want = sorted(((n, latfmt.logsumexp(v) - tot) for n, v in by_len.items()), key=lambda r: (-r[1], r[0]))
    assert len(want) >= 3 and abs(tot - total) < 1e-9
    assert [g[0] for g in got] == [w[0] for w in want]
    for g, w in zip(got, want):
        assert abs(float(g[1]) - w[1]) < 1e-6


@pytest.mark.parametrize("name", ["egs_word_lattice.ark.txt", "multi_final_lattice.ark.txt"])
def test_best_path2_equals_path_enumeration(golden_dir, name):
    """latbin/lattice-best-path2.cc has no known-answer vector in the reference (parity unpinned for
    that tool): the oracle's restatement (position posteriors of the padded lattice, arc costs
    1 - P, shortest path) is checked against enumerating every path of the padded lattice."""
    lat = latfmt.prepare(latfmt.read_text_archive(os.path.join(golden_dir, name))[0])
    got, cost, _ = orc.best_path2(lat)
    dlat, state_len, _ = orc.disambiguate_length(orc.arc_sort_olabel(lat))
    plat, plen = orc.add_length_disambiguation(dlat, state_len)
    paths = latfmt.brute_force_paths(plat)
    tot = latfmt.logsumexp(ll for _, ll in paths)
    mass = {}
    for arcs, ll in paths:
        for i in arcs:
            if plat.olabel[i] != 0:
                mass.setdefault((int(plat.olabel[i]), int(plen[plat.dst[i]])), []).append(ll)
    prob = {k: min(1.0, math.exp(latfmt.logsumexp(v) - tot)) for k, v in mass.items()}
    scored = []
    for arcs, _ in paths:
        c = sum(1.0 - prob[(int(plat.olabel[i]), int(plen[plat.dst[i]]))] for i in arcs if plat.olabel[i] != 0)
        scored.append((c, [int(plat.olabel[i]) for i in arcs if plat.olabel[i] not in (0, -1)]))
    best = min(c for c, _ in scored)
    assert abs(float(cost) - best) < 1e-5
    assert got in [t for c, t in scored if c < best + 1e-5]


def test_utterance_index_matches_readme(golden_dir):
    """kwsbin2/README.md:25: P(word occurs in the transcription), 8 records, printed order."""
    lat = _word_lattice(golden_dir)
    got = orc.utterance_index(lat)
    _, want = latfmt.parse_tuple_line(_expected(golden_dir)["utterance"], 1)
    assert len(got) == len(want) == 8
    for g, w in zip(got, want):
        assert g[0] == w[0]
        assert _fmt7(g[1]) == _fmt7(w[1]) or abs(g[1] - w[1]) < 5e-8, (g, w)


def test_utterance_total_matches_readme(golden_dir):
    """kwsbin2/README.md:25 pins the betas-only total: words on every path have logp 0."""
    lat = _word_lattice(golden_dir)
    beta, total = orc.betas(lat)
    _, want = latfmt.parse_tuple_line(_expected(golden_dir)["utterance"], 1)
    # word 3 lies only on the P=0.8 branch: logp = log(0.8) - total
    alpha, beta2, tot2, _, _ = orc.alphas_betas(lat)
    assert np.allclose(beta, beta2, atol=0, rtol=0)
    assert abs(total) < 1e-7
    w3 = [w for w in want if w[0] == 3][0][1]
    assert _fmt7(math.log(0.8) - total) == _fmt7(w3) or abs(math.log(0.8) - total - w3) < 5e-8


def test_frame_post_appendix_c2(golden_dir):
    """lattice-to-word-frame-post has no reference known-answer ("parity unpinned" for this
    accumulator); SURVEY.md Appendix C.2 gives the analytically derived table."""
    lat = _word_lattice(golden_dir)
    times, T, _ = orc.state_times(lat)
    alpha, beta, total, _, _ = orc.alphas_betas(lat)
    post, mis = orc.frame_post(lat, times, T, alpha, beta, total)
    assert T == 33 and not mis and len(post) == 33
    l8, l2 = math.log(0.8), math.log(0.2)
    table = {(0, 1): [(2, l8), (1, l2)], (2, 3): [(2, l8), (4, l2)], (4, 7): [(3, l8), (4, l2)],
             (8, 8): [(5, l8), (4, l2)], (9, 11): [(5, 0.0)], (12, 15): [(2, 0.0)],
             (16, 21): [(6, 0.0)], (22, 26): [(7, 0.0)], (27, 32): [(8, 0.0)]}
    for (a, b), want in table.items():
        for f in range(a, b + 1):
            assert [p[0] for p in post[f]] == [w[0] for w in want], f
            for (gl, gp), (wl, wp) in zip(post[f], want):
                assert abs(float(gp) - wp) < 1e-6


def test_logadd_semantics():
    inf = math.inf
    assert orc.log_add(-inf, -inf) == -inf
    assert orc.log_add(-inf, 1.5) == 1.5
    assert orc.log_add(2.0, -inf) == 2.0
    assert orc.log_add(0.0, -40.0) == 0.0  # below log(DBL_EPSILON): larger one returned exactly
    assert abs(orc.log_add(0.0, 0.0) - math.log(2.0)) < 1e-15
    assert abs(orc.log_add(-3.0, -1.0) - math.log(math.exp(-3) + math.exp(-1))) < 1e-15


@pytest.mark.parametrize("name", ["egs_expand_lattice.char.txt", "egs_expand_lattice2.char.txt",
                                  "egs_char_lattice.lattice-format.ark.txt",
                                  "egs_ctc_input.lattice-format.txt", "egs_word_lattice.ark.txt"])
def test_sweep_equals_path_enumeration(golden_dir, name):
    """The reference's test idiom (brute-force path oracle) applied to the five
    shipped lattices: arc posteriors from the sweep == logsumexp over enumerated paths."""
    (lat,) = latfmt.read_text_archive(os.path.join(golden_dir, name))
    lat = latfmt.prepare(lat)
    assert orc.is_top_sorted(lat)
    alpha, beta, total, tf, tb = orc.alphas_betas(lat)
    paths = latfmt.brute_force_paths(lat)
    assert len(paths) > 0
    assert abs(latfmt.logsumexp(p[1] for p in paths) - total) < 1e-9
    through = orc.arc_through(lat, alpha, beta)
    for i in range(lat.A):
        want = latfmt.logsumexp(ll for arcs, ll in paths if i in arcs)
        if want == -math.inf:
            assert through[i] == -math.inf
        else:
            assert abs(through[i] - want) < 1e-9


def test_kbest_oracle_against_enumeration_of_all_paths():
    """The oracle's k-best dynamic programme (the checker of klat_nbest) against the reference's
    own test idiom: every path of small random DAGs (fstext/expand-subpaths-test-utils.h:10-52)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "python"))
    import klat_amd
    rng = np.random.RandomState(0)
    for trial in range(80):
        S, fan = int(rng.randint(2, 14)), int(rng.randint(1, 4))
        a = klat_amd.synth(trial, [S], [fan])
        ab = np.concatenate([[0], np.cumsum(np.bincount(a["src"], minlength=S))]).astype(np.int64)
        lat = orc.Lat(ab, a["dst"], a["ilabel"], a["olabel"], a["graph"], a["acoustic"], np.ones(len(a["dst"]), np.int32),
                      a["final_graph"], a["final_acoustic"], np.zeros(S, np.int32))
        if trial % 2:
            lat.final_graph[rng.randint(0, S)] = 0
            lat.final_acoustic[:] = lat.final_graph
        cost = (rng.choice([0.0, 0.25, 0.5, 1.0, 0.3, 0.7], size=lat.A) if trial % 3 else rng.rand(lat.A)).astype(np.float32)
        allp = sorted(orc.enumerate_paths(lat, cost), key=lambda x: x[0])
        for k in (1, 3, 100):
            got = orc.kbest_paths(lat, cost, k)
            assert [float(c) for c, _ in got] == [float(c) for c, _ in allp[:k]]
            assert len(set(tuple(p) for _, p in got)) == len(got)
            for c, p in got:
                cc = np.float32(0)
                for i in p:
                    cc = np.float32(cc + cost[i])
                assert cc == c


@pytest.mark.parametrize("name", ["egs_expand_lattice.char.txt", "egs_expand_lattice2.char.txt",
                                  "egs_char_lattice.lattice-format.ark.txt", "egs_ctc_input.lattice-format.txt",
                                  "egs_word_lattice.ark.txt", "multi_final_lattice.ark.txt"])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.3, 0.5)])
def test_frame_post_equals_path_enumeration(golden_dir, name, scales):
    """lattice-to-word-frame-post has no known-answer vector in the reference (the one accumulator whose oracle is
    unpinned), so the oracle's frame posteriors are held against something that is NOT a restatement of the tool's loop
    (latbin/lattice-to-word-frame-post.cc:93-135): the reference's own test idiom, every path of the lattice enumerated
    (fstext/expand-subpaths-test-utils.h:10-52).  P(word w at frame k) = the mass of the paths whose arc across frame
    k carries w; frames in order, entries by (logp desc, label asc) on the float values (:126-131)."""
    for lat in latfmt.read_text_archive(os.path.join(golden_dir, name)):
        lat = orc.scale_and_penalize(latfmt.prepare(lat), *scales)
        try:
            times, T, _ = orc.state_times(lat)
        except RuntimeError:
            # a character lattice whose transition-id strings give its states no consistent times (the reference's
            # CompactLatticeStateTimes asserts on it): aligned here by making every arc as long as the difference of its
            # ends' longest distances from the start state, so that the lattice's structure still serves the test
            src_of = np.repeat(np.arange(lat.S), np.diff(lat.arc_begin))
            depth = np.zeros(lat.S, np.int32)
            for i in range(lat.A):  # (top-sorted: sources ascend)
                depth[lat.dst[i]] = max(depth[lat.dst[i]], depth[src_of[i]] + 1)
            lat.dur[:] = depth[lat.dst] - depth[src_of]
            lat.final_dur[:] = 0
            fin = ~(np.isinf(lat.final_graph) & np.isinf(lat.final_acoustic))
            lat.final_dur[fin] = depth[fin].max() - depth[fin]
            times, T, _ = orc.state_times(lat)
        alpha, beta, total, _, _ = orc.alphas_betas(lat)
        post, mis = orc.frame_post(lat, times, T, alpha, beta, total)
        paths = latfmt.brute_force_paths(lat)
        assert len(paths) > 0 and len(post) == T
        tot = latfmt.logsumexp(ll for _, ll in paths)
        assert abs(tot - total) < 1e-9
        src_of = np.repeat(np.arange(lat.S), np.diff(lat.arc_begin))
        mass = [dict() for _ in range(T)]
        for arcs, ll in paths:
            for i in arcs:
                if lat.ilabel[i] == 0:
                    continue
                for k in range(int(times[src_of[i]]), min(int(times[lat.dst[i]]), T)):
                    mass[k].setdefault(int(lat.ilabel[i]), []).append(ll)
        for k in range(T):
            want = {w: latfmt.logsumexp(v) - tot for w, v in mass[k].items()}
            got = post[k]
            assert sorted(l for l, _ in got) == sorted(want)
            for l, p in got:
                assert abs(float(p) - want[l]) <= 2e-6 * max(1.0, abs(want[l])), (name, k, l, float(p), want[l])
            vals = [(-np.float32(p), l) for l, p in got]
            assert vals == sorted(vals)  # (logp desc, label asc) on the stored floats
