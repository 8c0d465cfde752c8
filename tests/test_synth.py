"""The synthetic lattice generators (bench / test workloads): an independent pure-Python
restatement of SURVEY.md Appendix D and of this build's levelled ("wide") shape that must agree
bit for bit with the C++ generator in the library (csrc/klat_synth.cc), plus the structural
properties the bench relies on (top-sorted, connected, consistent state times)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15


class SplitMix64:
    def __init__(self, seed):
        self.x = seed & M64

    def next(self):
        self.x = (self.x + GOLD) & M64
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def _arc_draws(rng):
    r = rng.next()
    label = 0 if r % 20 == 0 else 1 + rng.next() % 1000
    g = f32((rng.next() >> 11) * 2.0 ** -53 * 8.0)
    a = f32((rng.next() >> 11) * 2.0 ** -53 * 40.0)
    return label, g, a


def appendix_d(seed, utt, S, fan):
    """SURVEY.md Appendix D, one lattice: returns (times, finals, arcs[(src,dst,label,graph,acoustic)])."""
    rng = SplitMix64(seed ^ ((GOLD * (utt + 1)) & M64))
    times = [0]
    for s in range(1, S):
        times.append(times[-1] + 1 + rng.next() % 4)
    arcs = []
    for s in range(S - 1):
        for j in range(fan):
            span = min(7, S - 2 - s)
            d = s + 1 if j == 0 else s + 1 + rng.next() % (span + 1)
            arcs.append((s, d) + _arc_draws(rng))
    finals = [S - 1]
    return times, finals, arcs


def wide(seed, utt, S, fan, w):
    """The levelled shape (csrc/klat_synth.cc header comment), one lattice."""
    rng = SplitMix64(seed ^ ((GOLD * (utt + 1)) & M64))
    level = lambda s: 0 if s == 0 else 1 + (s - 1) // w
    first = lambda l: 0 if l == 0 else 1 + (l - 1) * w
    count = lambda l: 1 if l == 0 else min(w, S - first(l))
    L = 0 if S <= 1 else level(S - 1)
    tl = [0]
    for l in range(1, L + 1):
        tl.append(tl[-1] + 1 + rng.next() % 4)
    times = [tl[level(s)] for s in range(S)]
    arcs = []
    if S >= 2:
        for k in range(count(1)):
            arcs.append((0, 1 + k) + _arc_draws(rng))
    for s in range(1, S):
        l = level(s)
        if l >= L:
            break
        nf, nc, col = first(l + 1), count(l + 1), s - first(l)
        for j in range(fan):
            d = nf + min(col, nc - 1) if j == 0 else nf + rng.next() % nc
            arcs.append((s, d) + _arc_draws(rng))
    finals = [s for s in range(S) if level(s) == L]
    return times, finals, arcs


def _compare(arrays, u, times, finals, arcs):
    s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
    a0, a1 = int(arrays["arc_off"][u]), int(arrays["arc_off"][u + 1])
    assert s1 - s0 == len(times) and a1 - a0 == len(arcs)
    assert arrays["state_time"][s0:s1].tolist() == times
    fg = arrays["final_graph"][s0:s1]
    assert [s for s in range(s1 - s0) if fg[s] == 0.0] == finals
    assert np.all(np.isinf(np.delete(fg, finals))) and np.array_equal(fg, arrays["final_acoustic"][s0:s1])
    got = list(zip(arrays["src"][a0:a1].tolist(), arrays["dst"][a0:a1].tolist(), arrays["ilabel"][a0:a1].tolist(),
                   arrays["graph"][a0:a1].tolist(), arrays["acoustic"][a0:a1].tolist()))
    assert got == arcs
    assert np.array_equal(arrays["ilabel"][a0:a1], arrays["olabel"][a0:a1])


def test_appendix_d_generator_matches_python_restatement():
    ns = np.array([1, 2, 3, 9, 10, 57, 200], np.int32)
    fo = np.array([3, 1, 2, 3, 6, 4, 3], np.int32)
    for seed, first in ((20260001, 0), (7, 12345)):
        arrays = klat_amd.synth(seed, ns, fo, first_utt=first, host_threads=3)
        for u in range(len(ns)):
            _compare(arrays, u, *appendix_d(seed, first + u, int(ns[u]), int(fo[u])))


@pytest.mark.parametrize("w", [1, 4, 32])
def test_wide_generator_matches_python_restatement(w):
    ns = np.array([1, 2, 3, 6, 34, 40, 131], np.int32)
    fo = np.array([2, 3, 1, 3, 3, 2, 4], np.int32)
    arrays = klat_amd.synth(20260005, ns, fo, first_utt=3, host_threads=2, wide=w)
    for u in range(len(ns)):
        _compare(arrays, u, *wide(20260005, 3 + u, int(ns[u]), int(fo[u]), w))


@pytest.mark.parametrize("w", [0, 8])
def test_generated_lattices_are_topsorted_connected_and_timed(w):
    ns = np.array([50, 120, 7], np.int32)
    fo = np.array([3, 2, 4], np.int32)
    a = klat_amd.synth(99, ns, fo, wide=w)
    for u in range(len(ns)):
        s0, S = int(a["state_off"][u]), int(ns[u])
        a0, a1 = int(a["arc_off"][u]), int(a["arc_off"][u + 1])
        src, dst = a["src"][a0:a1], a["dst"][a0:a1]
        t = a["state_time"][s0:s0 + S]
        assert np.all(dst > src) and np.all(np.diff(src) >= 0) and np.all(dst < S)
        assert np.all(t[dst] > t[src])                      # every arc has a positive duration = len(string)
        reach = np.zeros(S, bool)
        reach[0] = True
        for s, d in zip(src, dst):                          # arcs are in source order: one pass
            reach[d] |= reach[s]
        co = np.isfinite(a["final_graph"][s0:s0 + S]).copy()
        for s, d in zip(src[::-1], dst[::-1]):
            co[s] |= co[d]
        assert reach.all() and co.all()
    if w:
        depth = 1 + -(-(int(ns[1]) - 1) // w)
        assert depth < ns[1] // 4


@pytest.mark.parametrize("n,count", [(8, 19), (12, 40), (16, 60), (24, 127), (32, 191)])
def test_sorting_networks_sort(n, count):
    """The comparator networks of csrc/klat_frame.hip (fr_sort8_net ... fr_sort32_net) by the zero-one principle: every
    one of the 2^n inputs up to 24 keys; for 32 (merge exchange, which sorts by construction) a sample of them and every
    input of at most two ones or two zeros.  (tools/sort_net_check.py is the same check as a script.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import sort_net_check as chk
    ces = chk.network(n)
    assert len(ces) == count
    if n <= 24:
        assert chk.sorts_all(ces, n)
    else:
        rng = np.random.RandomState(n)
        few = np.array([[1 if i in (a, b) else 0 for i in range(n)] for a in range(n) for b in range(n)], np.int8)
        assert chk.sorts_these(ces, rng.randint(0, 2, (500000, n)).astype(np.int8))
        assert chk.sorts_these(ces, few.copy()) and chk.sorts_these(ces, 1 - few)
