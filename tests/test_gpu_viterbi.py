"""Tropical twin of the sweep: Viterbi costs, ComputeLatticeBeam and the
PruneLattice decisions from the GPU against the oracle's restatements
(latbin/lattice-prune-dyn-beam.cc:27-90; PruneLattice)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
from oracle import oracle as orc  # noqa: E402
import klat_amd  # noqa: E402
from test_gpu_accum import _split  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = klat_amd.Ctx(0)
    yield c
    c.close()


def _batch_arrays(long_spans):
    rng = np.random.RandomState(31)
    ns = np.concatenate([[1, 2], rng.randint(3, 300, size=30)]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(8080, ns, fo)
    # several final states per lattice, at different depths
    fin = rng.rand(len(arrays["final_graph"])) < 0.08
    arrays["final_graph"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 3
    arrays["final_acoustic"][fin] = rng.rand(int(fin.sum())).astype(np.float32) * 5
    if long_spans:
        lat = np.repeat(np.arange(len(ns)), np.diff(arrays["arc_off"]))
        src, dst = arrays["src"], arrays["dst"].copy()
        stretch = (np.arange(len(dst)) % 3 == 1) & (dst != src + 1)
        dst[stretch] = np.minimum(src[stretch] + rng.randint(20, 90, size=int(stretch.sum())), ns[lat][stretch] - 1)
        arrays["dst"] = dst.astype(np.int32)
    return arrays


@pytest.mark.parametrize("G,W", [(0, 0), (1, 16), (4, 8), (8, 32)])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.5, 0.1, 0.75)])
@pytest.mark.parametrize("long_spans", [False, True])
def test_viterbi_costs_beam_and_prune_mask(ctx, G, W, scales, long_spans):
    arrays = _batch_arrays(long_spans)
    b = klat_amd.Batch(ctx, arrays, lanes_per_lattice=G, ring_slots=W)
    b.viterbi(*scales)
    fw, bw, best = b.viterbi_costs()
    beam = 4.0
    keep_arc, keep_final, lattice_beam = b.prune_mask(beam)
    b.close()
    for u, lat in enumerate(_split(arrays)):
        s0, s1 = int(arrays["state_off"][u]), int(arrays["state_off"][u + 1])
        a0, a1 = int(arrays["arc_off"][u]), int(arrays["arc_off"][u + 1])
        sl = orc.scale_and_penalize(lat, *scales)
        wbeam, wfw, wbw = orc.lattice_beam(sl)
        assert np.allclose(fw[s0:s1], wfw, rtol=1e-12, atol=1e-9, equal_nan=True), u
        assert np.allclose(bw[s0:s1], wbw, rtol=1e-12, atol=1e-9, equal_nan=True), u
        assert abs(best[u] - wbw[0]) < 1e-9 or (np.isinf(best[u]) and np.isinf(wbw[0]))
        if np.isfinite(wbw[0]):
            assert abs(lattice_beam[u] - wbeam) < 1e-9, (u, lattice_beam[u], wbeam)
        # PruneLattice decisions: arcs and final weights
        pruned = sl.copy()
        ka = np.zeros(max(lat.A, 1), np.uint8)
        ks = np.zeros(max(lat.S, 1), np.uint8)
        orc.lib().kor_prune_lattice(orc.C.c_int32(lat.S), orc._p(pruned.arc_begin), orc._p(pruned.dst), orc._p(pruned.graph),
                                    orc._p(pruned.acoustic), orc._p(pruned.final_graph), orc._p(pruned.final_acoustic),
                                    orc.C.c_double(beam), orc._p(ka), orc._p(ks))
        # decisions within 1e-9 of the cutoff may legitimately differ (min-plus sums in another order)
        through = wfw[sl.src()] + (sl.graph.astype(np.float64) + sl.acoustic.astype(np.float64)) + wbw[sl.dst]
        sure = np.abs(through - (wbw[0] + beam)) > 1e-9
        assert np.array_equal(keep_arc[a0:a1][sure], ka[:lat.A][sure]), u
        want_final = ~(np.isinf(pruned.final_graph) & np.isinf(pruned.final_acoustic))
        fthrough = wfw + sl.final_graph.astype(np.float64) + sl.final_acoustic.astype(np.float64)
        fsure = ~(np.abs(fthrough - (wbw[0] + beam)) <= 1e-9)
        assert np.array_equal(keep_final[s0:s1].astype(bool)[fsure], want_final[fsure]), u


def test_prune_mask_needs_viterbi(ctx):
    arrays = klat_amd.synth(3, np.array([6], np.int32), np.array([2], np.int32))
    b = klat_amd.Batch(ctx, arrays)
    with pytest.raises(klat_amd.KlatError) as e:
        b.prune_mask(2.0)
    assert e.value.code == klat_amd.ERR_STATE
    b.close()


def test_c5_skewed_mix_viterbi_and_prune_properties(ctx):
    """BASELINE configs[4] (first half) at its shape: the skewed mix of SURVEY.md 8d C5 (sizes
    log-uniform in [200, 20 000] states, fanout in {2,3,4,6}), 400 lattices of it.  The oracle cannot
    walk 20 000-state lattices in a test's time, so the tropical twin and the pruning masks are
    checked through size-independent properties, and the lattices below 1 500 states against the
    oracle as well."""
    rng = np.random.RandomState(20260004)
    U = 400
    ns = np.exp(rng.uniform(np.log(200), np.log(20000), U)).astype(np.int32)
    fo = rng.choice([2, 3, 4, 6], U).astype(np.int32)
    arrays = klat_amd.synth(20260004, ns, fo, host_threads=8)
    scales = (1.0, 0.1, 0.0)
    b = klat_amd.Batch(ctx, arrays)
    assert b.info()["num_layouts"] >= 2  # the mix really spans several layout classes
    b.viterbi(*scales)
    fw, bw, best = b.viterbi_costs()
    b.forward_backward(*scales)
    total, _, _ = b.totals()
    so, ao = arrays["state_off"], arrays["arc_off"]
    cost = arrays["graph"].astype(np.float64) * 1.0 + (np.float32(0.1) * arrays["acoustic"]).astype(np.float64)  # ScaleLattice's float product
    lat_of_arc = np.repeat(np.arange(U), np.diff(ao))
    gsrc, gdst = arrays["src"] + so[:-1][lat_of_arc], arrays["dst"] + so[:-1][lat_of_arc]
    # 1. the best path's cost is both fwd[last] and bwd[start]; it bounds the log-semiring total
    assert np.allclose(fw[so[1:] - 1], best, rtol=1e-12) and np.allclose(bw[so[:-1]], best, rtol=1e-12)
    assert np.all(-total <= best + 1e-9) and np.all(best <= -total + np.log(6.0) * ns)  # at most fanout^S paths
    # 2. Bellman: no arc beats its destination's forward cost or its source's backward cost, and every
    #    state but the start has an arc that attains it
    through = fw[gsrc] + cost + bw[gdst]
    assert np.all(fw[gsrc] + cost >= fw[gdst] - 1e-9) and np.all(bw[gdst] + cost >= bw[gsrc] - 1e-9)
    attained = np.zeros(len(fw), bool)
    np.logical_or.at(attained, gdst, np.abs(fw[gsrc] + cost - fw[gdst]) <= 1e-9 * np.maximum(1.0, np.abs(fw[gdst])))
    attained[so[:-1]] = True
    assert attained.all()
    assert np.all(through >= best[lat_of_arc] - 1e-9)
    # 3. PruneLattice masks: exactly the arcs within the beam; monotone in the beam; a beam above the
    #    lattice's own beam keeps every arc; the best path always survives
    prev = None
    for beam in (0.5, 4.0, 50.0):
        ka, kf, lbeam = b.prune_mask(beam)
        sure = np.abs(through - (best[lat_of_arc] + beam)) > 1e-7
        assert np.array_equal(ka[sure].astype(bool), (through <= best[lat_of_arc] + beam)[sure])
        if prev is not None:
            assert np.all(ka >= prev)
        prev = ka
        assert np.all(lbeam >= -1e-9)
        per_lat_kept = np.bincount(lat_of_arc, weights=ka, minlength=U)
        assert np.all(per_lat_kept >= 1)
        roomy = lbeam < beam - 1e-6
        assert np.all(per_lat_kept[roomy] == np.diff(ao)[roomy])
        assert np.all(kf[so[1:] - 1] == 1)  # the only final state of a generated lattice ends the best path
    # 4. the lattices the oracle can walk: against its ComputeLatticeBeam restatement
    checked = 0
    for u, lat in enumerate(_split(arrays)):
        if ns[u] > 1500 or checked >= 12:
            continue
        wbeam, wfw, wbw = orc.lattice_beam(orc.scale_and_penalize(lat, *scales))
        assert np.allclose(fw[so[u]:so[u + 1]], wfw, rtol=1e-12, atol=1e-9) and np.allclose(bw[so[u]:so[u + 1]], wbw, rtol=1e-12, atol=1e-9)
        checked += 1
    assert checked >= 8
    b.close()
