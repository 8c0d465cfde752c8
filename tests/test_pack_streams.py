"""CPU-side checks of the host packer and of the scaled-linear sweep arithmetic: the packed
streams (klat_pack_host, no GPU) are interpreted in Python and compared with the oracle."""
import os

import numpy as np
import pytest

import latfmt
import stream_sim
from oracle import oracle as orc

import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

GOLDEN = ["egs_word_lattice.ark.txt", "egs_expand_lattice.char.txt", "egs_expand_lattice2.char.txt",
          "egs_char_lattice.lattice-format.ark.txt", "egs_ctc_input.lattice-format.txt"]


def _golden_lats(golden_dir):
    lats = []
    for name in GOLDEN:
        for lat in latfmt.read_text_archive(os.path.join(golden_dir, name)):
            lats.append(latfmt.prepare(lat))
    return lats


def _oracle(lats, gs=1.0, as_=1.0, pen=0.0):
    al, be, tf, tb = [], [], [], []
    for lat in lats:
        sl = orc.scale_and_penalize(lat, gs, as_, pen)
        a, b, tot, f, bw = orc.alphas_betas(sl)
        al.append(a), be.append(b), tf.append(f), tb.append(bw)
    return np.concatenate(al), np.concatenate(be), np.array(tf), np.array(tb)


def _close(got, want, tol=1e-9):
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(np.isneginf(got), np.isneginf(want))
    assert np.max(np.abs(got[fin] - want[fin]), initial=0.0) < tol


@pytest.mark.parametrize("G", [1, 2, 4, 16, 64])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_golden_lattices_through_streams(golden_dir, G, scales):
    lats = _golden_lats(golden_dir)
    arrays = klat_amd.concat_lattices(lats)
    packed = klat_amd.pack_host(arrays, lanes_per_lattice=G, ring_slots=8)
    a, b, tf, tb = stream_sim.run(packed, arrays["state_off"], *scales)
    wa, wb, wtf, wtb = _oracle(lats, *scales)
    _close(a, wa), _close(b, wb), _close(tf, wtf), _close(tb, wtb)


@pytest.mark.parametrize("G,W", [(1, 8), (4, 8), (4, 16), (8, 32)])
def test_synthetic_lattices_through_streams(G, W):
    ns = np.array([1, 2, 3, 40, 57, 64, 9, 33, 120], np.int32)
    fo = np.array([3, 3, 2, 3, 4, 6, 1, 3, 3], np.int32)
    arrays = klat_amd.synth(20260001, ns, fo)
    packed = klat_amd.pack_host(arrays, lanes_per_lattice=G, ring_slots=W)
    if W == 8:
        assert packed["info"]["far_terms"] > 0  # span 8 arcs exercise the FAR path
    a, b, tf, tb = stream_sim.run(packed, arrays["state_off"], 1.0, 0.1, 0.0)
    ab = np.concatenate([[0], np.cumsum(np.bincount(arrays["src"] + np.repeat(arrays["state_off"][:-1], np.diff(arrays["arc_off"])),
                                                 minlength=int(arrays["state_off"][-1])))]).astype(np.int64)
    wa, wb, wt, wthr = orc.batch_fb_post(arrays["state_off"], arrays["arc_off"], ab, arrays["dst"], arrays["ilabel"],
                                         arrays["graph"], arrays["acoustic"], arrays["final_graph"],
                                         arrays["final_acoustic"], 1.0, 0.1, 0.0)
    _close(a, wa, 1e-8), _close(b, wb, 1e-8)
    _close(0.5 * (tf + tb), wt, 1e-8)


def _job_layouts(packed):
    return sorted({((int(t) >> 8) & 0xFF, (int(t) >> 16) & 0xFFF) for t in packed["jobs"]["tag"]})


def test_automatic_layout_gives_every_lattice_its_own_lanes():
    """With lanes_per_lattice = 0 every lattice gets lanes ~ its fanout, lattices of equal lanes
    form a layout class (one sweep launch each, longest chain first), and in a skewed batch the
    longest lattices get more lanes and their jobs a higher wave priority."""
    ns = np.array([40, 40, 57, 64, 9, 33, 900, 12, 30, 30, 30, 30], np.int32)
    fo = np.array([2, 3, 4, 6, 1, 3, 6, 3, 2, 3, 2, 3], np.int32)
    arrays = klat_amd.synth(20260005, ns, fo)
    packed = klat_amd.pack_host(arrays)
    info = packed["info"]
    layouts = _job_layouts(packed)
    assert info["num_layouts"] == len(layouts) >= 3 and {g for g, _ in layouts} >= {1, 2, 4}
    jobs, lanes = packed["jobs"], packed["lanes"]
    lat_g = {}
    for j in jobs:
        g = (int(j["tag"]) >> 8) & 0xFF
        for li in lanes[int(j["lane_tab"]):int(j["lane_tab"]) + 64 // g]:
            if li["state_base"] >= 0:
                assert lat_g.setdefault(int(li["lattice"]), g) == g  # same lanes in both directions
    assert sorted(lat_g) == list(range(len(ns)))
    assert lat_g[4] == 1 and lat_g[0] == 2 and lat_g[1] == 4 and lat_g[6] >= 8
    # the 900-state lattice is the critical path: highest priority, and its class is launched first
    prio = jobs["tag"] & 3
    longest = int(np.argmax(jobs["nchunks"]))
    assert prio[longest] == 3 and prio.min() == 0 and longest == 0
    a, b, tf, tb = stream_sim.run(packed, arrays["state_off"], 1.0, 0.1, 0.0)
    ab = np.concatenate([[0], np.cumsum(np.bincount(arrays["src"] + np.repeat(arrays["state_off"][:-1], np.diff(arrays["arc_off"])),
                                                 minlength=int(arrays["state_off"][-1])))]).astype(np.int64)
    wa, wb, wt, _ = orc.batch_fb_post(arrays["state_off"], arrays["arc_off"], ab, arrays["dst"], arrays["ilabel"],
                                      arrays["graph"], arrays["acoustic"], arrays["final_graph"],
                                      arrays["final_acoustic"], 1.0, 0.1, 0.0)
    _close(a, wa, 1e-8), _close(b, wb, 1e-8)
    _close(0.5 * (tf + tb), wt, 1e-8)
    # a uniform batch stays one class with lanes ~ fanout and no priorities
    uni = klat_amd.pack_host(klat_amd.synth(7, np.full(300, 50, np.int32), np.full(300, 3, np.int32)))
    assert uni["info"]["num_layouts"] == 1 and uni["info"]["lanes_per_lattice"] == 4
    assert not np.any(uni["jobs"]["tag"] & 3)


def test_packer_rejects_bad_topology():
    arrays = klat_amd.synth(1, np.array([5], np.int32), np.array([2], np.int32))
    bad = dict(arrays)
    bad["dst"] = arrays["dst"].copy()
    bad["dst"][3] = 0  # arc going backwards
    with pytest.raises(klat_amd.KlatError) as e:
        klat_amd.pack_host(bad, 1, 8)
    assert e.value.code == klat_amd.ERR_BAD_TOPOLOGY
    with pytest.raises(klat_amd.KlatError) as e:
        klat_amd.pack_host(arrays, 3, 8)
    assert e.value.code == klat_amd.ERR_INVALID_ARG


def test_empty_batch_and_empty_lattice():
    z = lambda dt: np.zeros(0, dt)
    arrays = {"state_off": np.zeros(1, np.int64), "arc_off": np.zeros(1, np.int64), "src": z(np.int32),
              "dst": z(np.int32), "ilabel": z(np.int32), "olabel": z(np.int32), "graph": z(np.float32),
              "acoustic": z(np.float32), "final_graph": z(np.float32), "final_acoustic": z(np.float32)}
    p = klat_amd.pack_host(arrays, 1, 8)
    assert p["info"]["num_jobs"] == 0
    # an empty lattice between two real ones gets no stream and no posterior items
    a = klat_amd.synth(3, np.array([4, 6], np.int32), np.array([2, 2], np.int32))
    a["state_off"] = np.array([0, 4, 4, 10], np.int64)
    a["arc_off"] = np.array([0, int(a["arc_off"][1]), int(a["arc_off"][1]), int(a["arc_off"][2])], np.int64)
    p = klat_amd.pack_host(a, 2, 8)
    assert set(p["lanes"]["lattice"][p["lanes"]["state_base"] >= 0]) == {0, 2}


def test_forward_and_backward_jobs_share_their_lattices():
    """One job order for both directions (slice_jobs): every group of lattices has a forward and a backward job over
    the SAME lattices in the same slots -- what lets the device packer's fill pass write a group's two jobs side by
    side and read the group's arcs from HBM once (csrc/klat_pack_lean.hip)."""
    rng = np.random.RandomState(4)
    ns = np.concatenate([rng.randint(1, 500, size=90), [3000, 2, 1]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4, 7], size=len(ns)).astype(np.int32)
    a = klat_amd.synth(31, ns, fo)
    for G in (0, 1, 4, 16):
        h = klat_amd.pack_host(a, lanes_per_lattice=G)
        jobs, lanes = h["jobs"], h["lanes"]
        members = {0: {}, 1: {}}
        for j in jobs:
            lpw = 64 // ((int(j["tag"]) >> 8) & 0xFF)
            lat = tuple(int(x) for x in lanes["lattice"][int(j["lane_tab"]):int(j["lane_tab"]) + lpw])
            members[int(j["dir"])][lat[0]] = lat
        assert members[0] == members[1] and len(members[0]) * 2 == len(jobs), G
