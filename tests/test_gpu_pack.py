"""The device-side stream packer (csrc/klat_pack_device.hip, behind klat_batch_create) against the
host packer that defines the stream format (klat_pack_host): same layout choice, same jobs, and
the same term words and costs, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = klat_amd.Ctx(0)
    yield c
    c.close()


def _arrays(seed, long_spans, many_finals):
    rng = np.random.RandomState(seed)
    ns = np.concatenate([[1, 2, 3], rng.randint(2, 400, size=45), [1500]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4, 7], size=len(ns)).astype(np.int32)
    a = klat_amd.synth(seed, ns, fo)
    if long_spans:
        lat = np.repeat(np.arange(len(ns)), np.diff(a["arc_off"]))
        src, dst = a["src"], a["dst"].copy()
        st = (rng.rand(len(dst)) < 0.3) & (dst != src + 1)
        dst[st] = np.minimum(src[st] + rng.randint(2, 150, size=int(st.sum())), ns[lat][st] - 1)
        a["dst"] = dst.astype(np.int32)
    if many_finals:
        fin = rng.rand(len(a["final_graph"])) < 0.15
        a["final_graph"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
        a["final_acoustic"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
    # an empty lattice in the middle
    return a


@pytest.mark.parametrize("G,W", [(0, 0), (1, 16), (2, 8), (4, 32), (8, 64), (64, 16)])
@pytest.mark.parametrize("long_spans,many_finals", [(False, False), (True, True)])
def test_device_streams_equal_host_streams(ctx, G, W, long_spans, many_finals):
    a = _arrays(17 + G, long_spans, many_finals)
    host = klat_amd.pack_host(a, lanes_per_lattice=G, ring_slots=W)
    b = klat_amd.Batch(ctx, a, lanes_per_lattice=G, ring_slots=W)
    info = b.info()
    sw, sg, sa = b.streams()
    b.close()
    for k in ("lanes_per_lattice", "ring_slots", "num_jobs", "max_rounds", "fwd_slots", "bwd_slots", "far_terms", "num_layouts"):
        assert info[k] == host["info"][k], (k, info[k], host["info"][k])
    assert (info["num_layouts"] > 2) if G == 0 else (info["num_layouts"] == 1)  # automatic: lanes per lattice
    assert np.array_equal(sw, host["sw"])
    assert np.array_equal(sg.view(np.uint32), host["sg"].view(np.uint32))
    assert np.array_equal(sa.view(np.uint32), host["sa"].view(np.uint32))


def test_device_packer_reports_bad_topology(ctx):
    a = klat_amd.synth(2, np.array([6, 9], np.int32), np.array([2, 3], np.int32))
    a["dst"][int(a["arc_off"][1]) + 3] = 0  # an arc of the second lattice goes backwards
    with pytest.raises(klat_amd.KlatError) as e:
        klat_amd.Batch(ctx, a)
    assert e.value.code == klat_amd.ERR_BAD_TOPOLOGY and "lattice 1" in str(e.value)


def test_batch_from_device_resident_arrays(ctx):
    """The arc and state arrays of klat_batch_desc may be device pointers (only the two offset
    arrays are read on the host): same results as from host arrays."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")  # the runtime libklat.so itself is linked against
    a = _arrays(5, True, True)
    b = klat_amd.Batch(ctx, a)
    b.forward_backward(1.0, 0.1, 0.0)
    want = b.arc_logpost()
    b.close()
    dev, ptrs = dict(a), []
    for name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time"):
        h = np.ascontiguousarray(a[name])
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(max(h.nbytes, 4))) == 0
        assert hip.hipMemcpy(p, C.c_void_p(h.ctypes.data), C.c_size_t(h.nbytes), C.c_int(1)) == 0  # hipMemcpyHostToDevice
        ptrs.append(p)
        dev[name] = int(p.value)
    b = klat_amd.Batch(ctx, dev)
    b.forward_backward(1.0, 0.1, 0.0)
    got = b.arc_logpost()
    b.close()
    for p in ptrs:
        hip.hipFree(p)
    assert np.array_equal(got, want)


def _streams(ctx, a, lean, **kw):
    """the batch's streams through the lean passes (KLAT_PACK_LEAN=1: also where a small batch would keep the general
    ones for the segment path) or through the general passes alone (0)"""
    old = os.environ.get("KLAT_PACK_LEAN")
    os.environ["KLAT_PACK_LEAN"] = "1" if lean else "0"
    try:
        b = klat_amd.Batch(ctx, a, **kw)
        info = b.info()
        st = b.fused_streams() if kw.get("fuse_scales") else b.streams()
        b.close()
    finally:
        if old is None:
            os.environ.pop("KLAT_PACK_LEAN", None)
        else:
            os.environ["KLAT_PACK_LEAN"] = old
    return info, st


def _stretch(a, ns, rng, frac, far):
    lat = np.repeat(np.arange(len(ns)), np.diff(a["arc_off"]))
    src, dst = a["src"], a["dst"].copy()
    st = (rng.rand(len(dst)) < frac) & (dst != src + 1)
    dst[st] = np.minimum(src[st] + rng.randint(2, far, size=int(st.sum())), ns[lat][st] - 1)
    a["dst"] = np.maximum(dst, src + 1).astype(np.int32)


@pytest.mark.parametrize("fuse", [None, (1.0, 0.1, 0.5)])
@pytest.mark.parametrize("case", ["small mix", "spans to 60, many finals", "a few spans beyond the lean passes", "fixed lanes 1", "fixed lanes 8",
                                  "fixed lanes 64", "uniform", "skewed", "C2 / 10"])
def test_lean_passes_write_the_general_passes_streams(ctx, case, fuse):
    """csrc/klat_pack_lean.hip (lattices whose arcs are at most 64 states long) against the general passes of
    csrc/klat_pack_device.hip: the same streams bit for bit -- term words, costs, and with fused scales the float and
    the 16-bit residual -- on batches that mix both kinds of lattices (a job with one long-span lattice takes the
    general fill pass, its other lattices the general count pass after all), every lane count, many final states."""
    rng = np.random.RandomState(7)
    ns = np.concatenate([[1, 2, 3], rng.randint(2, 400, size=45), [1500]]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4, 7], size=len(ns)).astype(np.int32)
    kw = {}
    if case == "small mix":
        a = klat_amd.synth(11, ns, fo)
    elif case == "spans to 60, many finals":
        a = klat_amd.synth(12, ns, fo)
        _stretch(a, ns, rng, 0.3, 60)
        fin = rng.rand(len(a["final_graph"])) < 0.15
        a["final_graph"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
        a["final_acoustic"][fin] = (rng.rand(int(fin.sum())) * 3).astype(np.float32)
    elif case == "a few spans beyond the lean passes":
        a = klat_amd.synth(13, ns, fo)
        _stretch(a, ns, rng, 0.05, 150)
    elif case.startswith("fixed lanes"):
        G = int(case.split()[-1])
        a = klat_amd.synth(14 + G, ns, fo)
        _stretch(a, ns, rng, 0.2, 40)
        kw = {"lanes_per_lattice": G, "ring_slots": {1: 16, 8: 64, 64: 16}[G]}
    elif case == "uniform":
        a = klat_amd.synth(21, np.full(300, 700, np.int32), np.full(300, 3, np.int32))
    elif case == "skewed":
        n2 = np.exp(rng.uniform(np.log(50), np.log(6000), size=200)).astype(np.int32)
        a = klat_amd.synth(22, n2, rng.choice([2, 3, 4, 6], size=200).astype(np.int32))
    else:
        a = klat_amd.synth(20260001, np.full(1000, 2000, np.int32), np.full(1000, 3, np.int32))
    if fuse:
        kw["fuse_scales"] = fuse
    ig, sg = _streams(ctx, a, False, **kw)
    il, sl = _streams(ctx, a, True, **kw)
    for k in ("lanes_per_lattice", "ring_slots", "num_jobs", "max_rounds", "fwd_slots", "bwd_slots", "far_terms", "num_layouts"):
        assert ig[k] == il[k], (k, ig[k], il[k])
    for x, y in zip(sg, sl):
        assert x.shape == y.shape and x.tobytes() == y.tobytes()


def test_lean_count_pass_reports_malformed_lattices(ctx):
    os.environ["KLAT_PACK_LEAN"] = "1"
    try:
        a = klat_amd.synth(2, np.array([6, 9, 40], np.int32), np.array([2, 3, 3], np.int32))
        a["dst"][int(a["arc_off"][1]) + 3] = 0  # an arc of the second lattice goes backwards
        with pytest.raises(klat_amd.KlatError) as e:
            klat_amd.Batch(ctx, a, lanes_per_lattice=4)
        assert e.value.code == klat_amd.ERR_BAD_TOPOLOGY and "lattice 1" in str(e.value)
        a = klat_amd.synth(2, np.array([6, 9, 40], np.int32), np.array([2, 3, 3], np.int32))
        a["src"][int(a["arc_off"][2]) + 5] = -7  # a source below zero in the third
        with pytest.raises(klat_amd.KlatError) as e:
            klat_amd.Batch(ctx, a, lanes_per_lattice=4)
        assert e.value.code == klat_amd.ERR_INVALID_ARG and "lattice 2" in str(e.value)
    finally:
        os.environ.pop("KLAT_PACK_LEAN", None)


def test_borrowed_device_arrays_give_the_same_batch(ctx):
    """klat_pack_options.borrow_device_arrays: the caller's device arrays used in place."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    a = _arrays(5, False, True)
    b = klat_amd.Batch(ctx, a, fuse_scales=(1.0, 0.1, 0.0))
    b.forward_backward(1.0, 0.1, 0.0)
    want = b.arc_logpost()
    b.close()
    dev, ptrs = dict(a), []
    for name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time"):
        h = np.ascontiguousarray(a[name])
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(max(h.nbytes, 4))) == 0
        assert hip.hipMemcpy(p, C.c_void_p(h.ctypes.data), C.c_size_t(h.nbytes), C.c_int(1)) == 0
        ptrs.append(p)
        dev[name] = int(p.value)
    b = klat_amd.Batch(ctx, dev, fuse_scales=(1.0, 0.1, 0.0), borrow=True)
    b.forward_backward(1.0, 0.1, 0.0)
    got = b.arc_logpost()
    off, rec = b.segment_index()
    b.close()
    for p in ptrs:
        hip.hipFree(p)
    assert np.array_equal(got, want) and len(rec) > 0


def test_lean_passes_leave_a_crowded_position_to_the_general_ones(ctx):
    """A position of more than 127 terms does not fit the byte the lean passes keep its terms in (here: 140 parallel
    arcs between two states, so both a forward and a backward position are crowded while every arc is short): the
    lattice takes the general passes, its neighbours stay lean, the streams are the general passes' own."""
    rng = np.random.RandomState(3)
    S = 100
    src = np.concatenate([np.arange(50), np.full(140, 50), np.arange(51, S - 1)]).astype(np.int32)
    dst = (src + 1).astype(np.int32)
    A = len(src)
    tm = np.arange(S, dtype=np.int32)
    fin = np.full(S, np.inf, np.float32)
    fin0 = fin.copy()
    fin0[S - 1] = 0.0
    one = {"state_off": np.array([0, S], np.int64), "arc_off": np.array([0, A], np.int64), "src": src, "dst": dst,
           "ilabel": rng.randint(0, 20, A).astype(np.int32), "graph": (rng.rand(A) * 8).astype(np.float32),
           "acoustic": (rng.rand(A) * 40).astype(np.float32), "final_graph": fin0, "final_acoustic": fin0.copy(), "state_time": tm}
    one["olabel"] = one["ilabel"].copy()
    other = klat_amd.synth(9, np.array([80, 300, 41], np.int32), np.array([3, 2, 4], np.int32))
    a = {}
    for k in one:
        if k in ("state_off", "arc_off"):
            a[k] = np.concatenate([other[k], other[k][-1] + one[k][1:]])
        else:
            a[k] = np.concatenate([other[k], one[k]])
    for kw in ({"lanes_per_lattice": 4}, {"lanes_per_lattice": 4, "fuse_scales": (1.0, 0.1, 0.0)}, {"lanes_per_lattice": 1, "ring_slots": 16}):
        ig, sg = _streams(ctx, a, False, **kw)
        il, sl = _streams(ctx, a, True, **kw)
        assert ig["fwd_slots"] == il["fwd_slots"] and ig["bwd_slots"] == il["bwd_slots"] and ig["far_terms"] == il["far_terms"]
        for x, y in zip(sg, sl):
            assert x.tobytes() == y.tobytes()
    # and the sweep over it agrees with the oracle's totals through the usual parity path: forward == backward
    b = klat_amd.Batch(ctx, a, lanes_per_lattice=4)
    b.forward_backward(1.0, 0.1, 0.0)
    total, tf, tb = b.totals()
    b.close()
    assert np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-8


def test_batch_info_says_which_passes_packed_the_jobs(ctx):
    """klat_batch_info.lean_jobs: forced either way it is all or nothing; by default a batch with short arcs none of
    whose lattices is cut in segments is packed by the lean passes entirely, and a cut lattice's jobs are not."""
    a = klat_amd.synth(20260001, np.full(600, 900, np.int32), np.full(600, 3, np.int32))
    for env, want in (("1", "all"), ("0", "none"), (None, "by choice")):
        old = os.environ.pop("KLAT_PACK_LEAN", None)
        if env is not None:
            os.environ["KLAT_PACK_LEAN"] = env
        try:
            b = klat_amd.Batch(ctx, a)
            info = b.info()
            b.close()
        finally:
            os.environ.pop("KLAT_PACK_LEAN", None)
            if old is not None:
                os.environ["KLAT_PACK_LEAN"] = old
        assert 0 <= info["lean_jobs"] <= info["num_jobs"]
        if want == "none":
            assert info["lean_jobs"] == 0
        elif want == "all" and info["chain_lattices"] == 0:
            assert info["lean_jobs"] == info["num_jobs"]
        elif want == "by choice":
            if info["chain_lattices"] == 0:
                assert info["lean_jobs"] == info["num_jobs"], info
            else:
                assert info["lean_jobs"] < info["num_jobs"], info


def test_packers_agree_on_random_batches():
    """tools/pack_lean_fuzz.py, a short run: random sizes, fanouts, arc lengths on both sides of the lean passes' limit,
    parallel arcs, final states anywhere, dead arcs, fixed and automatic lanes, plain and fused costs."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_lean_fuzz.py"), "80", "12345"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "all 80 equal" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
