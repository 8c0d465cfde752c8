"""The multi-GPU part of the path on the one GPU a test box has: the RCCL transport with a single
rank (communicator, all-gather of sizes, device-side assembly, fetch), and the tools' --devices
mode as several real processes sharing GPU 0 with the host transport (--gather=host) -- fork before
HIP, partition by arcs, per-rank batches, gather, ordered emission on rank 0
(kwsbin2/lattice-word-index-segment.cc:96-128,260-275).  The 8-GPU RCCL run itself belongs to
the driver (bench.py --gpus 8, `index_gather`)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402
from klat_amd import dist as kdist  # noqa: E402

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "kaldi-lattice-utils_amd", "bin")


def _batch(ctx, n=40, seed=11):
    rng = np.random.RandomState(seed)
    ns = rng.randint(1, 300, size=n).astype(np.int32)
    fo = rng.choice([1, 2, 3, 4], size=n).astype(np.int32)
    arrays = klat_amd.synth(555, ns, fo)
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.1, 0.0)
    return b, arrays, ns


def test_rccl_transport_single_rank_gathers_the_batch_index():
    ctx = klat_amd.Ctx(0)
    b, arrays, ns = _batch(ctx)
    off, rec = b.segment_index()          # sizing call + host fetch: the expected answer
    comm = kdist.Comm.rccl(ctx, 0, 1, kdist.rccl_unique_id())
    assert comm.info() == {"rank": 0, "world": 1, "transport_ranks": 1}
    for _ in range(2):
        got, goff = comm.gather_index(b, klat_amd.SEGMENT_DTYPE, root=0)
        assert goff.tolist() == off.tolist() and got.tobytes() == rec.tobytes()
    assert comm.stats()["payload_bytes"] == 0
    # a later sweep invalidates the cached index: the gather must refuse stale records
    b.forward_backward(1.0, 0.2, 0.0)
    with pytest.raises(klat_amd.KlatError):
        comm.gather_index(b, klat_amd.SEGMENT_DTYPE, root=0)
    comm.close()
    b.close()
    ctx.close()


def test_device_records_match_host_fetch_for_position_index():
    import ctypes as C
    ctx = klat_amd.Ctx(0)
    b, arrays, ns = _batch(ctx, n=12, seed=3)
    state_len = np.zeros(int(arrays["state_off"][-1]), np.int32)  # any labelling will do for this plumbing check
    for u in range(len(ns)):
        s0, s1 = arrays["state_off"][u], arrays["state_off"][u + 1]
        state_len[s0:s1] = np.arange(s1 - s0) % 7
    off, rec = b.position_index(state_len)
    dev, poff, nl, nr = C.c_void_p(), C.POINTER(C.c_int64)(), C.c_int64(), C.c_int64()
    klat_amd._check(klat_amd.lib().klat_index_device_records(b.h, C.byref(dev), C.byref(poff), C.byref(nl), C.byref(nr)))
    assert nl.value == len(ns) and nr.value == len(rec) and [poff[i] for i in range(len(ns) + 1)] == off.tolist()
    assert dev.value  # a device pointer; its contents come back through the gather below
    comm = kdist.Comm.rccl(ctx, 0, 1, kdist.rccl_unique_id())
    got, goff = comm.gather_index(b, klat_amd.POSITION_DTYPE, root=0)
    assert got.tobytes() == rec.tobytes() and goff.tolist() == off.tolist()
    comm.close(), b.close(), ctx.close()


def test_device_records_merge_the_pieces_of_a_long_lattice():
    """A lattice above 16 384 arcs is indexed in pieces whose sorted runs the gather merges.  The records the
    multi-GPU gather sends from (klat_index_device_records, here BEFORE any host fetch) must be the merged ones:
    round 5 found them handed over run after run."""
    ctx = klat_amd.Ctx(0)
    ns = np.array([50, 7000, 120], np.int32)  # 3 x 6 999 = 20 997 arcs in the middle one
    arrays = klat_amd.synth(777, ns, np.full(3, 3, np.int32))
    want = klat_amd.Batch(ctx, arrays)
    want.forward_backward(1.0, 0.1, 0.0)
    off, rec = want.segment_index()  # sizing call + host fetch
    lo, hi = int(off[1]), int(off[2])
    assert np.all(np.diff(rec["logp"][lo:hi]) <= 0), "the long lattice's records are sorted by descending posterior"
    b = klat_amd.Batch(ctx, arrays)
    b.forward_backward(1.0, 0.1, 0.0)
    import ctypes as C
    f, keep = klat_amd.Batch._filter(None, None)
    o2, n = np.zeros(len(ns) + 1, np.int64), C.c_int64(0)
    klat_amd._check(klat_amd.lib().klat_segment_index(b.h, C.byref(f), klat_amd._p(o2), None, C.c_int64(0), C.byref(n)))  # sizing only
    comm = kdist.Comm.rccl(ctx, 0, 1, kdist.rccl_unique_id())
    got, goff = comm.gather_index(b, klat_amd.SEGMENT_DTYPE, root=0)
    assert goff.tolist() == off.tolist() and got.tobytes() == rec.tobytes()
    comm.close(), b.close(), want.close(), ctx.close()


@pytest.mark.parametrize("tool,extra", [("lattice-word-index-segment", []), ("lattice-word-index-position", []),
                                        ("lattice-word-index-segment", ["--beam=4", "--acoustic-scale=0.1"])])
def test_tool_devices_mode_three_processes_equals_one(tmp_path, tool, extra):
    rng = np.random.RandomState(8)
    ns = rng.randint(1, 150, size=37).astype(np.int32)
    ns[5] = 1  # a lattice without arcs: still one (empty) record line, kwsbin2/lattice-word-index-segment.cc:64-69
    fo = rng.choice([1, 2, 3], size=37).astype(np.int32)
    arrays = klat_amd.synth(2024, ns, fo)
    ark = tmp_path / "lats.ark.txt"
    from test_host_tools import _write_text_archive
    _write_text_archive(str(ark), arrays, ns)
    outs = []
    for dev in (["--device=0"], ["--devices=0,0,0", "--gather=host", "--batch-size=5"], ["--devices=0,0", "--gather=host", "--batch-size=64"]):
        p = subprocess.run([os.path.join(BIN, tool)] + extra + dev + ["ark:%s" % ark, "ark,t:-"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        outs.append(p.stdout)
    assert len(outs[0].splitlines()) == len(ns)
    # A lattice's lanes-per-lattice layout depends on the batch it shares (plan_layout), so sums may
    # differ in the last bits between the runs: compare as SURVEY.md C.3 prescribes -- keys in input
    # order, integer fields bit-exact as sets, values within tolerance, order valid up to ties.
    import latfmt
    nint = 4 if tool.endswith("position") else 3
    want = [latfmt.parse_tuple_line(l, nint) for l in outs[0].splitlines()]
    for other in outs[1:]:
        got = [latfmt.parse_tuple_line(l, nint) for l in other.splitlines()]
        assert [k for k, _ in got] == [k for k, _ in want] == ["utt%07d" % u for u in range(len(ns))]
        for (_, g), (_, w) in zip(got, want):
            wk = {r[:nint]: r[nint] for r in w}
            assert sorted(r[:nint] for r in g) == sorted(wk)
            assert all(abs(r[nint] - wk[r[:nint]]) <= 1e-6 * max(1.0, abs(r[nint])) for r in g)
            lp = [r[nint] for r in g]
            assert all(lp[i] >= lp[i + 1] - 1e-6 * max(1.0, abs(lp[i])) for i in range(len(lp) - 1))


def test_tool_devices_mode_reports_a_failing_rank(tmp_path):
    """A rank that cannot do its share (device index out of range) must end the whole run with a
    non-zero exit code instead of leaving the writer waiting."""
    ns = np.array([30, 20, 10, 25], np.int32)
    arrays = klat_amd.synth(1, ns, np.full(4, 2, np.int32))
    ark = tmp_path / "lats.ark.txt"
    from test_host_tools import _write_text_archive
    _write_text_archive(str(ark), arrays, ns)
    p = subprocess.run([os.path.join(BIN, "lattice-word-index-segment"), "--devices=0,99", "--gather=host", "ark:%s" % ark, "ark,t:-"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "device index out of range" in p.stderr or "worker rank failed" in p.stderr


def _archives(tmp_path, n=60, seed=12):
    rng = np.random.RandomState(seed)
    ns = rng.randint(2, 200, size=n).astype(np.int32)
    fo = rng.choice([1, 2, 3], size=n).astype(np.int32)
    arrays = klat_amd.synth(77, ns, fo)
    txt = tmp_path / "lats.ark.txt"
    from test_host_tools import _write_text_archive
    _write_text_archive(str(txt), arrays, ns)
    binary = tmp_path / "lats.ark"
    p = subprocess.run([os.path.join(BIN, "klat-lattice-copy"), "ark:%s" % txt, "ark:%s" % binary], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return arrays, ns, str(txt), str(binary)


@pytest.mark.parametrize("binary", [False, True])
def test_devices_mode_each_rank_parses_only_its_share(tmp_path, binary):
    """The reference reads every lattice once (kwsbin2/lattice-word-index-segment.cc:261-274).  With three
    ranks on a plain file each rank scans the entry boundaries and PARSES a third of the arcs -- not the
    whole archive, seven eighths of which it would throw away on an 8-GPU node -- and the records are what
    the single-GPU run writes."""
    import re
    arrays, ns, txt, bin_ = _archives(tmp_path)
    ark = bin_ if binary else txt
    one = subprocess.run([os.path.join(BIN, "lattice-word-index-segment"), "--acoustic-scale=0.1", "ark:%s" % ark, "ark,t:-"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    p = subprocess.run([os.path.join(BIN, "lattice-word-index-segment"), "--acoustic-scale=0.1", "--verbose=1", "--devices=0,0,0", "--gather=host",
                        "--batch-size=7", "ark:%s" % ark, "ark,t:-"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert [l.split()[0] for l in p.stdout.splitlines()] == [l.split()[0] for l in one.stdout.splitlines()]
    assert len(p.stdout) == len(one.stdout)
    total_arcs = int(arrays["arc_off"][-1])
    got = {}
    for m in re.finditer(r"Rank (\d)/3: parsed (\d+) of (\d+) lattices \((\d+) arcs\), archive scanned in place", p.stderr):
        got[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    assert sorted(got) == [0, 1, 2], p.stderr[-3000:]
    assert all(v[1] == len(ns) for v in got.values())
    assert sum(v[0] for v in got.values()) == len(ns) and sum(v[2] for v in got.values()) == total_arcs
    for r, (_, _, arcs) in got.items():
        assert 0.2 * total_arcs <= arcs <= 0.47 * total_arcs, (r, arcs, total_arcs)
    # a command instead of a file cannot be scanned ahead: every rank parses everything, same records
    q = subprocess.run([os.path.join(BIN, "lattice-word-index-segment"), "--acoustic-scale=0.1", "--verbose=1", "--devices=0,0", "--gather=host",
                        "--batch-size=16", "ark:cat %s|" % ark, "ark,t:-"], capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-3000:]
    assert [l.split()[0] for l in q.stdout.splitlines()] == [l.split()[0] for l in one.stdout.splitlines()]
    assert "archive read as a stream" in q.stderr


@pytest.mark.parametrize("tool,args", [("lattice-to-word-frame-post", ["--acoustic-scale=0.1"]),
                                       ("lattice-best-path2", ["--acoustic-scale=0.1", "--nbest=3"])])
def test_job_tools_devices_mode_pieces_concatenate_to_the_single_gpu_output(tmp_path, tool, args):
    """lattice-to-word-frame-post (entry point #1) and lattice-best-path2 (BASELINE configs[2]) with
    --devices: every rank takes a contiguous share of the archive and writes table JOB; the pieces in
    job order are the single-GPU output, key for key (per-utterance independent loops,
    latbin/lattice-to-word-frame-post.cc:68, latbin/lattice-best-path2.cc:78)."""
    arrays, ns, txt, bin_ = _archives(tmp_path, n=23, seed=4)
    one = subprocess.run([os.path.join(BIN, tool)] + args + ["ark:%s" % bin_, "ark,t:%s" % (tmp_path / "one.txt")],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    p = subprocess.run([os.path.join(BIN, tool)] + args + ["--devices=0,0,0", "--batch-size=4", "ark:%s" % bin_, "ark,t:%s" % (tmp_path / "piece.JOB.txt")],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    pieces = [open(tmp_path / ("piece.%d.txt" % j)).read() for j in (1, 2, 3)]
    assert all(pieces), "every rank writes a piece"
    want = open(tmp_path / "one.txt").read()
    got = "".join(pieces)
    wl, gl = want.splitlines(), got.splitlines()
    assert [l.split()[0] for l in gl] == [l.split()[0] for l in wl]
    if tool == "lattice-best-path2":
        assert got == want  # integer transcripts
    else:  # posteriors: the layout of a lattice's sweep depends on the batch it shares -- last bits may differ
        for a, b in zip(gl, wl):
            ta, tb = a.split(), b.split()
            assert len(ta) == len(tb)
            for x, y in zip(ta[1:], tb[1:]):
                assert x == y or abs(float(x) - float(y)) <= 1e-4 * max(1.0, abs(float(y)))
    # no JOB in the wspecifier: refused before any rank starts work
    r = subprocess.run([os.path.join(BIN, tool)] + args + ["--devices=0,0", "ark:%s" % bin_, "ark,t:%s" % (tmp_path / "x.txt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "JOB" in r.stderr
