"""The C++ host side: command-line surface of the three tools (SURVEY.md 2.3 / interface B1),
archive formats, and -- on the GPU box -- their output against the reference's README vectors."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import latfmt
from oracle import oracle as orc

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BIN = os.path.join(ROOT, "kaldi-lattice-utils_amd", "bin")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402

TOOLS = ["lattice-to-word-frame-post", "lattice-word-index-segment", "lattice-word-index-position"]


def run(tool, *args, stdin=None, timeout=300):
    p = subprocess.run([os.path.join(BIN, tool)] + list(args), input=stdin, capture_output=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


# ------------------------------------------------------------------ CPU-side (no GPU needed)
def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "klat.h")).read()
    names = sorted(set(re.findall(r"\b(klat_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = klat_amd.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/klat.h but not exported: %s" % missing
    assert b"gfx950" in lib.klat_version()
    # the multi-GPU library (include/klat_dist.h)
    from klat_amd import dist as kdist
    hdr = open(os.path.join(ROOT, "include", "klat_dist.h")).read()
    names = sorted(set(re.findall(r"^(?:int|void)\s+(klat_[a-z_0-9]+)\s*\(", hdr, re.M)))
    assert len(names) >= 10, names
    dlib = kdist.dlib()
    missing = [n for n in names if not hasattr(dlib, n)]
    assert not missing, "declared in include/klat_dist.h but not exported: %s" % missing


@pytest.mark.parametrize("tool", TOOLS)
def test_usage_and_exit_codes(tool):
    rc, out, err = run(tool)  # wrong number of arguments -> usage, exit(1)
    assert rc == 1 and "Usage: " + tool in err
    rc, out, err = run(tool, "--help")
    assert rc == 0 and "--acoustic-scale" in err
    rc, out, err = run(tool, "--no-such-option=1", "ark:a", "ark:b")
    assert rc == 255 and "Invalid option" in err  # exception path: return -1


def test_bad_beam_is_refused_loudly(tmp_path):
    rc, out, err = run("lattice-word-index-segment", "--beam=-5", "ark:/dev/null", "ark,t:-")
    assert rc == 255 and "--beam must be positive" in err


LATTICE_FORMAT_WITH_CHAINS = """chains
0 1 11 5 1.5,2.25
1 2 12 0 0.25,0.5
2 3 13 0 0.125,1
3 4 0 0
0 5 21 6 3,1
5 4 22 7 0.5,0.5
4 6 31 8
6 7 32 0 1,1
4 7 33 9 2,2
7 0.5,0.25

allwords
0 1 1 1 0.5,0.5
1 2 2 2 0.5,0.5
2

"""


def test_lattice_format_entries_merge_linear_chains(tmp_path):
    """A Lattice-format entry read as CompactLattice (kwsbin2/README.md:176) goes through
    ConvertLattice: states 1, 2, 3 (one arc in, one arc out without word, not final) fold into
    the arc 0 -> 4 carrying word 5, weight sums and three transition ids; state 5 stays (its arc
    out carries word 7), state 6 goes.  Checked against the test-side restatement in latfmt.py
    through klat-lattice-copy, text and binary."""
    src = tmp_path / "in.txt"
    src.write_text(LATTICE_FORMAT_WITH_CHAINS)
    want = latfmt.read_text_archive(str(src))
    assert [l.S for l in want] == [4, 3]  # kept: 0, 4, 5, 7
    w0 = want[0]
    assert (int(w0.dst[0]), int(w0.olabel[0]), int(w0.dur[0])) == (1, 5, 3)  # 0 -> (old 4 = new 1), word 5, ids 11_12_13
    assert abs(float(w0.graph[0]) - 1.875) < 1e-7 and abs(float(w0.acoustic[0]) - 3.75) < 1e-7
    for mode in ("ark,t:", "ark:"):
        mid = str(tmp_path / ("copy" + ("t" if "t" in mode else "b")))
        rc, out, err = run("klat-lattice-copy", "ark:" + str(src), mode + mid)
        assert rc == 0, err
        rc, out, err = run("klat-lattice-copy", "ark:" + mid, "ark,t:-")
        assert rc == 0, err
        got_path = tmp_path / "got.txt"
        got_path.write_bytes(out)
        got = latfmt.read_text_archive(str(got_path))
        assert [g.key for g in got] == ["chains", "allwords"]
        for g, w in zip(got, want):
            assert g.S == w.S and g.A == w.A
            for f in ("arc_begin", "dst", "ilabel", "olabel", "graph", "acoustic", "dur", "final_graph", "final_acoustic"):
                assert np.array_equal(getattr(g, f), getattr(w, f)), (g.key, f)


def _write_text_archive(path, arrays, ns):
    with open(path, "w") as f:
        for u in range(len(ns)):
            s0, a0, a1 = arrays["state_off"][u], arrays["arc_off"][u], arrays["arc_off"][u + 1]
            f.write("utt%07d\n" % u)
            for i in range(a0, a1):
                s, d = arrays["src"][i], arrays["dst"][i]
                dur = arrays["state_time"][s0 + d] - arrays["state_time"][s0 + s]
                f.write("%d\t%d\t%d\t%.9g,%.9g,%s\n" % (s, d, arrays["ilabel"][i], arrays["graph"][i], arrays["acoustic"][i],
                                                        "_".join(["7"] * dur)))
            f.write("%d\n\n" % (ns[u] - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("beam", [1.5, 6.0])
@pytest.mark.parametrize("where", ["host", "device"])
def test_segment_tool_with_beam_matches_oracle_pipeline(tmp_path, beam, where):
    """--beam: ScaleLattice -> AddWordInsPen -> PruneLattice -> TopSort -> times -> sweep -> index
    (kwsbin2/lattice-word-index-segment.cc:41-63).  The decisions are taken on the host lattice by
    lattice, or with --device-prune by the Viterbi sweep on the GPU; Connect() is host-side."""
    ns = np.array([40, 25, 60], np.int32)
    fo = np.array([3, 2, 4], np.int32)
    arrays = klat_amd.synth(99, ns, fo)
    txt = str(tmp_path / "in.txt")
    _write_text_archive(txt, arrays, ns)
    args = ["--acoustic-scale=0.1", "--insertion-penalty=0.5", "--beam=%g" % beam] + (["--device-prune=true"] if where == "device" else [])
    rc, out, err = run("lattice-word-index-segment", *args, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = out.decode().strip("\n").split("\n")
    assert len(lines) == 3
    npruned = 0
    for lat, line in zip(latfmt.read_text_archive(txt), lines):
        sl = orc.scale_and_penalize(lat, 1.0, 0.1, 0.5)
        pl = orc.prune_lattice(sl, beam)
        assert pl is not None
        npruned += pl.A < lat.A
        times, _, _ = orc.state_times(pl)
        alpha, beta, total, _, _ = orc.alphas_betas(pl)
        want = orc.segment_index(pl, times, alpha, beta, total)
        key, got = latfmt.parse_tuple_line(line, 3)
        assert key == lat.key
        assert sorted(g[:3] for g in got) == sorted(w[:3] for w in want)
        wk = {w[:3]: w[3] for w in want}
        assert all(abs(g[3] - wk[g[:3]]) < 1e-5 for g in got)
    assert npruned >= 2 or beam > 5  # the small beam really removes arcs


def test_binary_archive_round_trip(tmp_path, golden_dir):
    """The binary CompactLattice reader (Appendix B.4) against this build's own writer, and the
    exact byte layout of the stream header."""
    src = os.path.join(golden_dir, "egs_expand_lattice2.char.txt")
    b = str(tmp_path / "x.ark")
    rc, _, err = run("klat-lattice-copy", "ark:" + src, "ark:" + b)
    assert rc == 0, err
    raw = open(b, "rb").read()
    assert raw.startswith(b"lat2 ")
    magic, n1 = struct.unpack_from("<ii", raw, 5)
    assert magic == 2125659606 and raw[13:13 + n1] == b"vector"
    t = str(tmp_path / "x.txt")
    rc, _, err = run("klat-lattice-copy", "ark:" + b, "ark,t:" + t)
    assert rc == 0, err
    (a,) = latfmt.read_text_archive(src)
    (c,) = latfmt.read_text_archive(t)
    assert a.key == c.key == "lat2"
    for f in ("arc_begin", "dst", "ilabel", "graph", "acoustic", "dur", "final_graph", "final_acoustic"):
        assert np.array_equal(getattr(a, f), getattr(c, f)), f
    # piped specifiers
    rc, out, err = run("klat-lattice-copy", "ark:cat %s |" % b, "ark,t:| cat")
    assert rc == 0 and b"lat2" in out


# ------------------------------------------------------------------ GPU box
def _readme(golden_dir, kind):
    with open(os.path.join(golden_dir, "readme_expected.txt")) as f:
        for line in f:
            if line.startswith(kind + " "):
                return line.split(None, 1)[1].strip()


@pytest.mark.gpu
def test_segment_tool_matches_readme(golden_dir):
    rc, out, err = run("lattice-word-index-segment", "ark:" + os.path.join(golden_dir, "egs_word_lattice.ark.txt"), "ark,t:-")
    assert rc == 0, err
    line = out.decode().strip("\n")
    assert line.endswith(" ") and "; " in line  # holder text format: fields end with a space
    key, got = latfmt.parse_tuple_line(line, 3)
    _, want = latfmt.parse_tuple_line(_readme(golden_dir, "segment"), 3)
    assert key == "lat1" and sorted(g[:3] for g in got) == sorted(w[:3] for w in want)
    # text identity with the README line up to the order of records whose logp tie
    canon = lambda recs: sorted("%d %d %d %s" % (r[0], r[1], r[2], ("%.7g" % r[3]) if abs(r[3]) > 5e-8 else "0") for r in recs)
    assert canon(got) == canon(want)
    lps = [g[3] for g in got]
    assert all(x >= y - 1e-7 for x, y in zip(lps[:-1], lps[1:]))


@pytest.mark.gpu
def test_position_tool_matches_readme(golden_dir):
    rc, out, err = run("lattice-word-index-position", "ark:" + os.path.join(golden_dir, "egs_word_lattice.ark.txt"), "ark,t:-")
    assert rc == 0, err
    key, got = latfmt.parse_tuple_line(out.decode(), 4)
    _, want = latfmt.parse_tuple_line(_readme(golden_dir, "position"), 4)
    canon = lambda recs: sorted("%d %d %d %d %s" % (r[0], r[1], r[2], r[3], ("%.7g" % r[4]) if abs(r[4]) > 5e-8 else "0") for r in recs)
    assert key == "lat1" and canon(got) == canon(want)


@pytest.mark.gpu
def test_frame_post_tool_text_and_binary(tmp_path, golden_dir):
    src = "ark:" + os.path.join(golden_dir, "egs_word_lattice.ark.txt")
    rc, out, err = run("lattice-to-word-frame-post", src, "ark,t:-")
    assert rc == 0, err
    key, frames = latfmt.parse_posterior_line(out.decode())
    assert key == "lat1" and len(frames) == 33
    assert [l for l, _ in frames[0]] == [2, 1] and [l for l, _ in frames[8]] == [5, 4] and [l for l, _ in frames[20]] == [6]
    assert abs(frames[0][0][1] - np.log(0.8)) < 1e-6 and abs(frames[0][1][1] - np.log(0.2)) < 1e-6
    b = str(tmp_path / "post.ark")
    rc, _, err = run("lattice-to-word-frame-post", src, "ark:" + b)
    assert rc == 0, err
    raw = open(b, "rb").read()
    assert raw.startswith(b"lat1 \x00B\x04" + struct.pack("<i", 33))  # key, binary marker, size-tagged int32 T


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_position_post_tool_matches_oracle(tmp_path, golden_dir, scales):
    """lattice-to-word-position-post (latbin/lattice-to-word-position-post.cc): the golden word
    lattice plus a synthetic archive, against the oracle's direct restatement."""
    gs, ac, pen = scales
    ns = np.array([25, 1, 40, 9], np.int32)
    fo = np.array([3, 1, 2, 4], np.int32)
    arrays = klat_amd.synth(777, ns, fo)
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        f.write(open(os.path.join(golden_dir, "egs_word_lattice.ark.txt")).read().rstrip("\n") + "\n\n")
        latfmt.write_synth_text(arrays, f, ["u%d" % i for i in range(len(ns))])
    rc, out, err = run("lattice-to-word-position-post", "--graph-scale=%g" % gs, "--acoustic-scale=%g" % ac,
                       "--insertion-penalty=%g" % pen, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = [l for l in out.decode().splitlines() if l.strip()]
    lats = latfmt.read_text_archive(txt)
    assert len(lines) == len(lats) == 1 + len(ns)
    for line, lat in zip(lines, lats):
        key, got = latfmt.parse_posterior_line(line)
        assert key == lat.key
        slat = orc.scale_and_penalize(latfmt.prepare(lat), gs, ac, pen)
        dlat, state_len, max_len = orc.disambiguate_length(slat)
        if dlat.S == 0:
            assert got == []
            continue
        alpha, beta, total, _, _ = orc.alphas_betas(dlat)
        want = orc.position_post(dlat, state_len, max_len, alpha, beta, total)
        assert len(got) == len(want) == max_len
        for g, w in zip(got, want):
            assert sorted(l for l, _ in g) == sorted(l for l, _ in w)
            wd = dict(w)
            for l, p in g:
                assert abs(p - float(wd[l])) < 1e-4 * max(1.0, abs(p))
            assert all(g[i][1] >= g[i + 1][1] - 1e-6 for i in range(len(g) - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_length_dist_tool_matches_oracle(tmp_path, golden_dir, scales):
    """lattice-to-transcript-length-dist (latbin/lattice-to-transcript-length-dist.cc) against the
    oracle: a lattice with several final states and epsilon arcs, the golden word lattice and a
    synthetic one in the same archive."""
    gs, ac, pen = scales
    arrays = klat_amd.synth(99, np.array([30, 1], np.int32), np.array([3, 1], np.int32))
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        for name in ("multi_final_lattice.ark.txt", "egs_word_lattice.ark.txt"):
            f.write(open(os.path.join(golden_dir, name)).read().rstrip("\n") + "\n\n")
        latfmt.write_synth_text(arrays, f, ["u0", "u1"])
    rc, out, err = run("lattice-to-transcript-length-dist", "--graph-scale=%g" % gs, "--acoustic-scale=%g" % ac,
                       "--insertion-penalty=%g" % pen, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = [l for l in out.decode().splitlines() if l.strip()]
    lats = latfmt.read_text_archive(txt)
    assert len(lines) == len(lats) == 4
    for line, lat in zip(lines, lats):
        key, got = latfmt.parse_posterior_line(line)
        assert key == lat.key and len(got) == 1
        slat = orc.scale_and_penalize(latfmt.prepare(lat), gs, ac, pen)
        dlat, state_len, _ = orc.disambiguate_length(slat)
        alpha, beta, total, _, _ = orc.alphas_betas(dlat)
        want = orc.length_dist(dlat, state_len, alpha, total)
        assert [g[0] for g in got[0]] == [w[0] for w in want], (key, got, want)
        for g, w in zip(got[0], want):
            assert abs(g[1] - float(w[1])) < 1e-4 * max(1.0, abs(g[1]))
    assert len(latfmt.parse_posterior_line(lines[0])[1][0]) >= 3  # several lengths in the first lattice


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_best_path2_tool_matches_oracle(tmp_path, golden_dir, scales):
    """lattice-best-path2 (latbin/lattice-best-path2.cc) against the oracle: transcripts (integer
    labels, exact) wherever the minimum-cost path is unique, and the logged costs."""
    gs, ac, pen = scales
    ns = np.array([30, 12, 1, 45], np.int32)
    fo = np.array([3, 2, 1, 4], np.int32)
    arrays = klat_amd.synth(2024, ns, fo)
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        for name in ("egs_word_lattice.ark.txt", "multi_final_lattice.ark.txt"):
            f.write(open(os.path.join(golden_dir, name)).read().rstrip("\n") + "\n\n")
        latfmt.write_synth_text(arrays, f, ["u%d" % i for i in range(len(ns))])
    rc, out, err = run("lattice-best-path2", "--graph-scale=%g" % gs, "--acoustic-scale=%g" % ac,
                       "--insertion-penalty=%g" % pen, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = out.decode().split("\n")[:-1]
    lats = latfmt.read_text_archive(txt)
    assert len(lines) == len(lats) == 2 + len(ns)
    costs = [float(l.split("best cost is ")[1].split(" over ")[0]) for l in err.splitlines() if "best cost is" in l]
    assert len(costs) == len(lats)
    unique = 0
    for line, lat, cost in zip(lines, lats, costs):
        toks = line.split()
        assert toks[0] == lat.key
        got = [int(t) for t in toks[1:]]
        want, wcost, ways = orc.best_path2(orc.scale_and_penalize(latfmt.prepare(lat), gs, ac, pen))
        assert abs(cost - float(wcost)) < 1e-4 * max(1.0, abs(cost)), (lat.key, cost, wcost)
        if ways == 1:
            assert got == want, (lat.key, got, want)
            unique += 1
    assert unique >= 4
    # binary output: BasicVectorHolder<int32>
    b = str(tmp_path / "tra.ark")
    rc, _, err = run("lattice-best-path2", "ark:" + txt, "ark:" + b)
    assert rc == 0, err
    raw = open(b, "rb").read()
    n = len([int(t) for t in lines[0].split()[1:]]) if scales == (1.0, 1.0, 0.0) else None
    assert raw.startswith(lats[0].key.encode() + b" \x00B\x04")
    if n is not None:
        assert struct.unpack("<i", raw[len(lats[0].key) + 4:len(lats[0].key) + 8])[0] == n


@pytest.mark.gpu
def test_utterance_tool_matches_readme(golden_dir):
    """kwsbin2/README.md:25, the known answer of lattice-word-index-utterance."""
    rc, out, err = run("lattice-word-index-utterance", "ark:" + os.path.join(golden_dir, "egs_word_lattice.ark.txt"), "ark,t:-")
    assert rc == 0, err
    key, got = latfmt.parse_tuple_line(out.decode(), 1)
    _, want = latfmt.parse_tuple_line(_readme(golden_dir, "utterance"), 1)
    assert key == "lat1" and len(got) == len(want) == 8
    assert sorted(g[0] for g in got) == sorted(w[0] for w in want)
    wd = {w[0]: w[1] for w in want}
    for g in got:
        assert abs(g[1] - wd[g[0]]) < 5e-7, (g, wd[g[0]])
    assert all(got[i][1] >= got[i + 1][1] - 1e-9 for i in range(len(got) - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("args,flt", [([], {}), (["--acoustic-scale=0.1", "--insertion-penalty=0.5", "--exclude-words=3 7 11"], {"exclude": [3, 7, 11]}),
                                      (["--graph-scale=0.7", "--include-words=" + " ".join(str(i) for i in range(1, 40))], {"include": list(range(1, 40))})])
def test_utterance_tool_matches_oracle(tmp_path, golden_dir, args, flt):
    """lattice-word-index-utterance on an archive of synthetic lattices (more than 64 distinct
    words in one of them: several waves per lattice), the multi-final lattice and an empty one."""
    ns = np.array([60, 1, 25, 9], np.int32)
    fo = np.array([3, 1, 2, 4], np.int32)
    arrays = klat_amd.synth(515, ns, fo)
    arrays["olabel"][arrays["arc_off"][2]:arrays["arc_off"][3]] %= 5  # few words, many arcs each
    arrays["ilabel"] = arrays["olabel"].copy()
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        f.write(open(os.path.join(golden_dir, "multi_final_lattice.ark.txt")).read().rstrip("\n") + "\n\n")
        f.write("empty_utt\n\n")
        latfmt.write_synth_text(arrays, f, ["u%d" % i for i in range(len(ns))])
    rc, out, err = run("lattice-word-index-utterance", *args, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = out.decode().split("\n")[:-1]
    lats = latfmt.read_text_archive(txt)
    assert len(lines) == len(lats) == 2 + len(ns)
    gs = float([a.split("=")[1] for a in args if a.startswith("--graph-scale")][0]) if any(a.startswith("--graph-scale") for a in args) else 1.0
    ac = float([a.split("=")[1] for a in args if a.startswith("--acoustic-scale")][0]) if any(a.startswith("--acoustic-scale") for a in args) else 1.0
    pen = float([a.split("=")[1] for a in args if a.startswith("--insertion-penalty")][0]) if any(a.startswith("--insertion-penalty") for a in args) else 0.0
    many = 0
    for line, lat in zip(lines, lats):
        key, got = latfmt.parse_tuple_line(line, 1) if len(line.split()) > 1 else (line.split()[0], [])
        assert key == lat.key
        want = orc.utterance_index(orc.scale_and_penalize(latfmt.prepare(lat), gs, ac, pen), **flt)
        assert sorted(g[0] for g in got) == sorted(w[0] for w in want), key
        wd = dict(want)
        for g in got:
            assert abs(g[1] - wd[g[0]]) < 1e-6 * max(1.0, abs(g[1])), (key, g, wd[g[0]])
        assert all(got[i][1] >= got[i + 1][1] - 1e-9 for i in range(len(got) - 1))
        many = max(many, len(got))
    if not flt:
        assert many > 64


@pytest.mark.gpu
@pytest.mark.parametrize("limits", [("--max-arcs=60",), ("--max-states=12",), ("--max-arcs=200", "--max-states=40"), ()])
@pytest.mark.parametrize("scales", [(1.0, 1.0, 0.0), (0.7, 0.1, 0.5)])
def test_prune_dyn_beam_tool_matches_oracle_loop(tmp_path, golden_dir, limits, scales):
    """lattice-prune-dyn-beam (latbin/lattice-prune-dyn-beam.cc): the surviving arcs and states are
    those of the reference's iterated PruneLattice loop (oracle: literally iterated), the lattice
    comes back with its stored weights and transition ids, in text and in binary."""
    gs, ac, pen = scales
    ns = np.array([50, 30, 1, 80], np.int32)
    fo = np.array([3, 2, 1, 4], np.int32)
    arrays = klat_amd.synth(31337, ns, fo)
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        f.write(open(os.path.join(golden_dir, "multi_final_lattice.ark.txt")).read().rstrip("\n") + "\n\n")
        latfmt.write_synth_text(arrays, f, ["u%d" % i for i in range(len(ns))])
    args = ["--graph-scale=%g" % gs, "--acoustic-scale=%g" % ac, "--insertion-penalty=%g" % pen] + list(limits)
    out_txt, out_bin, back = str(tmp_path / "out.txt"), str(tmp_path / "out.ark"), str(tmp_path / "back.txt")
    rc, _, err = run("lattice-prune-dyn-beam", *args, "ark:" + txt, "ark,t:" + out_txt)
    assert rc == 0, err
    rc, _, err2 = run("lattice-prune-dyn-beam", *args, "ark:" + txt, "ark:" + out_bin)
    assert rc == 0, err2
    rc, _, err3 = run("klat-lattice-copy", "ark:" + out_bin, "ark,t:" + back)
    assert rc == 0, err3
    max_arcs = max([int(l.split("=")[1]) for l in limits if l.startswith("--max-arcs")] + [0]) or 2 ** 31 - 1
    max_states = max([int(l.split("=")[1]) for l in limits if l.startswith("--max-states")] + [0]) or 2 ** 31 - 1
    ins = latfmt.read_text_archive(txt)
    outs = latfmt.read_text_archive(out_txt)
    assert [o.key for o in outs] == [i.key for i in ins]
    pruned = 0
    for lat, got in zip(ins, outs):
        lat = latfmt.prepare(lat)
        arcs, states, beam = orc.prune_dyn_beam(orc.scale_and_penalize(lat, gs, ac, pen), max_arcs, max_states)
        assert arcs is not None
        pruned += len(arcs) < lat.A
        assert got.S == len(states) and got.A == len(arcs), (lat.key, got.S, got.A, len(states), len(arcs))
        newid = -np.ones(lat.S, np.int64)
        newid[states] = np.arange(len(states))
        assert np.array_equal(got.dst, newid[lat.dst[arcs]]) and np.array_equal(got.ilabel, lat.ilabel[arcs])
        assert np.allclose(got.graph, lat.graph[arcs], rtol=1e-6) and np.allclose(got.acoustic, lat.acoustic[arcs], rtol=1e-6)
        assert np.array_equal(got.dur, lat.dur[arcs])
        assert got.S <= max_states and got.A <= max_arcs or beam <= 1e-3
    if limits:
        assert pruned >= 2
    # the binary output reads back as the same lattices; real transition ids survive (all 7 here)
    assert "7_7" in open(out_txt).read()
    for a, b in zip(outs, latfmt.read_text_archive(back)):
        assert a.key == b.key and np.array_equal(a.dst, b.dst) and np.array_equal(a.dur, b.dur) and np.allclose(a.graph, b.graph, rtol=1e-6)


@pytest.mark.gpu
def test_prune_dyn_beam_with_an_arc_on_no_complete_path(tmp_path):
    """An arc into a dead end makes ComputeLatticeBeam +inf (latbin/lattice-prune-dyn-beam.cc:83-85);
    the first PruneLattice pass then only trims the lattice, which is enough here.  Where it is not,
    the reference never leaves its loop; this tool says so and writes the trimmed lattice."""
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        f.write("dead\n0\t1\t5\t1,2,7\n0\t2\t6\t0.5,1,7\n1\t3\t8\t1,1,7_7\n3\n\n")
    ins = latfmt.read_text_archive(txt)
    lat = latfmt.prepare(ins[0])
    assert orc.lattice_beam(lat)[0] == np.inf
    for limit, hang in (("--max-arcs=2", False), ("--max-arcs=1", True)):
        out = str(tmp_path / "out.txt")
        rc, _, err = run("lattice-prune-dyn-beam", limit, "ark:" + txt, "ark,t:" + out)
        assert rc == 0, err
        got = latfmt.read_text_archive(out)[0]
        assert got.S == 3 and got.A == 2 and sorted(got.ilabel.tolist()) == [5, 8]
        assert ("no complete path" in err) == hang
        if not hang:
            arcs, states, beam = orc.prune_dyn_beam(lat, 2, 2 ** 31 - 1)
            assert arcs.tolist() == [0, 2] and states.tolist() == [0, 1, 3] and beam == np.inf


@pytest.mark.gpu
def test_tools_on_a_multi_lattice_archive(tmp_path):
    """Input order is kept, empty lattices still produce a record, batching does not change the
    output, binary input gives the same result as text input, scales reach the GPU."""
    ns = np.array([30, 1, 17, 64, 5], np.int32)
    fo = np.array([3, 1, 2, 3, 1], np.int32)
    arrays = klat_amd.synth(4242, ns, fo)
    txt = str(tmp_path / "in.txt")
    with open(txt, "w") as f:
        for u in range(len(ns)):
            s0, a0, a1 = arrays["state_off"][u], arrays["arc_off"][u], arrays["arc_off"][u + 1]
            f.write("utt%07d\n" % u)
            for i in range(a0, a1):
                s, d = arrays["src"][i], arrays["dst"][i]
                dur = arrays["state_time"][s0 + d] - arrays["state_time"][s0 + s]
                f.write("%d\t%d\t%d\t%.9g,%.9g,%s\n" % (s, d, arrays["ilabel"][i], arrays["graph"][i], arrays["acoustic"][i],
                                                        "_".join(["7"] * dur)))
            f.write("%d\n\n" % (ns[u] - 1))
            if u == 2:
                f.write("empty_utt\n\n")
    args = ["--acoustic-scale=0.1", "--insertion-penalty=0.25"]
    rc, out1, err = run("lattice-word-index-segment", *args, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = out1.decode().strip("\n").split("\n")
    assert [l.split()[0] if l.split() else "" for l in lines][:4] == ["utt0000000", "utt0000001", "utt0000002", "empty_utt"]
    assert lines[3].strip() == "empty_utt" and lines[1].strip() == "utt0000001"  # no arcs -> empty record
    rc, out2, err = run("lattice-word-index-segment", *args, "--batch-size=1", "ark:" + txt, "ark,t:-")
    assert rc == 0 and out2 == out1
    b = str(tmp_path / "in.ark")
    assert run("klat-lattice-copy", "ark:" + txt, "ark:" + b)[0] == 0
    rc, out3, err = run("lattice-word-index-segment", *args, "ark:" + b, "ark,t:-")
    assert rc == 0 and out3 == out1
    # against the oracle for utterance 3
    lat = [l for l in latfmt.read_text_archive(txt) if l.key == "utt0000003"][0]
    sl = orc.scale_and_penalize(lat, 1.0, 0.1, 0.25)
    times, _, _ = orc.state_times(sl)
    alpha, beta, total, _, _ = orc.alphas_betas(sl)
    want = orc.segment_index(sl, times, alpha, beta, total)
    _, got = latfmt.parse_tuple_line([l for l in lines if l.startswith("utt0000003")][0], 3)
    assert sorted(g[:3] for g in got) == sorted(w[:3] for w in want)
    wk = {w[:3]: w[3] for w in want}
    assert all(abs(g[3] - wk[g[:3]]) < 1e-5 for g in got)


@pytest.mark.gpu
def test_tools_on_a_long_lattice(tmp_path):
    """A 9 000-state lattice (26 997 arcs) between two short ones, through the tools: the segment index takes it in pieces
    (more than 16 384 arcs: cut where the source states' time changes, the pieces' sorted runs merged when the records are
    fetched), the frame posteriors in pieces of ~6 144 states (more than 8 192 states).  Order, keys and values against
    the oracle; the same records with one rank per process group member (--devices is exercised in test_gpu_dist.py)."""
    ns = np.array([40, 9000, 25], np.int32)
    fo = np.array([2, 3, 3], np.int32)
    arrays = klat_amd.synth(20260051, ns, fo)
    txt = str(tmp_path / "in.txt")
    _write_text_archive(txt, arrays, ns)
    args = ["--acoustic-scale=0.2"]
    rc, out, err = run("lattice-word-index-segment", *args, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    lines = out.decode().strip("\n").split("\n")
    rc, pout, err = run("lattice-to-word-frame-post", *args, "ark:" + txt, "ark,t:-")
    assert rc == 0, err
    plines = pout.decode().strip("\n").split("\n")
    lats = latfmt.read_text_archive(txt)
    assert [l.key for l in lats] == ["utt%07d" % u for u in range(3)] and len(lines) == 3 and len(plines) == 3
    for u, lat in enumerate(lats):
        sl = orc.scale_and_penalize(lat, 1.0, 0.2, 0.0)
        times, T, _ = orc.state_times(sl)
        alpha, beta, total, _, _ = orc.alphas_betas(sl)
        want = orc.segment_index(sl, times, alpha, beta, total)
        key, got = latfmt.parse_tuple_line(lines[u], 3)
        assert key == lat.key and len(got) == len(want)
        wk = {w[:3]: w[3] for w in want}
        assert all(g[:3] in wk and abs(g[3] - wk[g[:3]]) < 1e-5 for g in got)
        assert all(x[3] >= y[3] for x, y in zip(got[:-1], got[1:]))  # (as printed: seven significant digits)
        wantf, _ = orc.frame_post(sl, times, T, alpha, beta, total)
        key, frames = latfmt.parse_posterior_line(plines[u])
        assert key == lat.key and len(frames) == T
        for f, w in zip(frames, wantf):
            assert sorted(l for l, _ in f) == sorted(int(l) for l, _ in w)
            wd = {int(l): float(p) for l, p in w}
            assert all(abs(p - wd[l]) < 1e-5 * max(1.0, abs(wd[l])) for l, p in f)


@pytest.mark.gpu
def test_prune_dyn_beam_tool_on_a_skewed_mix(tmp_path):
    """BASELINE configs[4], first half, through the tool: lattice-prune-dyn-beam over a skewed mix
    (sizes log-uniform in [200, 6 000] states, fanout in {2,3,4,6}; one Viterbi sweep of the whole
    batch decides every iteration of the reference's loop, latbin/lattice-prune-dyn-beam.cc:169-186).
    Size-independent properties on every lattice, the oracle's literal loop on the small ones."""
    rng = np.random.RandomState(4)
    U = 36
    ns = np.exp(rng.uniform(np.log(200), np.log(6000), U)).astype(np.int32)
    fo = rng.choice([2, 3, 4, 6], U).astype(np.int32)
    arrays = klat_amd.synth(20260004, ns, fo)
    txt, out = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    with open(txt, "w") as f:
        latfmt.write_synth_text(arrays, f, ["u%02d" % i for i in range(U)])
    max_arcs = 900
    rc, _, err = run("lattice-prune-dyn-beam", "--acoustic-scale=0.1", "--max-arcs=%d" % max_arcs, "ark:" + txt, "ark,t:" + out)
    assert rc == 0, err
    ins, outs = latfmt.read_text_archive(txt), latfmt.read_text_archive(out)
    assert [o.key for o in outs] == [i.key for i in ins]
    checked = 0
    for lat, got in zip(ins, outs):
        assert 1 <= got.A <= lat.A and got.S <= lat.S
        # within the limit -- or pruned down to (nearly) the best path when even that is longer than the
        # limit: the reference's loop gives up below --min-beam (latbin/lattice-prune-dyn-beam.cc:169-186)
        assert got.A <= max_arcs or got.A <= 1.05 * (got.S - 1)
        # still one connected lattice from the start to a final state, every arc forward
        src = got.src()
        assert np.all(got.dst > src)
        reach = np.zeros(got.S, bool)
        reach[0] = True
        for s, d in zip(src, got.dst):
            reach[d] |= reach[s]
        assert reach.all() and np.isfinite(got.final_graph[-1])
        # its arcs are arcs of the input (weights untouched: the tool writes the stored weights back)
        by_label = {}
        for g, a, l in zip(lat.graph, lat.acoustic, lat.ilabel):
            by_label.setdefault(int(l), []).append((float(g), float(a)))
        for g, a, l in zip(got.graph, got.acoustic, got.ilabel):  # text output carries ~7 significant digits
            assert any(abs(g - g0) <= 1e-5 * max(1.0, abs(g0)) and abs(a - a0) <= 1e-5 * max(1.0, abs(a0)) for g0, a0 in by_label.get(int(l), []))
        if lat.S <= 700:
            arcs, states, beam = orc.prune_dyn_beam(orc.scale_and_penalize(latfmt.prepare(lat), 1.0, 0.1, 0.0), max_arcs, 2 ** 31 - 1)
            assert got.A == len(arcs) and got.S == len(states), (lat.key, got.A, len(arcs))
            checked += 1
    assert checked >= 5


# ------------------------------------------------------------------ a5: TopSortCompactLatticeIfNeeded under test
def _write_permuted_text_archive(path, arrays, ns, seed, cyclic_utt=None):
    """The lattices of `arrays` with their state ids permuted: the start state is no longer 0, arcs lead from high ids to
    low ones, the arcs of a state are scattered over the entry (the text format allows it, kwsbin2/egs/lattice.ark.txt:2,5;
    the start state is the source of the first line).  `cyclic_utt`: that lattice gets an arc back from its last state to
    its start state -- no topological order exists."""
    rng = np.random.RandomState(seed)
    perms = []
    with open(path, "w") as f:
        for u in range(len(ns)):
            S = int(ns[u])
            s0, a0, a1 = arrays["state_off"][u], arrays["arc_off"][u], arrays["arc_off"][u + 1]
            perm = rng.permutation(S)
            if S > 1 and perm[0] == 0:
                perm[[0, 1]] = perm[[1, 0]]
            perms.append(perm)
            lines = []
            for i in range(a0, a1):
                s, d = arrays["src"][i], arrays["dst"][i]
                dur = arrays["state_time"][s0 + d] - arrays["state_time"][s0 + s]
                lines.append((int(s), "%d\t%d\t%d\t%.9g,%.9g,%s\n" % (perm[s], perm[d], arrays["ilabel"][i], arrays["graph"][i], arrays["acoustic"][i],
                                                                     "_".join(["7"] * dur))))
            if cyclic_utt == u:
                lines.append((S - 1, "%d\t%d\t%d\t%.9g,%.9g,%s\n" % (perm[S - 1], perm[0], 5, 1.0, 1.0, "7")))
            order = rng.permutation(len(lines))
            first = [k for k in order if lines[k][0] == 0][:1]  # an arc out of the start state leads the entry
            f.write("utt%07d\n" % u)
            for k in first + [k for k in order if k not in first]:
                f.write(lines[k][1])
            f.write("%d\n\n" % perm[S - 1])
    return perms


def _a5_inputs(tmp_path):
    ns = np.array([30, 2, 45, 12, 200], np.int32)
    fo = np.array([3, 1, 2, 4, 3], np.int32)
    arrays = klat_amd.synth(31337, ns, fo)
    srt, per = str(tmp_path / "sorted.txt"), str(tmp_path / "permuted.txt")
    _write_text_archive(srt, arrays, ns)
    _write_permuted_text_archive(per, arrays, ns, 5)
    return arrays, ns, srt, per


def _oracle_segment_records(lat):
    lat = latfmt.prepare(lat)
    times, _, _ = orc.state_times(lat)
    alpha, beta, total, _, _ = orc.alphas_betas(lat)
    return sorted(orc.segment_index(lat, times, alpha, beta, total))


def test_top_sort_renumbers_unsorted_lattices_and_refuses_cycles(tmp_path):
    """TopSortCompactLatticeIfNeeded (latbin/lattice-to-word-frame-post.cc:80, kwsbin2/lattice-word-index-segment.cc:61;
    host/lattice-functions.cc) on lattices that NEED it -- start state not 0, arcs from high ids to low, a state's arcs
    scattered -- through klat-lattice-copy --top-sort, text and binary: the result is top-sorted from state 0 and is the
    same lattice (the oracle's segment index of it equals that of the sorted twin: no state id survives into it); the
    archive scanner calls the permuted entries unsorted and the sorted ones sorted; a cyclic lattice ends the run with the
    reference's exit code -1 (:143-146) and Kaldi's message."""
    arrays, ns, srt, per = _a5_inputs(tmp_path)
    want = [_oracle_segment_records(l) for l in latfmt.read_text_archive(srt)]
    bin_per = str(tmp_path / "permuted.ark")
    rc, _, err = run("klat-lattice-copy", "ark:" + per, "ark:" + bin_per)  # (no sorting: the binary twin keeps the permuted ids)
    assert rc == 0, err
    for path, kind in ((per, "t"), (bin_per, "b"), (srt, "t")):
        rc, out, err = run("klat-archive-scan", "ark:" + path)
        assert rc == 0, err
        rows = [l.split() for l in out.decode().splitlines()]
        assert [r[3] for r in rows] == [kind] * len(ns)
        assert [r[6] for r in rows] == (["s"] * len(ns) if path == srt else ["u"] * len(ns))
    for src in (per, bin_per):
        for mode in ("ark,t:", "ark:"):
            mid = str(tmp_path / "sorted_again")
            rc, _, err = run("klat-lattice-copy", "--top-sort=true", "ark:" + src, mode + mid)
            assert rc == 0, err
            rc, out, err = run("klat-archive-scan", "ark:" + mid)
            assert rc == 0 and [l.split()[6] for l in out.decode().splitlines()] == ["s"] * len(ns), err
            back = str(tmp_path / "sorted_again.txt")
            rc, _, err = run("klat-lattice-copy", "ark:" + mid, "ark,t:" + back)
            assert rc == 0, err
            got = latfmt.read_text_archive(back)
            assert [g.key for g in got] == ["utt%07d" % u for u in range(len(ns))]
            for g, w, S, A in zip(got, want, ns, np.diff(arrays["arc_off"])):
                assert g.S == S and g.A == A
                src_of = np.repeat(np.arange(g.S), np.diff(g.arc_begin))
                assert np.all(g.dst > src_of)  # every arc goes to a higher state, the start state is 0
                recs = _oracle_segment_records(g)
                assert [r[:3] for r in recs] == [r[:3] for r in w]
                assert max(abs(a[3] - b[3]) for a, b in zip(recs, w)) < 1e-9
    cyc = str(tmp_path / "cyclic.txt")
    _write_permuted_text_archive(cyc, arrays, ns, 6, cyclic_utt=2)
    bin_cyc = str(tmp_path / "cyclic.ark")
    rc, _, err = run("klat-lattice-copy", "ark:" + cyc, "ark:" + bin_cyc)
    assert rc == 0, err
    for src in (cyc, bin_cyc):
        rc, out, err = run("klat-lattice-copy", "--top-sort=true", "ark:" + src, "ark,t:-")
        assert rc == 255 and "Topological sorting of state-level lattice failed" in err


@pytest.mark.gpu
@pytest.mark.parametrize("tool,nint", [("lattice-word-index-segment", 3), ("lattice-word-index-position", 4), ("lattice-to-word-frame-post", 0)])
def test_tools_top_sort_what_needs_it(tmp_path, tool, nint):
    """The tools on the permuted twins (text: the object reader; binary: ReadFlat finds the entry unsorted and falls
    back to the object form, host/bin/sharded-index.h) give the sorted twin's output; a cyclic entry ends them with -1
    (latbin/lattice-to-word-frame-post.cc:80,143-146)."""
    arrays, ns, srt, per = _a5_inputs(tmp_path)
    bin_per = str(tmp_path / "permuted.ark")
    rc, _, err = run("klat-lattice-copy", "ark:" + per, "ark:" + bin_per)
    assert rc == 0, err
    args = ["--acoustic-scale=0.1"]
    rc, out, err = run(tool, *args, "ark:" + srt, "ark,t:-")
    assert rc == 0, err
    want = out.decode().splitlines()
    assert len(want) == len(ns)
    for src in (per, bin_per):
        rc, out, err = run(tool, *args, "ark:" + src, "ark,t:-")
        assert rc == 0, err
        got = out.decode().splitlines()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            if nint:
                (gk, gr), (wk, wr) = latfmt.parse_tuple_line(g, nint), latfmt.parse_tuple_line(w, nint)
                assert gk == wk and sorted(r[:nint] for r in gr) == sorted(r[:nint] for r in wr)
                wv = {r[:nint]: r[nint] for r in wr}
                assert all(abs(r[nint] - wv[r[:nint]]) < 1e-6 for r in gr)
            else:
                (gk, gf), (wk, wf) = latfmt.parse_posterior_line(g), latfmt.parse_posterior_line(w)
                assert gk == wk and len(gf) == len(wf)
                for a, b in zip(gf, wf):
                    assert sorted(l for l, _ in a) == sorted(l for l, _ in b)
                    bv = dict(b)
                    assert all(abs(p - bv[l]) < 1e-5 for l, p in a)
    cyc = str(tmp_path / "cyclic.txt")
    _write_permuted_text_archive(cyc, arrays, ns, 6, cyclic_utt=3)
    bin_cyc = str(tmp_path / "cyclic.ark")
    rc, _, err = run("klat-lattice-copy", "ark:" + cyc, "ark:" + bin_cyc)
    assert rc == 0, err
    for src in (cyc, bin_cyc):  # (a failing reader stage must end the run, not leave the other stages waiting for it)
        rc, out, err = run(tool, *args, "ark:" + src, "ark,t:-", timeout=60)
        assert rc == 255 and "Topological sorting of state-level lattice failed" in err, (rc, err[-300:])
