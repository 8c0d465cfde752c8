"""The archive scanner every rank of a --devices run relies on (host/archive-index.cc): a plain file is
mapped and walked for the entry boundaries WITHOUT building a lattice -- a text entry ends at its blank
line, a binary one is walked state by state over its variable-length weights -- and an entry parsed from
its byte range must be the lattice the sequential reader builds.  CPU only (klat-archive-scan,
klat-lattice-copy)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BIN = os.path.join(ROOT, "kaldi-lattice-utils_amd", "bin")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd  # noqa: E402
from test_host_tools import _write_text_archive  # noqa: E402


def _run(tool, *args):
    p = subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, timeout=300)
    return p.returncode, p.stdout.decode(), p.stderr.decode(errors="replace")


def _make(tmp_path):
    rng = np.random.RandomState(3)
    ns = np.concatenate([[1, 2], rng.randint(2, 150, size=30)]).astype(np.int32)
    fo = rng.choice([1, 2, 3, 5], size=len(ns)).astype(np.int32)
    arrays = klat_amd.synth(606, ns, fo)
    txt, binary = str(tmp_path / "l.ark.txt"), str(tmp_path / "l.ark")
    _write_text_archive(txt, arrays, ns)
    rc, _, err = _run("klat-lattice-copy", "ark:" + txt, "ark:" + binary)
    assert rc == 0, err
    return arrays, ns, txt, binary


def test_scan_finds_every_entry_and_its_bytes_parse_to_the_same_lattice(tmp_path):
    arrays, ns, txt, binary = _make(tmp_path)
    want_arcs = np.diff(arrays["arc_off"])
    for path, kind in ((txt, "t"), (binary, "b")):
        rc, out, err = _run("klat-archive-scan", "ark:" + path)
        assert rc == 0, err
        rows = [l.split() for l in out.splitlines()]
        assert [r[0] for r in rows] == ["utt%07d" % u for u in range(len(ns))]
        assert all(r[3] == kind for r in rows)
        assert [int(r[4]) for r in rows] == ns.tolist() and [int(r[5]) for r in rows] == want_arcs.tolist()
        begins, ends = [int(r[1]) for r in rows], [int(r[2]) for r in rows]
        size = os.path.getsize(path)
        assert all(b < e for b, e in zip(begins, ends)) and ends[-1] <= size
        assert all(ends[i] <= begins[i + 1] for i in range(len(rows) - 1))  # (the next key sits between two values)
        # the byte ranges alone (--parse=false) are the same: the scan does not depend on the parser
        rc, out2, err = _run("klat-archive-scan", "--parse=false", "ark:" + path)
        assert rc == 0 and [l.split()[:4] for l in out2.splitlines()] == [r[:4] for r in rows]


def test_scan_refuses_a_truncated_binary_entry_and_a_pipe(tmp_path):
    arrays, ns, txt, binary = _make(tmp_path)
    data = open(binary, "rb").read()
    cut = str(tmp_path / "cut.ark")
    open(cut, "wb").write(data[: len(data) - 7])
    rc, out, err = _run("klat-archive-scan", "ark:" + cut)
    assert rc != 0 and "Failed reading lattice" in err
    rc, out, err = _run("klat-archive-scan", "ark:cat %s|" % binary)
    assert rc != 0 and "not a plain file" in err


def test_scan_refuses_negative_header_string_lengths(tmp_path):
    """The fst-type / arc-type strings of an OpenFst header carry int32 lengths: a negative one must end in the
    reader's error, never in a pointer that walks backwards through the mapping (scanner and flat reader alike)."""
    arrays, ns, txt, binary = _make(tmp_path)
    data = bytearray(open(binary, "rb").read())
    magic = (2125659606).to_bytes(4, "little")
    at = data.find(magic)
    assert at > 0
    for which, off in (("fst type", at + 4), ("arc type", at + 4 + 4 + 6)):  # "vector" is six bytes long
        broken = bytearray(data)
        assert which != "fst type" or int.from_bytes(broken[off:off + 4], "little") == 6
        broken[off:off + 4] = (-1).to_bytes(4, "little", signed=True)
        path = str(tmp_path / "neg.ark")
        open(path, "wb").write(broken)
        rc, out, err = _run("klat-archive-scan", "ark:" + path)
        assert rc != 0 and "Failed reading lattice" in err, (which, rc, err)
