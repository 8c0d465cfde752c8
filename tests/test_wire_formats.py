"""Wire formats either side of the path against byte fixtures that were NOT
produced by this repo's writers: tests/golden/wire_*.hex are assembled by hand -- the tuple tables from the reference's
holder source (util/basic-tuple-vector-holder.h:141-181: "\\0B", WriteBasicType(int32 size), every field through
WriteBasicType), the Posterior and the binary CompactLattice from the layouts of SURVEY.md Appendix B.5 / B.4
([UPSTREAM-derived]: that source is not in the reference tree).  A bug mirrored in this repo's writer AND reader passes a
round trip; it does not pass these.  CPU only (klat-table-write, klat-lattice-copy, klat-archive-scan)."""
import os
import subprocess

import numpy as np
import pytest

import latfmt

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BIN = os.path.join(ROOT, "kaldi-lattice-utils_amd", "bin")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _hex(name):
    """bytes and the '# text twin:' lines of a fixture"""
    raw, twin, in_twin = bytearray(), [], False
    for line in open(os.path.join(GOLDEN, name)):
        body, _, comment = line.partition("#")
        if not body.strip():
            c = comment.strip()
            if c.startswith("text twin"):
                in_twin = True
                c = c.split(":", 1)[1].strip()
                if c:
                    twin.append(c)
            elif in_twin and line.startswith("#   ") and c:
                twin.append(c)
            else:
                in_twin = False
            continue
        in_twin = False
        raw += bytes.fromhex(body.replace(" ", ""))
    return bytes(raw), twin


def _run(tool, *args, stdin=None):
    p = subprocess.run([os.path.join(BIN, tool)] + list(args), input=stdin, capture_output=True, timeout=120)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


@pytest.mark.parametrize("kind,fixture", [("segment", "wire_segment_table.hex"), ("position", "wire_position_table.hex"),
                                          ("utterance", "wire_utterance_table.hex"), ("posterior", "wire_posterior.hex")])
def test_table_writers_emit_the_reference_holders_bytes(kind, fixture):
    want, twin = _hex(fixture)
    text = ("\n".join(twin) + "\n").encode()
    rc, out, err = _run("klat-table-write", kind, "ark:-", stdin=text)
    assert rc == 0, err
    assert out == want, "binary %s table differs from the hand-assembled fixture:\n got  %s\n want %s" % (kind, out.hex(" "), want.hex(" "))
    # text mode: fields end with a space, tuples are separated by "; ", the entry ends with a newline
    # (util/basic-tuple-vector-holder.h:163-174; the Posterior: one bracket group per frame, Appendix B.5)
    rc, out, err = _run("klat-table-write", kind, "ark,t:-", stdin=text)
    assert rc == 0, err
    assert out.decode().splitlines() == [t + " " for t in twin]
    assert out.endswith(b" \n")


def test_binary_compact_lattice_reader_and_writer_against_the_hand_assembled_entry(tmp_path):
    want, twin = _hex("wire_compact_lattice.hex")
    assert twin[0] == "b1" and len(twin) == 5
    # the READER on bytes this repo did not write: sequential reader, mapped scanner + flat reader (klat-archive-scan
    # checks that the two agree), then the lattice itself
    ark = str(tmp_path / "b1.ark")
    open(ark, "wb").write(want)
    rc, out, err = _run("klat-archive-scan", "ark:" + ark)
    assert rc == 0, err
    assert out.decode().split() == ["b1", "3", str(len(want)), "b", "3", "3", "s"]
    txt = str(tmp_path / "b1.txt")
    rc, _, err = _run("klat-lattice-copy", "ark:" + ark, "ark,t:" + txt)
    assert rc == 0, err
    (lat,) = latfmt.read_text_archive(txt)
    assert lat.key == "b1" and lat.S == 3 and lat.A == 3
    assert lat.arc_begin.tolist() == [0, 2, 3, 3] and lat.dst.tolist() == [1, 2, 2] and lat.ilabel.tolist() == [3, 0, 4]
    assert lat.graph.tolist() == [0.5, 2.0, 0.25] and lat.acoustic.tolist() == [1.25, 0.0, 0.75] and lat.dur.tolist() == [2, 3, 1]
    assert np.isinf(lat.final_graph[0]) and np.isinf(lat.final_graph[1]) and float(lat.final_graph[2]) == 1.5 and float(lat.final_acoustic[2]) == 0.0
    # the WRITER: the text twin through klat-lattice-copy must be the fixture byte for byte, the properties word aside
    src = str(tmp_path / "twin.txt")
    open(src, "w").write("b1\n" + "\n".join(t.replace(" ", "\t", 3) if t[0] != "2" else t.replace(" ", "\t", 1) for t in twin[1:]) + "\n\n")
    rc, out, err = _run("klat-lattice-copy", "ark:" + src, "ark:-")
    assert rc == 0, err
    prop = want.index(b"compactlattice44") + 16 + 8  # behind the arc type: version, flags, then the 8 bytes of properties
    mask = lambda b: b[:prop] + b"\0" * 8 + b[prop + 8:]
    assert len(out) == len(want) and mask(out) == mask(want), "binary CompactLattice differs:\n got  %s\n want %s" % (out.hex(" "), want.hex(" "))
