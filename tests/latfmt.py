"""Test-side readers for the Kaldi text formats on either side of the hot path (SURVEY.md
Appendix B.2/B.3/B.5/B.6).  Independent of the product's C++ reader so that the two check
each other.  Test infrastructure only."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc  # noqa: E402


def _parse_weight(tok):
    """'g,a,t1_t2_..' -> (graph, acoustic, dur).  Missing pieces default to One."""
    parts = tok.split(",")
    g = float(parts[0]) if len(parts) > 0 and parts[0] != "" else 0.0
    a = float(parts[1]) if len(parts) > 1 and parts[1] != "" else 0.0
    dur = 0
    if len(parts) > 2 and parts[2] != "":
        dur = len(parts[2].split("_"))
    return g, a, dur


def read_text_archive(path):
    """Yields orc.Lat per entry.  Accepts CompactLattice text (B.2: 'src dst label
    g,a,tids') and Lattice text (B.3: 'src dst ilabel olabel [g,a]'); the latter goes through
    convert_lattice() below, the test-side restatement of Kaldi's ConvertLattice / Factor."""
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    out = []
    while i < len(lines):
        if lines[i].strip() == "":
            i += 1
            continue
        key = lines[i].strip()
        i += 1
        arcs, finals, ids = [], {}, {}
        raw = []  # Lattice-format arcs (s, d, ilabel, olabel, g, a)
        start = None

        def sid(x):
            x = int(x)
            if x not in ids:
                ids[x] = x
            return x

        while i < len(lines) and lines[i].strip() != "":
            t = lines[i].split()
            i += 1
            if len(t) <= 2 and (len(t) == 1 or "," in t[1] or _isnum(t[1])):
                # final state line: "<state> [weight]"
                s = sid(t[0])
                if start is None:
                    start = s
                if len(t) == 1:
                    finals[s] = (0.0, 0.0, 0)
                else:
                    finals[s] = _parse_weight(t[1])
                continue
            s, d = sid(t[0]), sid(t[1])
            if start is None:
                start = s
            rest = t[2:]
            if len(rest) >= 2 and "," not in rest[1] and _isint(rest[1]):
                # Lattice format: ilabel olabel [weight]
                il, ol = int(rest[0]), int(rest[1])
                g, a, _ = _parse_weight(rest[2]) if len(rest) > 2 else (0.0, 0.0, 0)
                raw.append((s, d, il, ol, g, a))
            else:
                lab = int(rest[0])
                g, a, dur = _parse_weight(rest[1]) if len(rest) > 1 else (0.0, 0.0, 0)
                arcs.append((s, d, lab, lab, g, a, dur))
        S = (max(ids) + 1) if ids else 0
        if raw:
            assert not arcs, "an entry is in one format or the other"
            S, arcs, finals, start = convert_lattice(S, raw, finals, start)
        lat = orc.lat_from_arcs(S, arcs, finals, key=key, start=start if start is not None else -1)
        out.append(lat)
    return out


def convert_lattice(S, raw, finals, start):
    """ConvertLattice(lat, &clat) [upstream Kaldi, fstext/lattice-utils-inl.h] = Factor
    [fstext/factor-inl.h] + per-arc rewrite, as a CompactLattice reader applies it to a Lattice
    entry (kwsbin2/README.md:176).  A state inside a linear chain -- not initial, not final, one
    arc in, one arc out, no output label on the arc out -- disappears; the arc entering the chain
    takes the float32 sums of the weights (left to right) and the concatenated non-zero input labels
    as its string; the word is the first arc's olabel.  Kept states are renumbered in order."""
    import numpy as np
    out_arcs = [[] for _ in range(S)]
    nin = [0] * S
    for (s, d, il, ol, g, a) in raw:
        out_arcs[s].append((d, il, ol, g, a))
        nin[d] += 1
    remove = [s != start and s not in finals and nin[s] == 1 and len(out_arcs[s]) == 1 and out_arcs[s][0][2] == 0
              for s in range(S)]
    new_id, n = {}, 0
    for s in range(S):
        if not remove[s]:
            new_id[s] = n
            n += 1
    arcs = []
    for s in range(S):
        if remove[s]:
            continue
        for (d, il, ol, g, a) in out_arcs[s]:
            g, a = np.float32(g), np.float32(a)
            dur = 1 if il != 0 else 0
            while remove[d]:
                d2, il2, _, g2, a2 = out_arcs[d][0]
                g, a = np.float32(g + np.float32(g2)), np.float32(a + np.float32(a2))
                dur += 1 if il2 != 0 else 0
                d = d2
            arcs.append((new_id[s], new_id[d], ol, ol, float(g), float(a), dur))
    new_finals = {new_id[s]: w for s, w in finals.items()}
    return n, arcs, new_finals, (new_id[start] if start is not None else None)


def _isint(s):
    try:
        int(s)
        return True
    except ValueError:
        return False


def _isnum(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def prepare(lat):
    """Start state -> 0 and topological order, as TopSortCompactLatticeIfNeeded does
    (only when needed).  Returns the lattice the sweep runs on."""
    start = lat.extra.get("start", 0)
    if lat.S == 0:
        return lat
    if start == 0 and orc.is_top_sorted(lat):
        return lat
    out, _ = orc.top_sort(lat, start)
    return out


def parse_tuple_line(line, ints):
    """'key a b c x ; a b c x ...' -> (key, [(ints..., float)])."""
    toks = line.strip().split(None, 1)
    key = toks[0]
    recs = []
    if len(toks) > 1:
        for chunk in toks[1].split(";"):
            f = chunk.split()
            if not f:
                continue
            assert len(f) == ints + 1, f
            recs.append(tuple(int(v) for v in f[:ints]) + (float(f[ints]),))
    return key, recs


def parse_posterior_line(line):
    """'key [ l p l p ] [ ] ...' -> (key, [[(label, logp)]])."""
    toks = line.strip().split()
    key = toks[0]
    frames, cur = [], None
    for t in toks[1:]:
        if t == "[":
            cur = []
        elif t == "]":
            frames.append([(int(cur[i]), float(cur[i + 1])) for i in range(0, len(cur), 2)])
            cur = None
        else:
            cur.append(t)
    return key, frames


def write_synth_text(arrays, f, keys):
    """Writes the lattices of a klat_amd.synth() batch as a text CompactLattice archive (arc
    durations from the state times, transition id 7)."""
    for u, key in enumerate(keys):
        s0, a0, a1 = arrays["state_off"][u], arrays["arc_off"][u], arrays["arc_off"][u + 1]
        n = arrays["state_off"][u + 1] - s0
        f.write("%s\n" % key)
        for i in range(a0, a1):
            s, d = arrays["src"][i], arrays["dst"][i]
            dur = arrays["state_time"][s0 + d] - arrays["state_time"][s0 + s]
            f.write("%d\t%d\t%d\t%.9g,%.9g,%s\n" % (s, d, arrays["ilabel"][i], arrays["graph"][i], arrays["acoustic"][i],
                                                    "_".join(["7"] * dur)))
        f.write("%d\n\n" % (n - 1))


def logsumexp(xs):
    xs = list(xs)
    if not xs:
        return -math.inf
    m = max(xs)
    if m == -math.inf:
        return m
    return m + math.log(sum(math.exp(x - m) for x in xs))


def brute_force_paths(lat):
    """Enumerate every start->final path of a tiny lattice (the reference's own test idiom,
    fstext/expand-subpaths-test-utils.h:10-52).  Returns list of (arc index list, loglik)."""
    paths = []

    def rec(s, arcs, ll):
        fg, fa = float(lat.final_graph[s]), float(lat.final_acoustic[s])
        if not (math.isinf(fg) and math.isinf(fa)):
            paths.append((list(arcs), ll - (fg + fa)))
        for i in range(int(lat.arc_begin[s]), int(lat.arc_begin[s + 1])):
            arcs.append(i)
            rec(int(lat.dst[i]), arcs, ll - (float(lat.graph[i]) + float(lat.acoustic[i])))
            arcs.pop()

    if lat.S:
        rec(0, [], 0.0)
    return paths
