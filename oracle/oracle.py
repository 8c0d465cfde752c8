"""ctypes wrapper over oracle/libklat_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  See klat_oracle.c for the reference citations.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libklat_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.kor_log_add.restype = C.c_double
        _LIB.kor_log_add.argtypes = [C.c_double, C.c_double]
        _LIB.kor_log_sub.restype = C.c_double
        _LIB.kor_log_sub.argtypes = [C.c_double, C.c_double]
        _LIB.kor_alphas_betas.restype = C.c_double
        _LIB.kor_betas.restype = C.c_double
        _LIB.kor_segment_index.restype = C.c_int64
        _LIB.kor_position_index.restype = C.c_int64
        _LIB.kor_frame_post.restype = C.c_int64
        _LIB.kor_position_post.restype = C.c_int64
        _LIB.kor_length_dist.restype = C.c_int64
        _LIB.kor_utterance_index.restype = C.c_int64
        _LIB.kor_lattice_beam.restype = C.c_double
        _LIB.kor_disambiguate_length.restype = C.c_int32
        _LIB.kor_free.argtypes = [C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class Lat:
    """One lattice in stored (CSR-by-source) order; float32 costs, int32 ids."""

    arc_begin: np.ndarray  # int64 [S+1]
    dst: np.ndarray  # int32 [A]
    ilabel: np.ndarray
    olabel: np.ndarray
    graph: np.ndarray  # float32 [A]
    acoustic: np.ndarray
    dur: np.ndarray  # int32 [A]  length of the transition-id string
    final_graph: np.ndarray  # float32 [S]; +inf = non-final
    final_acoustic: np.ndarray
    final_dur: np.ndarray  # int32 [S]
    key: str = ""
    extra: dict = field(default_factory=dict)

    @property
    def S(self):
        return len(self.arc_begin) - 1

    @property
    def A(self):
        return len(self.dst)

    def src(self):
        return np.repeat(np.arange(self.S, dtype=np.int32), np.diff(self.arc_begin)).astype(np.int32)

    def copy(self):
        return Lat(*[x.copy() if isinstance(x, np.ndarray) else x for x in
                     (self.arc_begin, self.dst, self.ilabel, self.olabel, self.graph, self.acoustic,
                      self.dur, self.final_graph, self.final_acoustic, self.final_dur)], key=self.key)


def lat_from_arcs(S, arcs, finals, key="", start=0):
    """arcs: list of (src,dst,ilabel,olabel,graph,acoustic,dur) in file order; finals:
    dict state -> (graph, acoustic, dur).  Arcs are grouped by source keeping file order."""
    by_src = [[] for _ in range(S)]
    for a in arcs:
        by_src[a[0]].append(a)
    flat = [a for s in range(S) for a in by_src[s]]
    ab = np.zeros(S + 1, np.int64)
    for s in range(S):
        ab[s + 1] = ab[s] + len(by_src[s])
    fg = np.full(S, np.inf, np.float32)
    fa = np.full(S, np.inf, np.float32)
    fd = np.zeros(S, np.int32)
    for s, (g, a, d) in finals.items():
        fg[s], fa[s], fd[s] = g, a, d
    col = lambda i, dt: np.array([a[i] for a in flat], dtype=dt)
    lat = Lat(ab, col(1, np.int32), col(2, np.int32), col(3, np.int32), col(4, np.float32),
              col(5, np.float32), col(6, np.int32), fg, fa, fd, key=key)
    lat.extra["start"] = start
    return lat


def log_add(x, y):
    return lib().kor_log_add(x, y)


def scale_and_penalize(lat, graph_scale=1.0, acoustic_scale=1.0, insertion_penalty=0.0):
    """Returns a scaled copy (a2/a3)."""
    out = lat.copy()
    lib().kor_scale_and_penalize(
        C.c_int64(out.A), C.c_int64(out.S), _p(out.ilabel), _p(out.graph), _p(out.acoustic),
        _p(out.final_graph), _p(out.final_acoustic), C.c_float(graph_scale),
        C.c_float(acoustic_scale), C.c_float(insertion_penalty))
    return out


def is_top_sorted(lat):
    return bool(lib().kor_is_top_sorted(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst)))


def top_sort(lat, start=0):
    """Returns (renumbered lattice, order[old]=new); raises on cycles."""
    order = np.zeros(lat.S, np.int32)
    rc = lib().kor_top_sort(C.c_int32(lat.S), C.c_int32(start), _p(lat.arc_begin), _p(lat.dst), _p(order))
    if rc != 0:
        raise RuntimeError("lattice is cyclic (kor_top_sort rc=%d)" % rc)
    return renumber(lat, order), order


def renumber(lat, order):
    S = lat.S
    inv = np.zeros(S, np.int64)
    inv[order] = np.arange(S)
    counts = np.diff(lat.arc_begin)[inv]
    ab = np.zeros(S + 1, np.int64)
    ab[1:] = np.cumsum(counts)
    idx = np.concatenate([np.arange(lat.arc_begin[o], lat.arc_begin[o + 1]) for o in inv]) if lat.A else np.zeros(0, np.int64)
    idx = idx.astype(np.int64)
    out = Lat(ab, order[lat.dst[idx]].astype(np.int32), lat.ilabel[idx], lat.olabel[idx], lat.graph[idx],
              lat.acoustic[idx], lat.dur[idx], lat.final_graph[inv], lat.final_acoustic[inv],
              lat.final_dur[inv], key=lat.key)
    out.extra["arc_old"] = idx
    return out


def state_times(lat):
    times = np.zeros(lat.S, np.int32)
    utt = C.c_int32(0)
    warn = C.c_int32(0)
    rc = lib().kor_state_times(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.dur),
                               _p(lat.final_graph), _p(lat.final_acoustic), _p(lat.final_dur),
                               _p(times), C.byref(utt), C.byref(warn))
    if rc != 0:
        raise RuntimeError("kor_state_times rc=%d" % rc)
    return times, utt.value, warn.value


def alphas_betas(lat):
    alpha = np.zeros(lat.S, np.float64)
    beta = np.zeros(lat.S, np.float64)
    tf = C.c_double(0)
    tb = C.c_double(0)
    tot = lib().kor_alphas_betas(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.graph),
                                 _p(lat.acoustic), _p(lat.final_graph), _p(lat.final_acoustic),
                                 _p(alpha), _p(beta), C.byref(tf), C.byref(tb))
    return alpha, beta, tot, tf.value, tb.value


def betas(lat):
    beta = np.zeros(lat.S, np.float64)
    tot = lib().kor_betas(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.graph),
                          _p(lat.acoustic), _p(lat.final_graph), _p(lat.final_acoustic), _p(beta))
    return beta, tot


def utterance_index(lat, include=None, exclude=None):
    """kwsbin2/lattice-word-index-utterance.cc on a scaled, top-sorted lattice: [(label, logp)]
    with logp = log P(label occurs somewhere on the path), best first."""
    if lat.S == 0:
        return []
    beta, _ = betas(lat)
    inc, exc = _sets(include, exclude)
    cap = lat.A + 1
    ol = np.zeros(cap, np.int32)
    lp = np.zeros(cap, np.float64)
    m = lib().kor_utterance_index(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.olabel),
                                  _p(lat.graph), _p(lat.acoustic), _p(beta), _p(inc), C.c_int32(len(inc)),
                                  _p(exc), C.c_int32(len(exc)), C.c_int64(cap), _p(ol), _p(lp))
    assert m >= 0
    return [(int(ol[i]), float(lp[i])) for i in range(m)]


def arc_through(lat, alpha, beta):
    out = np.zeros(lat.A, np.float64)
    lib().kor_arc_through(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.graph),
                          _p(lat.acoustic), _p(alpha), _p(beta), _p(out))
    return out


def _sets(include, exclude):
    inc = np.array(sorted(set(include or [])), np.int32)
    exc = np.array(sorted(set(exclude or [])), np.int32)
    return inc, exc


def segment_index(lat, times, alpha, beta, total, include=None, exclude=None):
    inc, exc = _sets(include, exclude)
    n = max(lat.A, 1)
    ol, t0, t1 = (np.zeros(n, np.int32) for _ in range(3))
    lp = np.zeros(n, np.float64)
    m = lib().kor_segment_index(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.olabel),
                                _p(lat.graph), _p(lat.acoustic), _p(times), _p(alpha), _p(beta),
                                C.c_double(total), _p(inc), C.c_int32(len(inc)), _p(exc),
                                C.c_int32(len(exc)), _p(ol), _p(t0), _p(t1), _p(lp))
    return [(int(ol[i]), int(t0[i]), int(t1[i]), float(lp[i])) for i in range(m)]


def position_index(lat, times, state_len, alpha, beta, total, include=None, exclude=None):
    inc, exc = _sets(include, exclude)
    n = max(lat.A, 1)
    ol, pos, t0, t1 = (np.zeros(n, np.int32) for _ in range(4))
    lp = np.zeros(n, np.float64)
    m = lib().kor_position_index(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.olabel),
                                 _p(lat.graph), _p(lat.acoustic), _p(times), _p(state_len),
                                 _p(alpha), _p(beta), C.c_double(total), _p(inc),
                                 C.c_int32(len(inc)), _p(exc), C.c_int32(len(exc)), _p(ol), _p(pos),
                                 _p(t0), _p(t1), _p(lp))
    return [(int(ol[i]), int(pos[i]), int(t0[i]), int(t1[i]), float(lp[i])) for i in range(m)]


def frame_post(lat, times, total_frames, alpha, beta, total):
    """Returns (list per frame of [(label, float32 logp)], misaligned flag)."""
    src = lat.src()
    word = lat.ilabel != 0
    span = np.maximum(times[lat.dst] - times[src], 0)
    cap = int(span[word].sum()) + 1
    fb = np.zeros(total_frames + 1, np.int64)
    ol = np.zeros(cap, np.int32)
    lp = np.zeros(cap, np.float32)
    mis = C.c_int32(0)
    m = lib().kor_frame_post(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.ilabel),
                             _p(lat.graph), _p(lat.acoustic), _p(times), C.c_int32(total_frames),
                             _p(alpha), _p(beta), C.c_double(total), C.c_int64(cap), _p(fb), _p(ol),
                             _p(lp), C.byref(mis))
    assert m >= 0
    out = [[(int(ol[i]), np.float32(lp[i])) for i in range(fb[f], fb[f + 1])] for f in range(total_frames)]
    return out, bool(mis.value)


def position_post(lat, state_len, max_len, alpha, beta, total):
    """latbin/lattice-to-word-position-post.cc over an already length-disambiguated lattice.
    Returns a list per position 1..max_len of [(label, float32 logp)]."""
    cap = lat.A + 1
    pb = np.zeros(max_len + 1, np.int64)
    ol = np.zeros(cap, np.int32)
    lp = np.zeros(cap, np.float32)
    m = lib().kor_position_post(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.ilabel),
                                _p(lat.graph), _p(lat.acoustic), _p(np.ascontiguousarray(state_len, np.int32)),
                                C.c_int32(max_len), _p(alpha), _p(beta), C.c_double(total), C.c_int64(cap),
                                _p(pb), _p(ol), _p(lp))
    assert m >= 0
    return [[(int(ol[i]), np.float32(lp[i])) for i in range(pb[n], pb[n + 1])] for n in range(max_len)]


def length_dist(lat, state_len, alpha, total):
    """latbin/lattice-to-transcript-length-dist.cc over a length-disambiguated lattice (finals
    already scaled).  Returns [(length, float32 logp)] best first."""
    cap = lat.S + 1
    ol = np.zeros(cap, np.int32)
    lp = np.zeros(cap, np.float32)
    m = lib().kor_length_dist(C.c_int32(lat.S), _p(lat.final_graph), _p(lat.final_acoustic),
                              _p(np.ascontiguousarray(state_len, np.int32)), _p(alpha), C.c_double(total),
                              C.c_int64(cap), _p(ol), _p(lp))
    assert m >= 0
    return [(int(ol[i]), np.float32(lp[i])) for i in range(m)]


def disambiguate_length(lat, start=0, use_input=False):
    """a14.  Returns (new lattice, state_len[int32], max_len)."""
    L = lib()
    ns = C.c_int32(0)
    nab = C.POINTER(C.c_int64)()
    nd = C.POINTER(C.c_int32)()
    nao = C.POINTER(C.c_int64)()
    nso = C.POINTER(C.c_int32)()
    sl = C.POINTER(C.c_int32)()
    max_len = L.kor_disambiguate_length(C.c_int32(lat.S), C.c_int32(start), _p(lat.arc_begin),
                                        _p(lat.dst), _p(lat.ilabel), _p(lat.olabel),
                                        C.c_int(1 if use_input else 0), C.byref(ns), C.byref(nab),
                                        C.byref(nd), C.byref(nao), C.byref(nso), C.byref(sl))
    NS = ns.value
    if NS == 0:
        z = lambda dt: np.zeros(0, dt)
        return Lat(np.zeros(1, np.int64), z(np.int32), z(np.int32), z(np.int32), z(np.float32),
                   z(np.float32), z(np.int32), z(np.float32), z(np.float32), z(np.int32), key=lat.key), z(np.int32), 0
    ab = np.ctypeslib.as_array(nab, (NS + 1,)).copy()
    NA = int(ab[-1])
    dst = np.ctypeslib.as_array(nd, (max(NA, 1),))[:NA].copy()
    arc_old = np.ctypeslib.as_array(nao, (max(NA, 1),))[:NA].copy()
    st_old = np.ctypeslib.as_array(nso, (NS,)).copy()
    state_len = np.ctypeslib.as_array(sl, (NS,)).copy()
    for ptr in (nab, nd, nao, nso, sl):
        L.kor_free(C.cast(ptr, C.c_void_p))
    out = Lat(ab, dst, lat.ilabel[arc_old], lat.olabel[arc_old], lat.graph[arc_old],
              lat.acoustic[arc_old], lat.dur[arc_old], lat.final_graph[st_old],
              lat.final_acoustic[st_old], lat.final_dur[st_old], key=lat.key)
    out.extra["arc_old"] = arc_old
    out.extra["state_old"] = st_old
    return out, state_len, max_len


def arc_sort_olabel(lat):
    """ArcSort(&clat, OLabelCompare()) of latbin/lattice-best-path2.cc:103 (stable)."""
    arcs, finals = [], {}
    src = lat.src()
    for s in range(lat.S):
        idx = sorted(range(int(lat.arc_begin[s]), int(lat.arc_begin[s + 1])), key=lambda i: int(lat.olabel[i]))
        for i in idx:
            arcs.append((s, int(lat.dst[i]), int(lat.ilabel[i]), int(lat.olabel[i]), lat.graph[i], lat.acoustic[i], int(lat.dur[i])))
        if not (np.isinf(lat.final_graph[s]) and np.isinf(lat.final_acoustic[s])):
            finals[s] = (lat.final_graph[s], lat.final_acoustic[s], int(lat.final_dur[s]))
    return lat_from_arcs(lat.S, arcs, finals, key=lat.key)


def add_length_disambiguation(lat, state_len, dis_label=-1):
    """AddSequenceLengthDismabiguationSymbol, fstext/fstext-utils2.h:219-271: auxiliary chain
    aux[0..max_len] (label dis_label, weight One, last one final) and every final weight moved onto
    an epsilon arc to aux[length of the state].  Returns (lattice, state_len incl. aux states)."""
    S = lat.S
    if S == 0:
        return lat, np.asarray(state_len, np.int32)
    max_len = int(np.max(state_len))
    aux = [S + k for k in range(max_len + 1)]
    arcs, finals = [], {aux[max_len]: (0.0, 0.0, 0)}
    for s in range(S):
        for i in range(int(lat.arc_begin[s]), int(lat.arc_begin[s + 1])):
            arcs.append((s, int(lat.dst[i]), int(lat.ilabel[i]), int(lat.olabel[i]), lat.graph[i], lat.acoustic[i], int(lat.dur[i])))
        if not (np.isinf(lat.final_graph[s]) and np.isinf(lat.final_acoustic[s])):
            arcs.append((s, aux[int(state_len[s])], 0, 0, lat.final_graph[s], lat.final_acoustic[s], int(lat.final_dur[s])))
    for k in range(max_len):
        arcs.append((aux[k], aux[k + 1], dis_label, dis_label, 0.0, 0.0, 0))
    out = lat_from_arcs(S + max_len + 1, arcs, finals, key=lat.key)
    return out, np.concatenate([np.asarray(state_len, np.int32), np.arange(max_len + 1, dtype=np.int32)])


def best_path2(lat):
    """latbin/lattice-best-path2.cc:103-188 on a scaled, penalised, top-sorted lattice: returns
    (transcript, cost as float32, number of distinct minimum-cost paths capped at 2).  Pure-Python
    loops: for small lattices only.  Ties between equal-cost paths are kept in (state, arc) order;
    the third value lets a test skip lattices whose answer depends on that."""
    if lat.S == 0:
        return [], np.float32(0.0), 1
    dlat, state_len, _ = disambiguate_length(arc_sort_olabel(lat))
    plat, plen = add_length_disambiguation(dlat, state_len)
    alpha, beta, total, _, _ = alphas_betas(plat)
    src = plat.src()
    acc = {}
    for i in range(plat.A):  # state order, arc order: the reference's two nested iterators
        label = int(plat.olabel[i])
        if label == 0:
            continue
        key = (label, int(plen[plat.dst[i]]))
        w = alpha[src[i]] + beta[plat.dst[i]] - (float(plat.graph[i]) + float(plat.acoustic[i]))
        acc[key] = w if key not in acc else log_add(acc[key], w)
    post = {k: min(0.0, v - beta[0]) for k, v in acc.items()}
    S = plat.S
    inf = np.float32(np.inf)
    dist = np.full(S, inf, np.float32)
    ways = np.zeros(S, np.int64)
    parent = [(-1, 0)] * S
    dist[0], ways[0] = np.float32(0.0), 1
    best, best_state, best_ways = inf, -1, 0
    for s in range(S):
        if dist[s] == inf:
            continue
        if not (np.isinf(plat.final_graph[s]) and np.isinf(plat.final_acoustic[s])):
            if dist[s] < best:
                best, best_state, best_ways = dist[s], s, int(ways[s])
            elif dist[s] == best:
                best_ways += int(ways[s])
        for i in range(int(plat.arc_begin[s]), int(plat.arc_begin[s + 1])):
            label, d = int(plat.olabel[i]), int(plat.dst[i])
            w = np.float32(0.0)
            if label != 0:
                p = post[(label, int(plen[d]))]
                w = np.float32(0.0 if p >= 0.0 else np.exp(np.log1p(-np.exp(p))))
            nd = np.float32(dist[s] + w)
            if nd < dist[d]:
                dist[d], parent[d], ways[d] = nd, (s, label), ways[s]
            elif nd == dist[d]:
                ways[d] += ways[s]
    out = []
    s = best_state
    while s >= 0 and parent[s][0] >= 0:
        if parent[s][1] not in (0, -1):
            out.append(parent[s][1])
        s = parent[s][0]
    return out[::-1], best, min(best_ways, 2)



def best_path2_arc_costs(plat, plen):
    """The tropical arc costs of latbin/lattice-best-path2.cc:121-179 on the padded lattice `plat`
    (state lengths `plen`): 0 for epsilon arcs, exp(LogSub(0, log P(label, position))) = 1 - P as
    float32 for the others, with P accumulated by LogAdd in (state, arc) order and normalised by
    bw[Start()], clipped at 0 (:143-146)."""
    alpha, beta, total, _, _ = alphas_betas(plat)
    src = plat.src()
    acc = {}
    for i in range(plat.A):
        label = int(plat.olabel[i])
        if label == 0:
            continue
        key = (label, int(plen[plat.dst[i]]))
        w = alpha[src[i]] + beta[plat.dst[i]] - (float(plat.graph[i]) + float(plat.acoustic[i]))
        acc[key] = w if key not in acc else log_add(acc[key], w)
    cost = np.zeros(plat.A, np.float32)
    for i in range(plat.A):
        label = int(plat.olabel[i])
        if label != 0:
            p = min(0.0, acc[(label, int(plen[plat.dst[i]]))] - beta[0])
            cost[i] = np.float32(0.0 if p >= 0.0 else np.exp(np.log1p(-np.exp(p))))
    return cost


def enumerate_paths(lat, arc_cost, limit=200000):
    """Every complete path (start state 0 to a final state) of a small top-sorted DAG, the
    reference's own test idiom (fstext/expand-subpaths-test-utils.h:10-52: ShortestPath with
    n = INT_MAX, then compare multisets).  Returns [(float32 cost, [arc ids])]; the cost is the
    left-to-right float32 sum (TropicalWeight Times on StdArc), final states add nothing
    (latbin/lattice-best-path2.cc:157-160)."""
    out = []
    final = ~(np.isinf(lat.final_graph) & np.isinf(lat.final_acoustic))

    def walk(s, cost, arcs):
        if len(out) > limit:
            raise RuntimeError("too many paths to enumerate")
        if final[s]:
            out.append((cost, list(arcs)))
        for i in range(int(lat.arc_begin[s]), int(lat.arc_begin[s + 1])):
            arcs.append(i)
            walk(int(lat.dst[i]), np.float32(cost + arc_cost[i]), arcs)
            arcs.pop()

    if lat.S:
        walk(0, np.float32(0.0), [])
    return out


def kbest_paths(lat, arc_cost, k):
    """The k lowest-cost complete paths by dynamic programming over the top-sorted states (the
    algorithm the GPU kernel follows: per state the k best partial paths, pushed along the arcs;
    latbin/lattice-best-path2.cc:185 asks ShortestPath for n = 1, this is the n-best extension).
    Returns [(float32 cost, [arc ids])] ascending by cost; equal costs in an unspecified order.
    numpy per state: fine up to a few thousand states."""
    S = lat.S
    if S == 0:
        return []
    cands = [[] for _ in range(S)]  # per state: list of (cost array, arc id, k ranks of the source)
    lists = [None] * S
    final = ~(np.isinf(lat.final_graph) & np.isinf(lat.final_acoustic))
    fin = []
    for s in range(S):
        if s == 0:
            cost, back = np.zeros(1, np.float32), np.full((1, 2), -1, np.int64)
        elif not cands[s]:
            lists[s] = (np.zeros(0, np.float32), np.zeros((0, 2), np.int64))
            continue
        else:
            allc = np.concatenate([c for c, _, _ in cands[s]])
            allb = np.concatenate([np.stack([np.full(len(c), a, np.int64), np.arange(len(c), dtype=np.int64)], 1) for c, a, _ in cands[s]])
            order = np.argsort(allc, kind="stable")[:k]
            cost, back = allc[order], allb[order]
        lists[s] = (cost, back)
        cands[s] = None
        if final[s]:
            fin.append(s)
        for i in range(int(lat.arc_begin[s]), int(lat.arc_begin[s + 1])):
            cands[int(lat.dst[i])].append(((cost + np.float32(arc_cost[i])).astype(np.float32), i, None))
    if not fin:
        return []
    allc = np.concatenate([lists[f][0] for f in fin])
    alls = np.concatenate([np.stack([np.full(len(lists[f][0]), f, np.int64), np.arange(len(lists[f][0]), dtype=np.int64)], 1) for f in fin])
    order = np.argsort(allc, kind="stable")[:k]
    src = lat.src()
    out = []
    for c, (f, r) in zip(allc[order], alls[order]):
        arcs, s, r = [], int(f), int(r)
        while True:
            a, pr = lists[s][1][r]
            if a < 0:
                break
            arcs.append(int(a))
            s, r = int(src[a]), int(pr)
        out.append((np.float32(c), arcs[::-1]))
    return out

def lattice_beam(lat):
    """ComputeLatticeBeam (latbin/lattice-prune-dyn-beam.cc:27-90): (beam, fwd costs, bwd costs)."""
    fw, bw = np.zeros(max(lat.S, 1)), np.zeros(max(lat.S, 1))
    b = lib().kor_lattice_beam(C.c_int32(lat.S), _p(lat.arc_begin), _p(lat.dst), _p(lat.graph), _p(lat.acoustic),
                               _p(lat.final_graph), _p(lat.final_acoustic), _p(fw), _p(bw))
    return float(b), fw[:lat.S], bw[:lat.S]


def prune_lattice(lat, beam):
    """A.4 on an already scaled lattice.  Returns the pruned, connected lattice (state ids in
    order) or None if nothing survives; `extra` holds the kept old state / arc indices."""
    out = lat.copy()
    keep_arc = np.zeros(max(lat.A, 1), np.uint8)
    keep_state = np.zeros(max(lat.S, 1), np.uint8)
    n = lib().kor_prune_lattice(C.c_int32(lat.S), _p(out.arc_begin), _p(out.dst), _p(out.graph), _p(out.acoustic),
                                _p(out.final_graph), _p(out.final_acoustic), C.c_double(beam), _p(keep_arc), _p(keep_state))
    if n == 0:
        return None
    ks = keep_state[:lat.S].astype(bool)
    newid = np.cumsum(ks) - 1
    src = lat.src()
    ka = keep_arc[:lat.A].astype(bool) & ks[src] & ks[lat.dst]
    idx = np.nonzero(ka)[0]
    counts = np.bincount(newid[src[idx]], minlength=n)
    ab = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    res = Lat(ab, newid[lat.dst[idx]].astype(np.int32), lat.ilabel[idx], lat.olabel[idx], lat.graph[idx], lat.acoustic[idx],
              lat.dur[idx], out.final_graph[ks], out.final_acoustic[ks], lat.final_dur[ks], key=lat.key)
    res.extra["arc_old"] = idx
    res.extra["state_old"] = np.nonzero(ks)[0]
    return res


def prune_dyn_beam(lat, max_arcs, max_states, beam_ratio=0.9, min_beam=1e-3):
    """The loop of latbin/lattice-prune-dyn-beam.cc:166-186 on a scaled, penalised, top-sorted
    lattice, literally: ComputeLatticeBeam, then PruneLattice with the beam multiplied by
    beam_ratio (a BaseFloat) until the lattice is small enough.  Returns (indices of the surviving
    arcs and states in the input lattice, final beam); (None, beam) if nothing survives."""
    beam, _, _ = lattice_beam(lat)
    ratio, floor = float(np.float32(beam_ratio)), float(np.float32(min_beam))
    cur = lat
    arcs, states = np.arange(lat.A), np.arange(lat.S)
    while beam > floor and (cur.A > max_arcs or cur.S > max_states):
        beam = ratio * beam
        nxt = prune_lattice(cur, float(np.float32(beam)))  # PruneLattice(BaseFloat beam, ...)
        if nxt is None:
            return None, None, beam
        arcs, states = arcs[nxt.extra["arc_old"]], states[nxt.extra["state_old"]]
        cur = nxt
    return arcs, states, beam


def batch_fb_post(state_off, arc_off, arc_begin_global, dst, ilabel, graph, acoustic, final_graph,
                  final_acoustic, graph_scale, acoustic_scale, insertion_penalty):
    """Timed CPU baseline body (bench.py cpu_baseline) and batch parity checker."""
    U = len(state_off) - 1
    S, A = int(state_off[-1]), int(arc_off[-1])
    alpha = np.zeros(S, np.float64)
    beta = np.zeros(S, np.float64)
    total = np.zeros(U, np.float64)
    through = np.zeros(A, np.float64)
    lib().kor_batch_fb_post(C.c_int32(U), _p(state_off), _p(arc_off), _p(arc_begin_global), _p(dst),
                            _p(ilabel), _p(graph), _p(acoustic), _p(final_graph), _p(final_acoustic),
                            C.c_float(graph_scale), C.c_float(acoustic_scale),
                            C.c_float(insertion_penalty), _p(alpha), _p(beta), _p(total), _p(through))
    return alpha, beta, total, through
