"""ctypes plumbing over lib/libklat.so (the C ABI of include/klat.h) for tests and bench.py.

The product is the C/HIP library and the C++ tools; this module only moves numpy arrays
across the ABI.  It never computes anything itself and has no CPU fallback: if the HIP
library is missing, or no GPU is present when a compute entry point is called, it raises.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(_PKG, "..", ".."))
LIB_PATH = os.environ.get("KLAT_LIB_OVERRIDE") or os.path.join(ROOT, "lib", "libklat.so")  # override: kernel experiments only

KLAT_OK = 0
ERR_INVALID_ARG, ERR_BAD_TOPOLOGY, ERR_HIP, ERR_NO_DEVICE, ERR_OOM, ERR_STATE, ERR_TIMES = range(1, 8)


class KlatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("klat error %d: %s" % (code, msg))
        self.code = code


class BatchDesc(C.Structure):
    _fields_ = [("num_lattices", C.c_int32), ("state_off", C.c_void_p), ("arc_off", C.c_void_p),
                ("src", C.c_void_p), ("dst", C.c_void_p), ("ilabel", C.c_void_p),
                ("olabel", C.c_void_p), ("graph", C.c_void_p), ("acoustic", C.c_void_p),
                ("final_graph", C.c_void_p), ("final_acoustic", C.c_void_p),
                ("state_time", C.c_void_p)]


class BatchDescCompact(C.Structure):
    _fields_ = [("num_lattices", C.c_int32), ("state_off", C.c_void_p), ("arc_off", C.c_void_p),
                ("out_degree", C.c_void_p), ("span", C.c_void_p), ("label", C.c_void_p),
                ("graph", C.c_void_p), ("acoustic", C.c_void_p), ("final_graph", C.c_void_p),
                ("final_acoustic", C.c_void_p), ("state_time", C.c_void_p)]


COMPACT_FIELDS = [("state_off", np.int64), ("arc_off", np.int64), ("out_degree", np.int32), ("span", np.uint8),
                  ("label", np.int32), ("graph", np.float32), ("acoustic", np.float32), ("final_graph", np.float32),
                  ("final_acoustic", np.float32), ("state_time", np.int32)]


def compact_arrays(arrays):
    """klat_batch_desc arrays -> klat_batch_desc_compact arrays (an acceptor whose arcs span at most 255
    states), or None if the batch does not fit that form."""
    span = arrays["dst"].astype(np.int64) - arrays["src"].astype(np.int64)
    if len(span) and (span.min() < 1 or span.max() > 255 or not np.array_equal(arrays["ilabel"], arrays["olabel"])):
        return None
    S = int(arrays["state_off"][-1])
    gsrc = arrays["src"].astype(np.int64) + np.repeat(arrays["state_off"][:-1], np.diff(arrays["arc_off"]))
    out = {"state_off": arrays["state_off"], "arc_off": arrays["arc_off"], "out_degree": np.bincount(gsrc, minlength=S).astype(np.int32),
           "span": span.astype(np.uint8), "label": arrays["ilabel"], "graph": arrays["graph"], "acoustic": arrays["acoustic"],
           "final_graph": arrays["final_graph"], "final_acoustic": arrays["final_acoustic"]}
    if arrays.get("state_time") is not None:
        out["state_time"] = arrays["state_time"]
    return out


class Scales(C.Structure):
    _fields_ = [("graph_scale", C.c_float), ("acoustic_scale", C.c_float),
                ("insertion_penalty", C.c_float)]


class PackOptions(C.Structure):
    _fields_ = [("lanes_per_lattice", C.c_int32), ("ring_slots", C.c_int32),
                ("host_threads", C.c_int32), ("fuse_scales", C.c_void_p), ("borrow_device_arrays", C.c_int32)]


class BatchInfo(C.Structure):
    _fields_ = [("num_states", C.c_int64), ("num_arcs", C.c_int64), ("fwd_slots", C.c_int64),
                ("bwd_slots", C.c_int64), ("far_terms", C.c_int64),
                ("lanes_per_lattice", C.c_int32), ("ring_slots", C.c_int32),
                ("num_jobs", C.c_int32), ("max_rounds", C.c_int32), ("num_layouts", C.c_int32),
                ("level_lattices", C.c_int32), ("chain_lattices", C.c_int32), ("chain_segments", C.c_int32),
                ("lean_jobs", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Timings(C.Structure):
    _fields_ = [("sweep_ms", C.c_float), ("totals_ms", C.c_float), ("posterior_ms", C.c_float),
                ("total_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PackTimings(C.Structure):
    _fields_ = [("count_ms", C.c_float), ("level_plan_ms", C.c_float), ("rounds_ms", C.c_float),
                ("fill_ms", C.c_float), ("posterior_prep_ms", C.c_float), ("device_ms", C.c_float),
                ("host_ms", C.c_float), ("chain_prep_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LabelFilter(C.Structure):
    _fields_ = [("include", C.c_void_p), ("num_include", C.c_int32), ("exclude", C.c_void_p),
                ("num_exclude", C.c_int32)]


SEGMENT_DTYPE = np.dtype([("label", "<i4"), ("t0", "<i4"), ("t1", "<i4"), ("pad", "<i4"), ("logp", "<f8")])
POSITION_DTYPE = np.dtype([("label", "<i4"), ("pos", "<i4"), ("t0", "<i4"), ("t1", "<i4"), ("logp", "<f8")])
JOB_DTYPE = np.dtype([("chunk_off", "<i8"), ("nchunks", "<i4"), ("dir", "<i4"), ("lane_tab", "<i4"), ("tag", "<i4")])
LANE_DTYPE = np.dtype([("state_base", "<i8"), ("num_states", "<i4"), ("lattice", "<i4")])
ITEM_DTYPE = np.dtype([("arc_begin", "<i8"), ("count", "<i4"), ("lattice", "<i4")])

_lib = None


def lib():
    """Loads the HIP extension; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.klat_last_error.restype = C.c_char_p
        _lib.klat_version.restype = C.c_char_p
        _lib.klat_ctx_stream.restype = C.c_void_p
        _lib.klat_ctx_stream.argtypes = [C.c_void_p]
        for name in ("klat_ctx_destroy", "klat_batch_destroy", "klat_packed_destroy", "klat_host_free"):
            getattr(_lib, name).restype = None
            getattr(_lib, name).argtypes = [C.c_void_p]
        atexit.register(_close_all)  # after the HIP runtime registered its own handlers: runs before them
    return _lib


def _check(rc):
    if rc != KLAT_OK:
        raise KlatError(rc, lib().klat_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pinned_empty(shape, dtype):
    """numpy array over page-locked host memory (klat_host_alloc); freed with the array."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    p = C.c_void_p()
    _check(lib().klat_host_alloc(C.byref(p), C.c_size_t(max(n * dt.itemsize, 1))))
    buf = (C.c_ubyte * max(n * dt.itemsize, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, lib().klat_host_free, C.c_void_p(p.value))
    return arr


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def forward_backward_host(ctx, compact, scales=(1.0, 1.0, 0.0), pieces=0, want_post=True, fuse=False, out=None, host_threads=0):
    """klat_forward_backward_host: compact host arrays in, host alpha / beta / total (/ arc posteriors) out, the batch
    pipelined in pieces.  `out`: dict of preallocated arrays (alpha, beta, total, post), e.g. page-locked ones."""
    d, keep = BatchDescCompact(), {}
    d.num_lattices = len(compact["state_off"]) - 1
    for name, dt in COMPACT_FIELDS:
        a = compact.get(name)
        if a is None:
            setattr(d, name, None)
            continue
        a = np.ascontiguousarray(a, dtype=dt)
        keep[name] = a
        setattr(d, name, a.ctypes.data)
    U = d.num_lattices
    S = int(keep["state_off"][-1]) if U else 0
    A = int(keep["arc_off"][-1]) if U else 0
    out = dict(out or {})
    out.setdefault("alpha", np.empty(S, np.float64))
    out.setdefault("beta", np.empty(S, np.float64))
    out.setdefault("total", np.empty(U, np.float64))
    if want_post:
        out.setdefault("post", np.empty(A, np.float64))
    sc = Scales(*scales)
    opt = PackOptions(0, 0, host_threads, C.cast(C.pointer(sc), C.c_void_p) if fuse else None)
    _check(lib().klat_forward_backward_host(ctx.h, C.byref(d), C.byref(opt), C.byref(sc), C.c_int32(pieces), _p(out["alpha"]), _p(out["beta"]),
                                            _p(out["total"]), _p(out["post"]) if want_post else None))
    return out


def device_count():
    n = C.c_int(0)
    rc = lib().klat_device_count(C.byref(n))
    return n.value if rc == KLAT_OK else 0


FIELDS = [("state_off", np.int64), ("arc_off", np.int64), ("src", np.int32), ("dst", np.int32),
          ("ilabel", np.int32), ("olabel", np.int32), ("graph", np.float32),
          ("acoustic", np.float32), ("final_graph", np.float32), ("final_acoustic", np.float32),
          ("state_time", np.int32)]


def make_desc(arrays):
    """arrays: dict with the klat_batch_desc fields as numpy arrays (state_time optional).
    Returns (BatchDesc, keepalive list)."""
    keep = {}
    d = BatchDesc()
    d.num_lattices = len(arrays["state_off"]) - 1
    for name, dt in FIELDS:
        a = arrays.get(name)
        if a is None:
            setattr(d, name, None)
            continue
        if isinstance(a, int):  # a device pointer (the caller keeps the allocation alive)
            setattr(d, name, a)
            continue
        a = np.ascontiguousarray(a, dtype=dt)
        keep[name] = a
        setattr(d, name, a.ctypes.data)
    return d, keep


def concat_lattices(lats, times=None):
    """Flattens a list of single lattices (objects with arc_begin,dst,ilabel,olabel,graph,
    acoustic,final_graph,final_acoustic) into the klat_batch_desc arrays."""
    so, ao = [0], [0]
    for l in lats:
        so.append(so[-1] + (len(l.arc_begin) - 1))
        ao.append(ao[-1] + len(l.dst))
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
    srcs = [np.repeat(np.arange(len(l.arc_begin) - 1, dtype=np.int32), np.diff(l.arc_begin)) for l in lats]
    out = {
        "state_off": np.array(so, np.int64), "arc_off": np.array(ao, np.int64),
        "src": cat(srcs, np.int32), "dst": cat([l.dst for l in lats], np.int32),
        "ilabel": cat([l.ilabel for l in lats], np.int32), "olabel": cat([l.olabel for l in lats], np.int32),
        "graph": cat([l.graph for l in lats], np.float32), "acoustic": cat([l.acoustic for l in lats], np.float32),
        "final_graph": cat([l.final_graph for l in lats], np.float32),
        "final_acoustic": cat([l.final_acoustic for l in lats], np.float32),
    }
    if times is not None:
        out["state_time"] = cat(list(times), np.int32)
    return out


# Handles still open when the interpreter exits are closed here, batches before contexts and both
# before the HIP runtime's own exit handlers run (module teardown order is arbitrary, and
# destroying a stream after the runtime is gone aborts the process).
_live = weakref.WeakSet()


def _close_all():
    for kind in (Batch, Ctx):
        for obj in [o for o in list(_live) if isinstance(o, kind)]:
            obj.close()


class Ctx:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().klat_ctx_create(C.c_int(device), C.byref(self.h)))
        _live.add(self)

    def timing_history(self, count):
        """Kernel times (dicts) of the last `count` forward_backward calls on this context."""
        arr = (Timings * count)()
        _check(lib().klat_ctx_timing_history(self.h, C.c_int32(count), arr))
        return [t.as_dict() for t in arr]

    def trim(self):
        """hands the context's recycled device blocks back to the runtime; returns the bytes released"""
        n = C.c_int64(0)
        _check(lib().klat_ctx_trim(self.h, C.byref(n)))
        return n.value

    def sync(self):
        _check(lib().klat_ctx_sync(self.h))

    @property
    def stream(self):
        return lib().klat_ctx_stream(self.h)

    def close(self):
        if self.h:
            lib().klat_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    def __init__(self, ctx, arrays, lanes_per_lattice=0, ring_slots=0, host_threads=0, fuse_scales=None, borrow=False):
        """fuse_scales: (graph_scale, acoustic_scale, insertion_penalty) every sweep of the batch will use
        (klat_pack_options.fuse_scales: 10-byte stream slots), or None.  borrow: device arrays among `arrays` are used
        in place (klat_pack_options.borrow_device_arrays): the caller keeps them alive while the batch lives."""
        self.ctx = ctx
        self.h = C.c_void_p()
        _live.add(self)
        self.desc, self._keep = make_desc(arrays)
        self.U = self.desc.num_lattices
        self.S = int(self._keep["state_off"][-1]) if self.U else 0
        self.A = int(self._keep["arc_off"][-1]) if self.U else 0
        fs = Scales(*fuse_scales) if fuse_scales is not None else None
        opt = PackOptions(lanes_per_lattice, ring_slots, host_threads, C.addressof(fs) if fs is not None else None, 1 if borrow else 0)
        self.h = C.c_void_p()
        _check(lib().klat_batch_create(ctx.h, C.byref(self.desc), C.byref(opt), C.byref(self.h)))

    @classmethod
    def from_compact(cls, ctx, compact, lanes_per_lattice=0, ring_slots=0, host_threads=0):
        """klat_batch_create_compact: `compact` as compact_arrays() returns it."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.h = C.c_void_p()
        _live.add(self)
        d, keep = BatchDescCompact(), {}
        d.num_lattices = len(compact["state_off"]) - 1
        for name, dt in COMPACT_FIELDS:
            a = compact.get(name)
            if a is None:
                setattr(d, name, None)
                continue
            a = np.ascontiguousarray(a, dtype=dt)
            keep[name] = a
            setattr(d, name, a.ctypes.data)
        self.desc, self._keep = d, keep
        self.U = d.num_lattices
        self.S = int(keep["state_off"][-1]) if self.U else 0
        self.A = int(keep["arc_off"][-1]) if self.U else 0
        opt = PackOptions(lanes_per_lattice, ring_slots, host_threads, None, 0)
        _check(lib().klat_batch_create_compact(ctx.h, C.byref(d), C.byref(opt), C.byref(self.h)))
        return self

    def info(self):
        i = BatchInfo()
        _check(lib().klat_batch_get_info(self.h, C.byref(i)))
        return i.as_dict()

    def pack_timings(self):
        t = PackTimings()
        _check(lib().klat_batch_get_pack_timings(self.h, C.byref(t)))
        return t.as_dict()

    def chain_fallback(self):
        """True if the last forward_backward swept the segment path's lattices again through their streams"""
        n = C.c_int32(0)
        _check(lib().klat_batch_chain_fallback(self.h, C.byref(n)))
        return bool(n.value)

    def topological_depth(self):
        d = np.zeros(max(self.U, 1), np.int32)
        _check(lib().klat_batch_topological_depth(self.h, _p(d)))
        return d[:self.U]

    def forward_backward(self, graph_scale=1.0, acoustic_scale=1.0, insertion_penalty=0.0, arc_posteriors=True):
        s = Scales(graph_scale, acoustic_scale, insertion_penalty)
        if arc_posteriors:
            _check(lib().klat_forward_backward(self.h, C.byref(s)))
        else:  # KLAT_FB_NO_ARC_POSTERIORS: what the tools run (the accumulators read alpha / beta)
            _check(lib().klat_forward_backward_ex(self.h, C.byref(s), C.c_uint32(1)))

    def timings(self):
        t = Timings()
        _check(lib().klat_get_timings(self.h, C.byref(t)))
        return t.as_dict()

    def exact_sweep_jobs(self):
        n = C.c_int32(0)
        _check(lib().klat_get_exact_sweep_jobs(self.h, C.byref(n)))
        return n.value

    def alpha_beta(self, out=None):
        a, b = out if out is not None else (np.empty(self.S, np.float64), np.empty(self.S, np.float64))
        _check(lib().klat_get_alpha_beta(self.h, _p(a), _p(b)))
        return a, b

    def totals(self):
        t, f, b = (np.empty(self.U, np.float64) for _ in range(3))
        _check(lib().klat_get_totals(self.h, _p(t), _p(f), _p(b)))
        return t, f, b

    def arc_logpost(self, out=None):
        x = out if out is not None else np.empty(self.A, np.float64)
        _check(lib().klat_get_arc_logpost(self.h, _p(x)))
        return x

    @staticmethod
    def _filter(include, exclude):
        inc = np.array(sorted(set(include or [])), np.int32)
        exc = np.array(sorted(set(exclude or [])), np.int32)
        f = LabelFilter(inc.ctypes.data if len(inc) else None, len(inc),
                        exc.ctypes.data if len(exc) else None, len(exc))
        return f, (inc, exc)

    def segment_index(self, include=None, exclude=None):
        f, keep = self._filter(include, exclude)
        off = np.zeros(self.U + 1, np.int64)
        n = C.c_int64(0)
        _check(lib().klat_segment_index(self.h, C.byref(f), _p(off), None, C.c_int64(0), C.byref(n)))
        rec = np.zeros(max(n.value, 1), SEGMENT_DTYPE)
        _check(lib().klat_segment_index(self.h, C.byref(f), _p(off), _p(rec), C.c_int64(len(rec)), C.byref(n)))
        return off, rec[:n.value]

    def position_index(self, state_len, include=None, exclude=None):
        f, keep = self._filter(include, exclude)
        sl = np.ascontiguousarray(state_len, np.int32)
        off = np.zeros(self.U + 1, np.int64)
        n = C.c_int64(0)
        _check(lib().klat_position_index(self.h, _p(sl), C.byref(f), _p(off), None, C.c_int64(0), C.byref(n)))
        rec = np.zeros(max(n.value, 1), POSITION_DTYPE)
        _check(lib().klat_position_index(self.h, _p(sl), C.byref(f), _p(off), _p(rec), C.c_int64(len(rec)), C.byref(n)))
        return off, rec[:n.value]

    def frame_post(self, num_frames):
        nf = np.ascontiguousarray(num_frames, np.int32)
        T = int(nf.sum())
        off = np.zeros(T + 1, np.int64)
        n = C.c_int64(0)
        _check(lib().klat_frame_post(self.h, _p(nf), _p(off), None, None, C.c_int64(0), C.byref(n)))
        labels = np.zeros(max(n.value, 1), np.int32)
        logps = np.zeros(max(n.value, 1), np.float32)
        _check(lib().klat_frame_post(self.h, _p(nf), _p(off), _p(labels), _p(logps), C.c_int64(len(labels)), C.byref(n)))
        return off, labels[:n.value], logps[:n.value]

    def frame_stored_records(self):
        """Records the last frame_post kept on the device: frames without an arc boundary share
        their predecessor's (klat_frame_stored_records)."""
        n = C.c_int64(0)
        _check(lib().klat_frame_stored_records(self.h, C.byref(n)))
        return n.value

    def streams(self):
        """(sw, sg, sa) as built on the device by klat_batch_create."""
        i = self.info()
        n = i["fwd_slots"] + i["bwd_slots"]
        sw, sg, sa = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        _check(lib().klat_batch_get_streams(self.h, _p(sw), _p(sg), _p(sa)))
        return sw, sg, sa

    def fused_streams(self):
        """(sw, sg, sl) of a batch built with fuse_scales: term words, (float)cost, 16-bit residuals."""
        i = self.info()
        n = i["fwd_slots"] + i["bwd_slots"]
        sw, sg, sl = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int16)
        _check(lib().klat_batch_get_fused_streams(self.h, _p(sw), _p(sg), _p(sl)))
        return sw, sg, sl

    def viterbi(self, graph_scale=1.0, acoustic_scale=1.0, insertion_penalty=0.0):
        s = Scales(graph_scale, acoustic_scale, insertion_penalty)
        _check(lib().klat_viterbi(self.h, C.byref(s)))

    def viterbi_costs(self):
        fw, bw, best = np.zeros(max(self.S, 1)), np.zeros(max(self.S, 1)), np.zeros(max(self.U, 1))
        _check(lib().klat_get_viterbi(self.h, _p(fw), _p(bw), _p(best)))
        return fw[:self.S], bw[:self.S], best[:self.U]

    def prune_mask(self, beam):
        ka, kf, lb = np.zeros(max(self.A, 1), np.uint8), np.zeros(max(self.S, 1), np.uint8), np.zeros(max(self.U, 1))
        _check(lib().klat_prune_mask(self.h, C.c_float(beam), _p(ka), _p(kf), _p(lb)))
        return ka[:self.A], kf[:self.S], lb[:self.U]

    def nbest(self, arc_cost, k, max_labels):
        """klat_nbest: returns (num_paths[U], path_cost[U,k], path_len[U,k], path_labels[U,k,max_labels])."""
        ac = np.ascontiguousarray(arc_cost, np.float32)
        npaths = np.zeros(max(self.U, 1), np.int32)
        cost = np.full((max(self.U, 1), k), np.inf, np.float32)
        plen = np.zeros((max(self.U, 1), k), np.int32)
        labels = np.zeros((max(self.U, 1), k, max_labels), np.int32)
        _check(lib().klat_nbest(self.h, _p(ac), C.c_int32(k), C.c_int32(max_labels), _p(npaths), _p(cost), _p(plen), _p(labels)))
        return npaths[:self.U], cost[:self.U], plen[:self.U], labels[:self.U]

    def utterance_scores(self, word_off, words):
        """log P(word occurs on the path) for the query words of every lattice (klat_utterance_scores)."""
        wo = np.ascontiguousarray(word_off, np.int64)
        ws = np.ascontiguousarray(words, np.int32)
        out = np.zeros(max(len(ws), 1), np.float64)
        _check(lib().klat_utterance_scores(self.h, _p(wo), _p(ws), _p(out)))
        return out[:len(ws)]

    def close(self):
        if self.h:
            lib().klat_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_host(arrays, lanes_per_lattice=0, ring_slots=0, host_threads=0):
    """Host-only: returns the packed lockstep streams (no GPU needed)."""
    desc, keep = make_desc(arrays)
    opt = PackOptions(lanes_per_lattice, ring_slots, host_threads)
    h = C.c_void_p()
    _check(lib().klat_pack_host(C.byref(desc), C.byref(opt), C.byref(h)))
    try:
        info = BatchInfo()
        ns, nl, ni = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().klat_packed_info(h, C.byref(info), C.byref(ns), C.byref(nl), C.byref(ni)))
        sw = np.zeros(ns.value, np.int32)
        sg = np.zeros(ns.value, np.float32)
        sa = np.zeros(ns.value, np.float32)
        jobs = np.zeros(info.num_jobs, JOB_DTYPE)
        lanes = np.zeros(nl.value, LANE_DTYPE)
        items = np.zeros(ni.value, ITEM_DTYPE)
        _check(lib().klat_packed_copy(h, _p(sw), _p(sg), _p(sa), _p(jobs), _p(lanes), _p(items)))
    finally:
        lib().klat_packed_destroy(h)
    return {"info": info.as_dict(), "sw": sw, "sg": sg, "sa": sa, "jobs": jobs, "lanes": lanes, "items": items}


def synth(seed, num_states, fanout, first_utt=0, host_threads=0, wide=0):
    """SURVEY.md Appendix D generator (C++ implementation in the library); wide > 0 selects the
    levelled shape of klat_synth_wide_generate with that level width."""
    ns = np.ascontiguousarray(num_states, np.int32)
    fo = np.ascontiguousarray(fanout, np.int32)
    n = len(ns)
    St, At = C.c_int64(0), C.c_int64(0)
    if wide:
        _check(lib().klat_synth_wide_sizes(C.c_int32(n), _p(ns), _p(fo), C.c_int32(wide), C.byref(St), C.byref(At)))
    else:
        _check(lib().klat_synth_sizes(C.c_int32(n), _p(ns), _p(fo), C.byref(St), C.byref(At)))
    out = {"state_off": np.zeros(n + 1, np.int64), "arc_off": np.zeros(n + 1, np.int64)}
    for name, dt in FIELDS[2:]:
        size = At.value if name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic") else St.value
        out[name] = np.zeros(size, dt)
    tail = (_p(out["state_off"]), _p(out["arc_off"]), _p(out["src"]), _p(out["dst"]),
            _p(out["ilabel"]), _p(out["olabel"]), _p(out["graph"]), _p(out["acoustic"]),
            _p(out["final_graph"]), _p(out["final_acoustic"]), _p(out["state_time"]), C.c_int32(host_threads))
    if wide:
        _check(lib().klat_synth_wide_generate(C.c_uint64(seed), C.c_int64(first_utt), C.c_int32(n), _p(ns), _p(fo),
                                              C.c_int32(wide), *tail))
    else:
        _check(lib().klat_synth_generate(C.c_uint64(seed), C.c_int64(first_utt), C.c_int32(n), _p(ns), _p(fo), *tail))
    return out
