"""ctypes plumbing over lib/libklat_dist.so (include/klat_dist.h) for tests and bench.py.

The sharding (`klat_partition_by_arcs`) and the gather of index records to the writing rank
(`klat_gather_records` / `klat_gather_index`) are C++ in the library; nothing is computed here.
Two transports: RCCL from device buffers (the product path, `Comm.rccl`), and a host-memory
transport whose callbacks are served by a torch.distributed process group (`Comm.over_group`;
gloo in the CPU tests) -- the same C++ gather logic runs over both."""
import ctypes as C
import os

import numpy as np

from . import KLAT_OK, KlatError, ROOT, lib as _klat_lib

DIST_LIB_PATH = os.path.join(ROOT, "lib", "libklat_dist.so")
UNIQUE_ID_BYTES = 128
_dlib = None

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
_GRP = C.CFUNCTYPE(C.c_int, C.c_void_p)
_P2P = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather_i64", _AG), ("group_begin", _GRP), ("send", _P2P),
                ("recv", _P2P), ("group_end", _GRP)]


def dlib():
    """Loads the multi-GPU library (links RCCL); raises if it has not been built."""
    global _dlib
    if _dlib is None:
        _klat_lib()  # libklat.so first: the error string lives there
        if not os.path.exists(DIST_LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % DIST_LIB_PATH)
        _dlib = C.CDLL(DIST_LIB_PATH)
        _dlib.klat_comm_destroy.restype = None
        _dlib.klat_comm_destroy.argtypes = [C.c_void_p]
    return _dlib


def _check(rc):
    if rc != KLAT_OK:
        raise KlatError(rc, _klat_lib().klat_last_error().decode())


def partition_by_arcs(arc_counts, world_size):
    """Contiguous utterance ranges with (nearly) equal total arcs (klat_partition_by_arcs).  Returns
    bounds[world_size+1]; rank r owns utterances [bounds[r], bounds[r+1]).  Input order is preserved
    inside and across ranks, so concatenating the ranks' outputs restores the reference's emission
    order (serial destructors, kwsbin2/lattice-word-index-segment.cc:96-128)."""
    a = np.ascontiguousarray(arc_counts, dtype=np.int64)
    b = np.zeros(world_size + 1, np.int64)
    _check(dlib().klat_partition_by_arcs(a.ctypes.data_as(C.c_void_p), C.c_int64(len(a)), C.c_int32(world_size),
                                         b.ctypes.data_as(C.c_void_p)))
    return b


def rccl_unique_id():
    buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
    _check(dlib().klat_rccl_unique_id(buf))
    return bytes(buf)


class Comm:
    """klat_comm handle."""

    def __init__(self):
        self.h = C.c_void_p()
        self._keep = None

    @classmethod
    def rccl(cls, ctx, rank, world, unique_id):
        self = cls()
        buf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _check(dlib().klat_comm_create_rccl(ctx.h, C.c_int32(rank), C.c_int32(world), buf, C.byref(self.h)))
        return self

    @classmethod
    def over_group(cls, group=None):
        """Host-memory transport served by a torch.distributed process group (any backend that
        moves CPU tensors, i.e. gloo)."""
        import torch
        import torch.distributed as dist

        self = cls()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        pending = []

        def view(ptr, nbytes):
            return torch.from_numpy(np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(ptr)))

        def all_gather(_u, send, recv, count):
            try:
                s = view(send, 8 * count).view(torch.int64).clone()
                outs = [torch.zeros(count, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(outs, s, group=group)
                view(recv, 8 * count * world).view(torch.int64).copy_(torch.cat(outs))
                return 0
            except Exception:  # never let an exception cross the C boundary
                return 1

        def begin(_u):
            pending.clear()
            return 0

        def send(_u, buf, nbytes, peer):
            pending.append(dist.P2POp(dist.isend, view(buf, nbytes), peer, group))
            return 0

        def recv(_u, buf, nbytes, peer):
            pending.append(dist.P2POp(dist.irecv, view(buf, nbytes), peer, group))
            return 0

        def end(_u):
            try:
                if pending:
                    for w in dist.batch_isend_irecv(list(pending)):
                        w.wait()
                pending.clear()
                return 0
            except Exception:
                return 1

        t = Transport(None, _AG(all_gather), _GRP(begin), _P2P(send), _P2P(recv), _GRP(end))
        self._keep = t
        _check(dlib().klat_comm_create_custom(C.c_int32(rank), C.c_int32(world), C.byref(t), C.byref(self.h)))
        return self

    def info(self):
        r, w, n = C.c_int32(), C.c_int32(), C.c_int32()
        _check(dlib().klat_comm_info(self.h, C.byref(r), C.byref(w), C.byref(n)))
        return {"rank": r.value, "world": w.value, "transport_ranks": n.value}

    def gather_records(self, records, record_off, root=0):
        """records: this rank's records (host numpy structured array with a custom transport; an int
        device pointer + dtype itemsize 24 is what gather_index handles); record_off:
        int64[num_local_utts+1].  On root returns (all_records, all_record_off) over the utterances of
        all ranks in rank order; elsewhere (None, None)."""
        rec = np.ascontiguousarray(records)
        off = np.ascontiguousarray(record_off, dtype=np.int64)
        tu, tr = C.c_int64(), C.c_int64()
        _check(dlib().klat_gather_records(self.h, rec.ctypes.data_as(C.c_void_p), C.c_int32(rec.dtype.itemsize),
                                          off.ctypes.data_as(C.c_void_p), C.c_int64(len(off) - 1), C.c_int32(root),
                                          C.byref(tu), C.byref(tr)))
        return self._fetch(rec.dtype, root, tu.value, tr.value)

    def gather_index(self, batch, dtype, root=0, fetch=True):
        """The index of `batch` (after its sizing call) gathered to root from the ranks' device
        buffers over RCCL."""
        tu, tr = C.c_int64(), C.c_int64()
        _check(dlib().klat_gather_index(self.h, batch.h, C.c_int32(root), C.byref(tu), C.byref(tr)))
        if not fetch:
            return tu.value, tr.value
        return self._fetch(np.dtype(dtype), root, tu.value, tr.value)

    def _fetch(self, dtype, root, total_utts, total_records):
        if self.info()["rank"] != root:
            return None, None
        out = np.zeros(total_records, dtype)
        off = np.zeros(total_utts + 1, np.int64)
        _check(dlib().klat_gather_fetch(self.h, off.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                        C.c_int64(len(out))))
        return out, off

    def stats(self):
        s, b = C.c_double(), C.c_int64()
        _check(dlib().klat_gather_stats(self.h, C.byref(s), C.byref(b)))
        return {"seconds": s.value, "payload_bytes": b.value}

    def close(self):
        if self.h:
            dlib().klat_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_records(records, record_off, dst=0, group=None, device=None):
    """One-shot form over a torch.distributed group with CPU tensors (the tests): the C++ gather of
    klat_dist.h over the host transport."""
    comm = Comm.over_group(group)
    try:
        return comm.gather_records(records, record_off, root=dst)
    finally:
        comm.close()
