"""Multi-GPU plumbing of the path: utterances are sharded across ranks by
cumulative arc count, each rank runs its shard on its own GPU with no data-path collective,
and the only exchange is the gather of the index records to rank 0 (all-gather of the
per-rank counts + point-to-point payload, RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests).  torch.distributed is used as the transport only."""
import numpy as np


def partition_by_arcs(arc_counts, world_size):
    """Contiguous utterance ranges with (nearly) equal total arcs -- not equal utterance
    counts -- so that skewed lattice sizes balance (BASELINE config 5).  Returns
    bounds[world_size+1]; rank r owns utterances [bounds[r], bounds[r+1]).  Input order is
    preserved inside and across ranks, so concatenating the ranks' outputs restores the
    reference's emission order (serial destructors, kwsbin2/lattice-word-index-segment.cc:96-128)."""
    arc_counts = np.asarray(arc_counts, dtype=np.int64)
    n = len(arc_counts)
    # weight 1 per utterance on top of its arcs so that empty lattices still spread out
    csum = np.concatenate([[0], np.cumsum(arc_counts + 1)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


def gather_records(records, record_off, dst=0, group=None, device=None):
    """Gathers each rank's (records, record_off) to rank `dst`.

    records: 1-D numpy structured/byte array of this rank's records (utterance-major);
    record_off: int64[num_local_utts+1].  On `dst` returns (all_records, all_record_off)
    covering the utterances of all ranks in rank order; elsewhere returns (None, None).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    itemsize = records.dtype.itemsize
    meta = torch.tensor([len(records) * itemsize, len(record_off) - 1], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.cpu().numpy() for m in metas]
    payload = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).copy()).to(dev)
    offs = torch.from_numpy(np.diff(record_off).astype(np.int64)).to(dev)  # per-utterance counts
    if rank == dst:
        bufs, cnts, ops = [], [], []
        for r in range(world):
            if r == dst:
                bufs.append(payload), cnts.append(offs)
                continue
            b = torch.empty(int(metas[r][0]), dtype=torch.uint8, device=dev)
            c = torch.empty(int(metas[r][1]), dtype=torch.int64, device=dev)
            bufs.append(b), cnts.append(c)
            if b.numel():
                ops.append(dist.P2POp(dist.irecv, b, r, group))
            if c.numel():
                ops.append(dist.P2POp(dist.irecv, c, r, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        all_rec = np.concatenate([b.cpu().numpy() for b in bufs]).view(records.dtype) if bufs else records[:0]
        counts = np.concatenate([c.cpu().numpy() for c in cnts]) if cnts else np.zeros(0, np.int64)
        return all_rec, np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ops = []
    if payload.numel():
        ops.append(dist.P2POp(dist.isend, payload, dst, group))
    if offs.numel():
        ops.append(dist.P2POp(dist.isend, offs, dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None, None
