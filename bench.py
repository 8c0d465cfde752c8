#!/usr/bin/env python3
"""bench.py -- lattice-arcs/second of the forward-backward + arc-posterior path on MI355X.

One "step" = one pass of the hot path (klat_forward_backward: forward+backward lockstep sweep,
per-lattice totals, arc log-posteriors, alpha/beta in natural log) over one batch of synthetic
lattices already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU (10k
lattices x 2k states x fanout 3, SURVEY.md Appendix D generator, --acoustic-scale=0.1
--graph-scale=1), i.e. weak scaling with utterances sharded across ranks and no data-path
collective.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(python -m torch.distributed.run, one rank per GPU) BEFORE touching the GPU, and exits with their
code.  Under torchrun (WORLD_SIZE set) --gpus must agree with it.

Beside the headline the line carries:
  shapes       sweep / step time and roofline fractions on the shard one GPU of an 8-GPU node would
               hold of BASELINE configs[3] (C4/8: 12 500 x 5 000 states) and configs[4] (C5/8: the
               skewed mix), and on a wide shape whose topological depth is far below its states;
  one_shot     stream packing + one forward-backward (+ segment index) on a FRESH C2 batch whose arrays
               are on the device: what a tool pays per batch (every caller sweeps a batch once);
  accumulate   device time of the segment-index and frame-posterior kernels on the C2 batch;
  index_gather (N > 1) klat_segment_index on every rank's shard + the gather of the 24-byte
               records to rank 0 over RCCL from device buffers (include/klat_dist.h), timed;
  strong_c4    STRONG scaling on north_star's target shape: BASELINE configs[3] whole (100 000 x 5 000
               states), the same list at every N, each rank sweeping its klat_partition_by_arcs range:
               forward-backward + segment index + gather to rank 0, node arcs/s.
KLAT_BENCH_DRYRUN=1: no GPU work at all -- the ranks rendezvous over gloo and run the launch,
sharding and gather plumbing on the host transport (the CPU test of the N > 1 path)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

# A skewed batch sweeps its layout classes on four HIP streams at once; the HIP runtime
# maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and torch holds one,
# so without this two classes share a queue and run one after the other (C5/8: 6.4 ms instead of 3.5).
# Read when the HIP runtime starts: set before torch or the library touch the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
TRAFFIC_PROFILE = os.path.join("profiles", "r05_hbm_traffic.json")  # PMC passes of this round's kernels
SCALES = (1.0, 0.1, 0.0)
FUSE = SCALES  # klat_pack_options.fuse_scales of every batch of the run (--no-fuse: None): 10-byte stream slots


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ns, fo, seed, scales, target_seconds=12.0):
    """The oracle (CPU restatement of the reference algorithm, kind "port") timed on this box's host
    cores on a bounded sample of the same workload, one lattice per task over `cores` threads
    (TaskSequencer's model, kwsbin2/lattice-word-index-segment.cc:260-275)."""
    import klat_amd
    from oracle import oracle as orc

    # a one-GPU box is given a 16-core share of the host (more threads only time-slice them)
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))

    def prep(n, first):
        a = klat_amd.synth(seed, ns[:n], fo[:n], first_utt=first, host_threads=8)
        S = int(a["state_off"][-1])
        gsrc = a["src"] + np.repeat(a["state_off"][:-1], np.diff(a["arc_off"]))
        a["arc_begin"] = np.concatenate([[0], np.cumsum(np.bincount(gsrc, minlength=S))]).astype(np.int64)
        return a

    def run(a, reps=1):
        for _ in range(reps):
            orc.batch_fb_post(a["state_off"], a["arc_off"], a["arc_begin"], a["dst"], a["ilabel"], a["graph"],
                              a["acoustic"], a["final_graph"], a["final_acoustic"], *scales)

    per_thread = min(32, len(ns))  # lattices per thread (distinct utterances per thread)
    probe = prep(per_thread, 0)
    t0 = time.perf_counter()
    run(probe)
    t1 = time.perf_counter() - t0  # one thread, one pass over its shard
    one_thread = int(probe["arc_off"][-1]) / t1
    reps = int(max(1, target_seconds / max(t1, 1e-9)))
    shards = [prep(per_thread, i * per_thread) for i in range(cores)]
    arcs = sum(int(s["arc_off"][-1]) for s in shards) * reps
    threads = [threading.Thread(target=run, args=(s, reps)) for s in shards]  # ctypes drops the GIL
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": arcs / dt, "unit": "lattice-arcs/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "one_thread_value": one_thread,
            "sample": "%d threads x %d lattices x %d states x fanout %d, %d passes each (%d arcs), fb + arc "
                      "posteriors, one lattice per task, %.1f s; each thread's %d lattices (%.1f MB) stay "
                      "cache-resident across its passes, which flatters the CPU"
                      % (cores, per_thread, int(ns[0]), int(fo[0]), reps, arcs, dt, per_thread,
                         int(probe["arc_off"][-1]) * 24 / 1e6)}


def source_hash():
    """sha1 over the kernel sources (csrc/), so that a profile taken from other sources shows: the traffic file records
    the hash (and the commit) it was taken at (tools/prof_summarize.py), the bench line names both."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "kaldi-lattice-utils_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):  # (kernels and the structures they share with the host)
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:12]


def measured_traffic(kernel_substr):
    """HBM bytes per sweep launch from this round's PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 runs of this same command, corrected as MI355X_MICROARCH.md prescribes;
    condensed by tools/prof_summarize.py into the one file named above).  None if absent."""
    f = os.path.join(ROOT, TRAFFIC_PROFILE)
    try:
        t = json.load(open(f))
    except Exception:
        return None
    meta = t.get("__profile__", {})
    now = source_hash()
    for k, v in t.items():
        if kernel_substr in k and isinstance(v, dict) and "hbm_bytes_per_launch" in v:
            return {"bytes": v["hbm_bytes_per_launch"], "kernel": k,
                    "source": "%s (taken at commit %s, kernel sources %s; this run's kernel sources %s%s)" % (
                        TRAFFIC_PROFILE, meta.get("commit", "?"), meta.get("source_hash", "?"), now,
                        "" if meta.get("source_hash") == now else ": the profile is of other sources")}
    return None


def c5_sizes(U):
    """BASELINE configs[4] skewed mix: sizes log-uniform in [200, 20 000] states,
    fanout uniform in {2,3,4,6}."""
    rng = np.random.RandomState(20260004)
    ns = np.exp(rng.uniform(np.log(200), np.log(20000), U)).astype(np.int32)
    fo = rng.choice([2, 3, 4, 6], U).astype(np.int32)
    return ns, fo


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """The parent of a `--gpus N` run: starts the ranks as child processes and returns their exit
    code.  Nothing here touches the GPU (no torch.cuda, no klat_amd call)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def median(xs):
    return float(np.median(np.asarray(xs, dtype=np.float64))) if len(xs) else 0.0


def time_shape(ctx, klat_amd, name, seed, ns, fo, host_threads, steps=30, warm=100, wide=0, lanes=0, frames=False):
    """sweep / step time of one more batch shape on this GPU (medians over `steps` steps).  A batch
    of few lattices that are far wider than deep runs on the level path (klat_level.hip); for such
    a batch the time through the streams (lanes fixed by the caller) is reported beside it.  (A hundred untimed
    steps first: the seconds of host work that build a shape leave the GPU's clocks down, and ten steps of a
    millisecond did not bring them back -- the same batch timed 1.39 ms cold and 1.10 ms warm.)"""
    arrays = klat_amd.synth(seed, ns, fo, host_threads=host_threads, wide=wide)
    b = klat_amd.Batch(ctx, arrays, host_threads=host_threads, lanes_per_lattice=lanes, fuse_scales=FUSE)
    info = b.info()
    depth = b.topological_depth()
    for _ in range(warm):
        b.forward_backward(*SCALES)
    for _ in range(steps):
        b.forward_backward(*SCALES)
    hist = ctx.timing_history(steps)
    total, tf, tb = b.totals()
    ok = bool(np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-6)
    A, S = info["num_arcs"], info["num_states"]
    sweep, step = median([t["sweep_ms"] for t in hist]), median([t["total_ms"] for t in hist])
    out = {"workload": name, "lattices": int(len(ns)), "arcs": A, "states": S,
           "topological_depth": {"max": int(depth.max()), "mean": float(depth.mean()),
                                 "note": "states on a lattice's longest path; the sweep takes max_rounds rounds whatever the depth"},
           "lanes_per_lattice": info["lanes_per_lattice"], "layout_classes": info["num_layouts"],
           "lattices_in_segments": info["chain_lattices"], "segments": info["chain_segments"],
           "sweep_waves": info["num_jobs"], "max_rounds": info["max_rounds"],
           "sweep_ms": sweep, "step_ms": step, "arcs_per_s": A / (step * 1e-3),
           "sweep_frac": (32.0 * A + 32.0 * S) / (sweep * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "step_frac": (56.0 * A + 48.0 * S) / (step * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "fwd_bwd_totals_agree": ok}
    if frames:  # frame posteriors (latbin/lattice-to-word-frame-post.cc:93-135) on this shape too: larger lattices than C2's
        import ctypes as C
        L = klat_amd.lib()
        nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
        foff = np.zeros(int(nf.sum()) + 1, np.int64)
        ne, fms, stored, fr = C.c_int64(0), C.c_float(0), C.c_int64(0), []
        for rep in range(4):  # (medians of three runs behind a first one, which sizes the record pool)
            klat_amd._check(L.klat_frame_post(b.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
            klat_amd._check(L.klat_get_accum_timings(b.h, None, C.byref(fms)))
            if rep > 0:
                fr.append(fms.value)
        klat_amd._check(L.klat_frame_stored_records(b.h, C.byref(stored)))
        fr_bytes = 28.0 * A + 4.0 * S + 8.0 * stored.value + 8.0 * int(nf.sum())  # (as in accumulate_leg)
        out["frame_ms"], out["frame_entries"], out["frame_records_stored"] = median(fr), ne.value, stored.value
        out["frame_frac"] = fr_bytes / (median(fr) * 1e-3) / 1e9 / HBM_PEAK_GBS
        # ... and the segment index (kwsbin2/lattice-word-index-segment.cc:143-174,98-126): 20 B/arc + 4 B/state in, 24 B/record out
        f, fkeep = klat_amd.Batch._filter(None, None)
        off, nrec, sms, sg = np.zeros(len(ns) + 1, np.int64), C.c_int64(0), C.c_float(0), []
        for rep in range(4):
            klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(nrec)))
            klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(sms), None))
            if rep > 0:
                sg.append(sms.value)
        out["segment_ms"], out["segment_records"] = median(sg), nrec.value
        out["segment_frac"] = (20.0 * A + 4.0 * S + 24.0 * nrec.value) / (median(sg) * 1e-3) / 1e9 / HBM_PEAK_GBS
    b.close()
    if info["level_lattices"] and lanes == 0:
        out["level_lattices"] = info["level_lattices"]
        through = time_shape(ctx, klat_amd, name, seed, ns, fo, host_threads, steps, warm, wide, lanes=4)
        out["streams_instead"] = {k: through[k] for k in ("sweep_ms", "step_ms", "lanes_per_lattice", "max_rounds")}
    return out


def accumulate_leg(klat_amd, batch, arrays, A):
    """Device time of the accumulate kernels (a9-a12) on the resident C2 batch: segment index
    (kwsbin2/lattice-word-index-segment.cc:143-174,98-126) and frame posteriors
    (latbin/lattice-to-word-frame-post.cc:93-135), medians of 5 runs each behind a first one."""
    import ctypes as C
    L = klat_amd.lib()
    U = batch.U
    f, keep = klat_amd.Batch._filter(None, None)
    off = np.zeros(U + 1, np.int64)
    n = C.c_int64(0)
    seg, fr = [], []
    ims, fms = C.c_float(0), C.c_float(0)
    for rep in range(6):  # (the first call builds the size classes and the scratch rows: not a steady-state time)
        klat_amd._check(L.klat_segment_index(batch.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        klat_amd._check(L.klat_get_accum_timings(batch.h, C.byref(ims), None))
        if rep > 0:
            seg.append(ims.value)
    records = n.value
    nf = arrays["state_time"][arrays["state_off"][1:] - 1].astype(np.int32)
    foff = np.zeros(int(nf.sum()) + 1, np.int64)
    ne = C.c_int64(0)
    for rep in range(6):  # (the first call sizes the record pool)
        klat_amd._check(L.klat_frame_post(batch.h, klat_amd._p(nf), klat_amd._p(foff), None, None, C.c_int64(0), C.byref(ne)))
        klat_amd._check(L.klat_get_accum_timings(batch.h, None, C.byref(fms)))
        if rep > 0:
            fr.append(fms.value)
    seg_ms, fr_ms = median(seg), median(fr)
    stored = C.c_int64(0)
    klat_amd._check(L.klat_frame_stored_records(batch.h, C.byref(stored)))
    # algorithmic bytes: src,dst,olabel (12) + logpost (8) per arc + state_time 4 B/state in, 24 B/record out;
    # frames: src,dst,ilabel,graph,acoustic (20) + logpost (8) per arc in; out 8 B per STORED record
    # (frames without an arc boundary share their predecessor's records) + 8 B per frame (first record, count)
    S = int(arrays["state_off"][-1])
    seg_bytes = 20.0 * A + 4.0 * S + 24.0 * records
    fr_bytes = 28.0 * A + 4.0 * S + 8.0 * stored.value + 8.0 * int(nf.sum())
    return {"workload": "the C2 batch of this run", "segment_ms": seg_ms, "segment_records": records,
            "segment_frac": seg_bytes / (seg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if seg_ms > 0 else None,
            "segment_algorithmic_bytes": seg_bytes,
            "frame_ms": fr_ms, "frame_entries": ne.value, "frame_records_stored": stored.value,
            "frame_frac": fr_bytes / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fr_ms > 0 else None,
            "frame_algorithmic_bytes": fr_bytes,
            "frac": min(x for x in (seg_bytes / (seg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if seg_ms > 0 else 1e9,
                                    fr_bytes / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fr_ms > 0 else 1e9))}


def one_shot_leg(klat_amd, torch, ctx, arrays, A, reps=5):
    """What every caller of the tools runs ONCE per batch: klat_batch_create's device passes (the stream
    packer: the reference's untimed preprocessing, kwsbin2/lattice-word-index-segment.cc:41-62,140-141)
    + one klat_forward_backward + the segment index, on a FRESH batch whose CSR arrays are already on
    the device.  Device milliseconds (HIP events) and the wall time of the three calls; medians."""
    import ctypes as C
    L = klat_amd.lib()
    dev, keep = dict(arrays), []
    for name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time"):
        t = torch.from_numpy(np.ascontiguousarray(arrays[name])).cuda()
        keep.append(t)
        dev[name] = int(t.data_ptr())
    torch.cuda.synchronize()
    f, fkeep = klat_amd.Batch._filter(None, None)
    rows = []
    for _ in range(reps):
        t0 = time.perf_counter()
        b = klat_amd.Batch(ctx, dev, fuse_scales=FUSE, borrow=True)  # (klat_pack_options.borrow_device_arrays: no copy of the arrays)
        t1 = time.perf_counter()
        b.forward_backward(*SCALES, arc_posteriors=False)  # what the tools call: sweep + totals, no per-arc pass
        ctx.sync()
        t2 = time.perf_counter()
        off = np.zeros(b.U + 1, np.int64)
        n, ims = C.c_int64(0), C.c_float(0)
        klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
        t3 = time.perf_counter()
        klat_amd._check(L.klat_get_accum_timings(b.h, C.byref(ims), None))
        pt, ft = b.pack_timings(), b.timings()
        rows.append((pt, ft["total_ms"], ims.value, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        b.close()
    rows = rows[1:]  # (the first call also allocates)
    med = lambda xs: median(list(xs))
    pack = {k: med(r[0][k] for r in rows) for k in ("count_ms", "level_plan_ms", "rounds_ms", "fill_ms", "posterior_prep_ms", "device_ms")}
    fb, seg = med(r[1] for r in rows), med(r[2] for r in rows)
    wall = [med(r[3] for r in rows), med(r[4] for r in rows), med(r[5] for r in rows)]
    del keep
    S = int(arrays["state_off"][-1])
    fb_bytes = 32.0 * A + 32.0 * S  # SURVEY.md 8d "fb-only": what ONE forward-backward of the batch must move
    return {"what": "fresh C2 batch, CSR arrays resident on the device and used in place (borrow_device_arrays): stream packing + one "
                    "forward-backward without the per-arc posterior pass (KLAT_FB_NO_ARC_POSTERIORS, the tools' call) + segment "
                    "index; device time by HIP events, wall time of the calls",
            "frac": fb_bytes / ((pack["device_ms"] + fb) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_what": "fb-only algorithmic bytes (32 B/arc + 32 B/state) / (pack + sweep device time) / 8 TB/s",
            "pack_device_ms": pack, "forward_backward_device_ms": fb, "segment_index_device_ms": seg,
            "device_ms": pack["device_ms"] + fb, "arcs_per_s": A / ((pack["device_ms"] + fb) * 1e-3),
            "device_ms_with_index": pack["device_ms"] + fb + seg, "arcs_per_s_with_index": A / ((pack["device_ms"] + fb + seg) * 1e-3),
            "wall_ms": {"batch_create": wall[0], "forward_backward": wall[1], "segment_index": wall[2], "total": sum(wall)},
            "arcs_per_s_wall": A / (sum(wall[:2]) * 1e-3), "arcs_per_s_wall_with_index": A / (sum(wall) * 1e-3)}


C4 = {"lattices": 100000, "states": 5000, "fanout": 3, "seed": 20260003}  # BASELINE configs[3], north_star's scaling target
STRONG_REFERENCE = os.path.join("profiles", "r05_strong_c4_1gpu.json")   # this leg at N = 1, taken on a box of the same pool


def strong_c4_leg(klat_amd, torch, dist, ctx, comm, rank, world, host_threads, barrier, lattices, reps=3, steps=5, warmup=3):
    """north_star's own scaling claim: the SAME list of lattices (BASELINE configs[3]: 100 000 x 5 000 states x
    fanout 3) at every N, strong scaling -- the reference's parallel model is TaskSequencer over one fixed archive
    (kwsbin2/lattice-word-index-segment.cc:260-275).  Every rank takes its klat_partition_by_arcs range (it
    generates only that range: first_utt = its first utterance), then, timed between barriers on all ranks:
    klat_forward_backward + klat_segment_index + klat_gather_index to rank 0 (device buffers over RCCL).
    Node rate = the list's arcs / the slowest rank's time."""
    import ctypes as C
    from klat_amd import dist as kdist
    L = klat_amd.lib()
    weights = np.full(lattices, C4["states"] * C4["fanout"], np.int64)
    bounds = kdist.partition_by_arcs(weights, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    # the rank's range, generated and uploaded in slices (host peak: one slice), concatenated on the device
    fields = ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time")
    parts = {f: [] for f in fields}
    state_off, arc_off = [np.zeros(1, np.int64)], [np.zeros(1, np.int64)]
    t_gen = time.perf_counter()
    for a0 in range(lo, hi, 12500):
        n = min(12500, hi - a0)
        sl = klat_amd.synth(C4["seed"], np.full(n, C4["states"], np.int32), np.full(n, C4["fanout"], np.int32), first_utt=a0,
                            host_threads=host_threads)
        for f in fields:
            parts[f].append(torch.from_numpy(np.ascontiguousarray(sl[f])).cuda())
        state_off.append(sl["state_off"][1:] + state_off[-1][-1])
        arc_off.append(sl["arc_off"][1:] + arc_off[-1][-1])
        del sl
    dev = {"state_off": np.concatenate(state_off), "arc_off": np.concatenate(arc_off)}
    keep = []
    for f in fields:
        t = torch.cat(parts[f]) if len(parts[f]) > 1 else parts[f][0]
        parts[f] = None
        keep.append(t)
        dev[f] = int(t.data_ptr())
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen
    t0 = time.perf_counter()
    b = klat_amd.Batch(ctx, dev, host_threads=host_threads, fuse_scales=FUSE)
    create_s = time.perf_counter() - t0
    A = b.A
    for _ in range(warmup):
        b.forward_backward(*SCALES)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.forward_backward(*SCALES)
    barrier()
    step_s = (time.perf_counter() - t0) / steps
    f, fkeep = klat_amd.Batch._filter(None, None)
    off = np.zeros(b.U + 1, np.int64)
    n = C.c_int64(0)
    times, compute, err = [], [], None
    tu = tr = 0
    for it in range(reps + 1):  # (the first pass allocates: not counted)
        barrier()
        t0 = time.perf_counter()
        try:
            b.forward_backward(*SCALES)
            klat_amd._check(L.klat_segment_index(b.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
            if it:
                compute.append(time.perf_counter() - t0)  # (the sizing call has read the records' counts back: the device is done)
        except Exception as e:  # (the whole list with its index on ONE GPU is the tightest fit)
            err = "%s: %s" % (type(e).__name__, str(e)[:200])
        if world > 1:  # a rank that failed must not leave the others inside the gather: they all learn of it first
            ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok[0]) == 0.0 and not err:
                err = "another rank failed before the gather"
        if err:
            break
        if comm is not None:
            tu, tr = comm.gather_index(b, klat_amd.SEGMENT_DTYPE, root=0, fetch=False)
        barrier()
        if it:
            times.append(time.perf_counter() - t0)
    tot = torch.tensor([float(A), float(n.value), step_s, median(times) if times else 0.0, 1.0 if err else 0.0, median(compute) if compute else 0.0],
                       dtype=torch.float64, device="cuda")
    if world > 1:
        sums = tot.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(tot, op=dist.ReduceOp.MAX)
        total_arcs, total_records, failed = float(sums[0]), float(sums[1]), float(tot[4]) > 0
    else:
        total_arcs, total_records, failed = float(A), float(n.value), err is not None
    step_max, pipe_max, compute_max = float(tot[2]), float(tot[3]), float(tot[5])
    out = None
    if rank == 0:
        ref = None
        try:
            ref = json.load(open(os.path.join(ROOT, STRONG_REFERENCE)))
        except Exception:
            pass
        out = {"workload": "BASELINE configs[3] WHOLE: %d lattices x %d states x fanout %d (seed %d), the same list at every N; "
                           "rank r sweeps its klat_partition_by_arcs range" % (lattices, C4["states"], C4["fanout"], C4["seed"]),
               "scaling": "strong", "ranks": world, "total_arcs": total_arcs, "rank0_lattices": hi - lo, "rank0_arcs": A,
               "steps": steps, "warmup": warmup,
               "rank0_generate_seconds": gen_s, "rank0_batch_create_seconds": create_s,
               "step_seconds": step_max, "node_arcs_per_s_step": total_arcs / step_max if step_max > 0 else None,
               "pipeline": "klat_forward_backward + klat_segment_index" + (" + klat_gather_index to rank 0 (RCCL, device buffers)" if comm is not None else ""),
               "pipeline_seconds": pipe_max if not failed else None,
               "node_arcs_per_s": total_arcs / pipe_max if (pipe_max > 0 and not failed) else None,
               # the same without the gather: every rank keeps its records (the tools' `ark:out.JOB` tables) -- the 34 GB of
               # this list's records into ONE GPU's links are a transfer the one-GPU line does not have
               "pipeline_compute_seconds": compute_max if not failed else None,
               "node_arcs_per_s_compute": total_arcs / compute_max if (compute_max > 0 and not failed) else None,
               "index_records": total_records, "error": err}
        if comm is not None and not failed:
            st = comm.stats()
            out["gather"] = {"utterances": int(tu), "records": int(tr), "payload_bytes_into_rank0": st["payload_bytes"], "seconds": st["seconds"],
                             "GB_per_s_into_rank0": st["payload_bytes"] / st["seconds"] / 1e9 if st["seconds"] > 0 else None}
        if world == 1:
            out["speedup_vs_1gpu"] = 1.0
        elif ref and ref.get("total_arcs") == total_arcs and ref.get("node_arcs_per_s") and out["node_arcs_per_s"]:
            out["one_gpu_reference"] = {"node_arcs_per_s": ref["node_arcs_per_s"], "node_arcs_per_s_step": ref.get("node_arcs_per_s_step"),
                                        "source": STRONG_REFERENCE + " (this leg at N = 1 on a box of the same pool)"}
            out["speedup_vs_1gpu"] = out["node_arcs_per_s"] / ref["node_arcs_per_s"]
            if ref.get("node_arcs_per_s_step"):
                out["speedup_vs_1gpu_step"] = out["node_arcs_per_s_step"] / ref["node_arcs_per_s_step"]
            if out.get("node_arcs_per_s_compute"):  # (at N = 1 the pipeline has no gather: its rate is the compute rate)
                out["speedup_vs_1gpu_compute"] = out["node_arcs_per_s_compute"] / ref["node_arcs_per_s"]
        else:
            out["speedup_vs_1gpu"] = None  # (no 1-GPU record of this list size: compare the per-N lines of the scaling run)
    b.close()
    del keep
    torch.cuda.empty_cache()
    ctx.trim()
    return out


def promote_strong(out, world):
    """At N > 1 the line answers north_star's question directly: the top-level fields the driver parses (value,
    ms_per_step, scaling, config.workload) are the STRONG-scaling step on BASELINE configs[3] whole -- the same list at
    every N, forward-backward + arc posteriors, max over ranks between barriers; the weak-scaled configs[1] line (what
    N = 1 reports) moves to `weak_c2`, the gather-inclusive pipeline stays under `strong_c4`."""
    st = out.get("strong_c4")
    if world <= 1 or not st or st.get("error") or not st.get("node_arcs_per_s_step"):
        out["promoted"] = None
        return
    keep = ("value", "ms_per_step", "scaling", "config", "steps", "warmup")
    out["weak_c2"] = {k: out.get(k) for k in keep}
    out["value"] = st["node_arcs_per_s_step"]
    out["ms_per_step"] = 1e3 * st["step_seconds"]
    out["scaling"] = "strong"
    out["steps"], out["warmup"] = st.get("steps", out.get("steps")), st.get("warmup", out.get("warmup"))
    out["config"] = {"workload": st["workload"], "total_arcs": st["total_arcs"], "ranks": st["ranks"],
                     "sharding": "klat_partition_by_arcs ranges of ONE list, no data-path collective",
                     "fused_scales": FUSE is not None}
    out["speedup_vs_1gpu_step"] = st.get("speedup_vs_1gpu_step")
    out["transport_ranks"] = (out.get("index_gather") or {}).get("transport_ranks")
    out["promoted"] = "strong_c4.node_arcs_per_s_step (north_star: >= 6x at 8 GPUs on configs[3]); roofline / kernel_ms / one_shot / weak_c2 describe the configs[1] batch"


def dry_run(args, rank, world):
    """CPU rehearsal of the N-rank plumbing (launch, rendezvous, sharding, C++ gather over the host
    transport).  No kernel runs and nothing is measured: `value` is null."""
    import torch.distributed as dist

    import klat_amd
    from klat_amd import dist as kdist

    if world > 1:
        dist.init_process_group("gloo")
    U = 64
    ns = np.full(U, 40, np.int32)
    fo = np.full(U, 3, np.int32)
    arrays = klat_amd.synth(args.seed, ns, fo)  # every rank sees the same utterance list
    bounds = kdist.partition_by_arcs(np.diff(arrays["arc_off"]), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    # stand-in records: one per arc of the rank's utterances, tagged with the global arc number
    a0, a1 = int(arrays["arc_off"][lo]), int(arrays["arc_off"][hi])
    rec = np.zeros(a1 - a0, klat_amd.SEGMENT_DTYPE)
    rec["label"] = np.arange(a0, a1)
    off = (arrays["arc_off"][lo:hi + 1] - a0).astype(np.int64)
    gathered = None
    if world > 1:
        comm = kdist.Comm.over_group()
        all_rec, all_off = comm.gather_records(rec, off, root=0)
        nranks = comm.info()["transport_ranks"]
        comm.close()
        if rank == 0:
            assert all_off.tolist() == arrays["arc_off"].tolist() and all_rec["label"].tolist() == list(range(len(all_rec)))
            gathered = {"ranks": nranks, "records": int(len(all_rec)), "order_restored": True}
        dist.barrier()
    # the strong-scaling leg's plumbing: ONE list (a small stand-in for configs[3]), every rank generates only its
    # klat_partition_by_arcs range with first_utt = its first utterance; the ranges must tile the list exactly
    strong = None
    L = min(args.strong_lattices, 96)
    sb = kdist.partition_by_arcs(np.full(L, 40 * 3, np.int64), world)
    slo, shi = int(sb[rank]), int(sb[rank + 1])
    mine = klat_amd.synth(C4["seed"], np.full(shi - slo, 40, np.int32), np.full(shi - slo, 3, np.int32), first_utt=slo)
    whole = klat_amd.synth(C4["seed"], np.full(L, 40, np.int32), np.full(L, 3, np.int32)) if rank == 0 else None
    counts = np.array([shi - slo, int(mine["arc_off"][-1])], np.int64)
    if world > 1:
        import torch
        import torch.distributed as dist2
        dist2.init_process_group("gloo") if not dist2.is_initialized() else None
        t = torch.from_numpy(counts.copy())
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist2.all_gather(allc, t)
        tot = np.sum([c.numpy() for c in allc], axis=0)
        dist2.barrier()
        dist2.destroy_process_group()
    else:
        tot = counts
    if rank == 0:
        a0, a1 = int(whole["arc_off"][slo]), int(whole["arc_off"][shi])
        assert np.array_equal(mine["dst"], whole["dst"][a0:a1]) and np.array_equal(mine["graph"], whole["graph"][a0:a1]), "a rank's range must be the list's own lattices"
        assert int(tot[0]) == L and int(tot[1]) == int(whole["arc_off"][-1]), (tot, L)
        strong = {"scaling": "strong", "ranks": world, "lattices": L, "total_arcs": int(tot[1]), "ranges_tile_the_list": True}
    if rank == 0:
        line = {"metric": "lattice-arcs/sec (whole node) on forward-backward + arc posteriors", "value": None,
                "unit": "lattice-arcs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                "config": {"workload": "dry run"}, "dry_run": True, "index_gather": gathered, "strong_c4": strong}
        if strong is not None and world > 1:  # the same promotion as a real run's (promote_strong), with stand-in figures
            strong.update({"workload": "stand-in for BASELINE configs[3] WHOLE: %d lattices, the same list at every N" % L,
                           "node_arcs_per_s_step": 1.0, "step_seconds": 1.0, "steps": args.steps, "warmup": args.warmup})
            if gathered:
                line["index_gather"] = dict(gathered, transport_ranks=gathered["ranks"])
            promote_strong(line, world)
            line["value"] = None  # (nothing was measured)
        print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--lattices", type=int, default=10000, help="lattices per GPU")
    ap.add_argument("--states", type=int, default=2000)
    ap.add_argument("--fanout", type=int, default=3)
    ap.add_argument("--seed", type=int, default=20260001)
    ap.add_argument("--lanes-per-lattice", type=int, default=0)
    ap.add_argument("--ring-slots", type=int, default=0)
    ap.add_argument("--settle-seconds", type=float, default=1.0, help="untimed steps before the counted warm-up (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true")
    ap.add_argument("--no-accumulate", action="store_true")
    ap.add_argument("--no-one-shot", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the host-in / host-out legs (profiles: their pieces' launches carry the step's kernel names)")
    ap.add_argument("--no-fuse", action="store_true", help="generic 12-byte stream slots (the scales stay kernel arguments)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg on BASELINE configs[3]")
    ap.add_argument("--strong-lattices", type=int, default=C4["lattices"], help="lattices of the strong-scaling list (default: configs[3] whole)")
    args = ap.parse_args()

    global FUSE
    if args.no_fuse:
        FUSE = None
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # parent: no GPU call before or after this point
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or let --gpus spawn them)" % (args.gpus, world))
    if os.environ.get("KLAT_BENCH_DRYRUN"):
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    import klat_amd

    if not torch.cuda.is_available() or klat_amd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > 1 and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d GPUs visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    U = args.lattices
    ns = np.full(U, args.states, np.int32)
    fo = np.full(U, args.fanout, np.int32)
    host_threads = max(1, (os.cpu_count() or 1) // max(1, world))  # ranks share the host cores
    arrays = klat_amd.synth(args.seed, ns, fo, first_utt=rank * U, host_threads=host_threads)
    ctx = klat_amd.Ctx(local_rank)
    t0 = time.perf_counter()
    batch = klat_amd.Batch(ctx, arrays, lanes_per_lattice=args.lanes_per_lattice, ring_slots=args.ring_slots,
                           host_threads=host_threads, fuse_scales=FUSE)
    pack_s = time.perf_counter() - t0
    info = batch.info()
    A, S = info["num_arcs"], info["num_states"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU's clock ramps for the first ~50 ms of load: a fixed duration of untimed steps first,
    # then the counted warm-up
    t_end = time.perf_counter() + args.settle_seconds
    while time.perf_counter() < t_end:
        for _ in range(16):
            batch.forward_backward(*SCALES)
        ctx.sync()
    for _ in range(args.warmup):
        batch.forward_backward(*SCALES)
    barrier()
    # The K steps are enqueued without waiting in between (the library records HIP events around
    # its kernels on the ctx stream into a ring); their per-kernel durations are read after the
    # timed region.
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.forward_backward(*SCALES)
    barrier()
    elapsed = time.perf_counter() - t0
    hist = ctx.timing_history(min(args.steps, 512))
    sweep_ms = [t["sweep_ms"] for t in hist]
    post_ms = [t["posterior_ms"] for t in hist]
    tot_ms = [t["total_ms"] for t in hist]
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # D2H of the step's results, outside the timed region and reported on its own
    t0 = time.perf_counter()
    post = batch.arc_logpost()
    alpha, beta = batch.alpha_beta()
    d2h_s = time.perf_counter() - t0
    d2h_bytes = post.nbytes + alpha.nbytes + beta.nbytes
    del post, alpha, beta

    # light end-of-run sanity: forward and backward totals agree
    total, tf, tb = batch.totals()
    assert os.environ.get("KLAT_EXPERIMENT") or np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-6

    def device_copy_gbs():
        """Second roofline denominator: what a plain device-to-device copy of 1 GiB
        reaches on this box, read + write bytes per second."""
        x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        y = torch.empty_like(x)
        x.fill_(1.0)
        for _ in range(3):
            y.copy_(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        return 10 * 2.0 * x.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * A * args.steps / elapsed
        sweep_med = median(sweep_ms)
        # algorithmic bytes of the sweep launch (SURVEY.md 8d "fb-only"): 32 B/arc + 32 B/state
        # (with fused scales the stream slot is 10 bytes, not 12: the ALGORITHMIC figure of SURVEY.md 8d stays the
        # denominator-independent 32 B per arc, so a faster launch shows as a higher fraction)
        sweep_bytes = 32.0 * A + 32.0 * S
        achieved = sweep_bytes / (sweep_med * 1e-3) / 1e9
        step_bytes = 56.0 * A + 48.0 * S  # fb + arc posterior: 72 B/arc at fanout 3
        traffic = measured_traffic("sweep_kernel<%d, true" % info["lanes_per_lattice"])
        copy_gbs = device_copy_gbs()
        out = {
            "metric": "lattice-arcs/sec (whole node) on forward-backward + arc posteriors",
            "value": value, "unit": "lattice-arcs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d lattices/GPU x %d states x fanout %d, "
                                   "acoustic-scale 0.1 (SURVEY.md App. D seed %d)" % (U, args.states, args.fanout, args.seed),
                       "lattices_per_gpu": U, "states": args.states, "fanout": args.fanout,
                       "arcs_per_gpu": A, "lanes_per_lattice": info["lanes_per_lattice"],
                       "ring_slots": info["ring_slots"], "sweep_waves": info["num_jobs"],
                       "topological_depth": args.states, "sharding": "utterances across ranks, no data-path collective",
                       "fused_scales": FUSE is not None, "stream_slot_bytes": 10 if FUSE is not None else 12},
            "roofline": {"bound": "hbm", "kernel": "klat::sweep_kernel (forward+backward lockstep sweep)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "device_copy_gbs": copy_gbs, "frac_of_device_copy": achieved / copy_gbs,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "algorithmic_bytes_per_launch": sweep_bytes, "avg_launch_ms": float(np.mean(sweep_ms)),
                         "median_launch_ms": sweep_med},
            "roofline_step": {"bound": "hbm", "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_step": step_bytes},
            "kernel_ms": {"sweep": sweep_med, "posterior": median(post_ms), "device_total": median(tot_ms),
                          "statistic": "median over the timed steps (HIP events on the ctx stream)"},
            "batch_create_seconds": pack_s,
            "results_d2h": {"seconds": d2h_s, "bytes": d2h_bytes, "what": "arc log-posteriors + alpha + beta to pageable host memory"},
        }

    # ---- the boundary when the caller holds HOST lattices and wants HOST results: create (H2D of the CSR
    # arrays + stream packing) -> sweep -> fetch, serial on this context; never `value`.  Twice: the full
    # descriptor into pageable result arrays (what round 2 measured), and the compact descriptor
    # (klat_batch_desc_compact: 13 B per arc up instead of 28) into page-locked result buffers.
    if rank == 0 and not args.no_boundary:
        def boundary(make_batch, post, ab):
            times = []
            for _ in range(4):
                t0 = time.perf_counter()
                b2 = make_batch()
                b2.forward_backward(*SCALES)
                b2.arc_logpost(out=post)
                b2.alpha_beta(out=ab)
                times.append(time.perf_counter() - t0)
                b2.close()
            return median(times[1:])

        post = np.empty(A, np.float64)
        ab = (np.empty(S, np.float64), np.empty(S, np.float64))
        sec = boundary(lambda: klat_amd.Batch(ctx, arrays, host_threads=host_threads, fuse_scales=FUSE), post, ab)
        out["boundary"] = {"what": "host CSR arrays in, host alpha/beta/arc posteriors out, one batch after the other "
                                   "(H2D + packing + sweep + D2H; PCIe-bound)", "seconds_per_batch": sec,
                           "arcs_per_s": A / sec, "h2d_bytes": int(sum(v.nbytes for v in arrays.values())),
                           "d2h_bytes": int(post.nbytes + ab[0].nbytes + ab[1].nbytes)}
        del post, ab
        compact = klat_amd.compact_arrays(arrays)
        if compact is not None:
            ppost = klat_amd.pinned_empty(A, np.float64)
            pab = (klat_amd.pinned_empty(S, np.float64), klat_amd.pinned_empty(S, np.float64))
            pin = {k: (klat_amd.pinned_copy(v) if k not in ("state_off", "arc_off") else v) for k, v in compact.items()}
            sec_c = boundary(lambda: klat_amd.Batch.from_compact(ctx, pin, host_threads=host_threads), ppost, pab)
            out["boundary"]["compact"] = {"what": "klat_batch_desc_compact (out-degrees, one label array, one-byte arc spans) from and into "
                                                  "page-locked buffers (klat_host_alloc)", "seconds_per_batch": sec_c, "arcs_per_s": A / sec_c,
                                          "h2d_bytes": int(sum(v.nbytes for v in compact.values())),
                                          "d2h_bytes": int(ppost.nbytes + pab[0].nbytes + pab[1].nbytes)}
            # the same traffic through ONE call that pipelines the batch in pieces over three streams (upload of piece
            # k+1 | packing and sweep of piece k | results of piece k-1): klat_forward_backward_host
            try:
                pout = {"alpha": pab[0], "beta": pab[1], "total": klat_amd.pinned_empty(U, np.float64), "post": ppost}
                times = []
                # (a context of its own, closed afterwards: the call keeps two helper contexts -- two more HIP streams -- with
                # the context it runs on, and the skewed shapes below are sensitive to how a process's streams map onto the
                # hardware queues)
                pctx = klat_amd.Ctx(0)
                try:
                    for _ in range(4):
                        t0 = time.perf_counter()
                        klat_amd.forward_backward_host(pctx, pin, SCALES, pieces=0, fuse=FUSE, out=pout, host_threads=host_threads)
                        times.append(time.perf_counter() - t0)
                finally:
                    pctx.close()
                sec_p = median(times[1:])
                out["boundary"]["pipelined"] = {"what": "klat_forward_backward_host: the compact descriptor in, alpha / beta / totals / arc posteriors out, "
                                                        "page-locked buffers, the batch in pieces over three streams", "seconds_per_batch": sec_p,
                                                "arcs_per_s": A / sec_p}
            except Exception as e:
                out["boundary"]["pipelined"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            del ppost, pab, pin

    # ---- the one-shot path of the tools: pack + sweep (+ index) on a fresh batch, device-resident input
    if rank == 0 and not args.no_one_shot:
        try:
            out["one_shot"] = one_shot_leg(klat_amd, torch, ctx, arrays, A)
        except Exception as e:  # must not take the headline with it
            out["one_shot"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # ---- accumulate stage on the resident C2 batch (all ranks: the gather leg needs every shard's index)
    if not args.no_accumulate:
        acc = accumulate_leg(klat_amd, batch, arrays, A) if (rank == 0 or world > 1) else None
        if rank == 0:
            out["accumulate"] = acc

    # ---- N > 1: segment index of every shard gathered to rank 0 over RCCL, from device buffers
    comm = None
    if world > 1:
        try:  # (a failure of this leg must not take the headline line with it)
            from klat_amd import dist as kdist
            uid = [kdist.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = kdist.Comm.rccl(ctx, rank, world, uid[0])
            nranks = comm.info()["transport_ranks"]
            assert nranks == world, "RCCL communicator has %d ranks, expected %d" % (nranks, world)
            import ctypes as C
            f, keep = klat_amd.Batch._filter(None, None)
            off = np.zeros(batch.U + 1, np.int64)
            n = C.c_int64(0)
            local_err = None
            try:
                klat_amd._check(klat_amd.lib().klat_segment_index(batch.h, C.byref(f), klat_amd._p(off), None, C.c_int64(0), C.byref(n)))
            except Exception as e:
                local_err = e
            ok = torch.tensor([0.0 if local_err else 1.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # (every rank enters the gather, or none does)
            if float(ok[0]) == 0.0:
                raise local_err or RuntimeError("another rank failed before the gather")
            comm.gather_index(batch, klat_amd.SEGMENT_DTYPE, root=0, fetch=False)  # warm-up: buffers, RCCL channels
            times = []
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                tu, tr = comm.gather_index(batch, klat_amd.SEGMENT_DTYPE, root=0, fetch=False)
                barrier()
                times.append(time.perf_counter() - t0)
            if rank == 0:
                st = comm.stats()
                sec = median(times)
                assert tu == world * batch.U, (tu, world, batch.U)
                rec, roff = comm._fetch(klat_amd.SEGMENT_DTYPE, 0, tu, tr)  # order check on the host, outside the timing
                assert roff[-1] == tr and np.array_equal(roff[:batch.U + 1], off), "rank 0's own records must come first, unchanged"
                out["index_gather"] = {"ranks": nranks, "transport_ranks": nranks, "transport": "RCCL send/recv from device buffers, all-gather of sizes",
                                       "utterances": int(tu), "records": int(tr), "payload_bytes_into_rank0": st["payload_bytes"],
                                       "seconds": sec, "GB_per_s_into_rank0": st["payload_bytes"] / sec / 1e9 if sec > 0 else None,
                                       "order": "rank order = input order (checked)"}
                del rec, roff
        except Exception as e:
            if rank == 0:
                out["index_gather"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # ---- strong scaling on BASELINE configs[3] whole: north_star's own target shape, the same list at every N
    batch.close()
    ctx.trim()
    if not args.no_strong:
        try:
            # (at N > 1 this leg's step is the line's headline: the K steps and W warm-ups the driver asked for)
            strong = strong_c4_leg(klat_amd, torch, dist, ctx, comm, rank, world, host_threads, barrier, args.strong_lattices,
                                   steps=args.steps if world > 1 else 5, warmup=args.warmup if world > 1 else 3)
            if rank == 0:
                out["strong_c4"] = strong
        except Exception as e:
            if rank == 0:
                out["strong_c4"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if comm is not None:
        comm.close()

    # ---- other BASELINE shapes, on rank 0 (the others wait at the barrier below)
    if rank == 0 and not args.no_shapes:
        batch.close()
        shapes = []
        c5n, c5f = c5_sizes(2500)
        for name, seed, sn, sf, wide in (
                ("C4/8 shard (configs[3]): 12500 lattices x 5000 states x fanout 3", 20260003,
                 np.full(12500, 5000, np.int32), np.full(12500, 3, np.int32), 0),
                ("C5/8 shard (configs[4] mix): 2500 lattices, 200..20000 states log-uniform, fanout {2,3,4,6}", 20260004, c5n, c5f, 0),
                ("wide: 10000 lattices x 2000 states x fanout 3 in levels of 32 states (depth ~63)", 20260005,
                 np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32), 32),
                ("few and wide: 64 lattices x 2000 states x fanout 3 in levels of 64 states (depth ~32), the level path", 20260006,
                 np.full(64, 2000, np.int32), np.full(64, 3, np.int32), 64),
                # what ONE call of the reference's interface holds (B2: one lattice per TaskSequencer worker,
                # kwsbin2/lattice-word-index-segment.cc:264-272): a lone lattice is all chain -- swept in segments
                ("one lattice of 5000 states x fanout 3 (a configs[3] lattice on its own)", 20260007,
                 np.full(1, 5000, np.int32), np.full(1, 3, np.int32), 0),
                ("one lattice of 20000 states x fanout 3 (the longest of the configs[4] mix on its own)", 20260008,
                 np.full(1, 20000, np.int32), np.full(1, 3, np.int32), 0)):
            try:
                shapes.append(time_shape(ctx, klat_amd, name, seed, sn, sf, host_threads, wide=wide, frames=name.startswith(("C4/8", "C5/8"))))
            except Exception as e:  # a shape that fails must not take the headline with it
                shapes.append({"workload": name, "error": str(e)[:300]})
        out["shapes"] = shapes
    if world > 1:
        dist.barrier()

    if rank == 0:
        if not args.no_cpu_baseline:  # (at every N: the same host cores, the same sample)
            out["cpu_baseline"] = cpu_baseline(ns, fo, args.seed, SCALES)
        promote_strong(out, world)
        print(json.dumps(out))
    batch.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
