#!/usr/bin/env python3
"""bench.py -- lattice-arcs/second of the forward-backward + arc-posterior path on MI355X.

One "step" = one pass of the hot path (klat_forward_backward: forward+backward lockstep
sweep, per-lattice totals, arc log-posteriors, alpha/beta in natural log) over one batch of
synthetic lattices already resident in HBM.  Workload at every N: BASELINE.json configs[1]
per GPU (10k lattices x 2k states x fanout 3, SURVEY.md Appendix D generator, scales
--acoustic-scale=0.1 --graph-scale=1), i.e. weak scaling with utterances sharded across
ranks and no data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(ns, fo, seed, scales, target_seconds=12.0):
    """The oracle (CPU restatement of the reference algorithm, kind "port") timed on this
    box's host cores on a bounded sample of the same workload, one lattice per task over
    `cores` threads (TaskSequencer's model, kwsbin2/lattice-word-index-segment.cc:260-275)."""
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
    return {"value": arcs / dt, "unit": "lattice-arcs/s", "cores": cores, "kind": "port",
            "one_thread_value": one_thread,
            "sample": "%d threads x %d lattices x %d states x fanout %d, %d passes each (%d arcs), "
                      "fb + arc posteriors, one lattice per task, %.1f s"
                      % (cores, per_thread, int(ns[0]), int(fo[0]), reps, arcs, dt)}


def measured_traffic(G):
    """HBM bytes per sweep launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 runs of this same command, corrected as MI355X_MICROARCH.md prescribes;
    condensed by tools/prof_summarize.py into profiles/*_hbm_traffic.json).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), key=os.path.getmtime)
    for f in reversed(files):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        for k, v in t.items():
            if "sweep_kernel<%d, true" % G in k or "sweep_kernel<%d>" % G in k:  # the fast sweep
                return {"bytes": v["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT)}
    return None


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    import klat_amd

    if not torch.cuda.is_available() or klat_amd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    U = args.lattices
    ns = np.full(U, args.states, np.int32)
    fo = np.full(U, args.fanout, np.int32)
    scales = (1.0, 0.1, 0.0)
    host_threads = max(1, (os.cpu_count() or 1) // max(1, world))  # ranks share the host cores
    arrays = klat_amd.synth(args.seed, ns, fo, first_utt=rank * U, host_threads=host_threads)
    ctx = klat_amd.Ctx(local_rank)
    t0 = time.perf_counter()
    batch = klat_amd.Batch(ctx, arrays, lanes_per_lattice=args.lanes_per_lattice, ring_slots=args.ring_slots,
                           host_threads=host_threads)
    pack_s = time.perf_counter() - t0
    info = batch.info()
    A, S = info["num_arcs"], info["num_states"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.forward_backward(*scales)
    barrier()
    # The K steps are enqueued without waiting in between (the library records HIP events around
    # its kernels on the ctx stream into a ring); their per-kernel durations are read after the
    # timed region.
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.forward_backward(*scales)
    barrier()
    elapsed = time.perf_counter() - t0
    hist = ctx.timing_history(min(args.steps, 512))
    sweep_ms = [t["sweep_ms"] for t in hist]
    post_ms = [t["posterior_ms"] for t in hist]
    conv_ms = [t["convert_ms"] for t in hist]
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

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * A * args.steps / elapsed
        sweep_avg = float(np.mean(sweep_ms))
        # algorithmic bytes of the sweep launch (SURVEY.md 8d "fb-only"): 32 B/arc + 32 B/state
        sweep_bytes = 32.0 * A + 32.0 * S
        achieved = sweep_bytes / (sweep_avg * 1e-3) / 1e9
        step_bytes = 56.0 * A + 48.0 * S  # fb + arc posterior: 72 B/arc at fanout 3
        traffic = measured_traffic(info["lanes_per_lattice"])
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
                       "topological_depth": args.states, "sharding": "utterances across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "klat::sweep_kernel (forward+backward lockstep sweep)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "device_copy_gbs": copy_gbs, "frac_of_device_copy": achieved / copy_gbs,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "algorithmic_bytes_per_launch": sweep_bytes, "avg_launch_ms": sweep_avg},
            "roofline_step": {"bound": "hbm", "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_step": step_bytes},
            "kernel_ms": {"sweep": sweep_avg, "posterior": float(np.mean(post_ms)),
                          "convert": float(np.mean(conv_ms)), "device_total": float(np.mean(tot_ms))},
            "batch_create_seconds": pack_s,
            "results_d2h": {"seconds": d2h_s, "bytes": d2h_bytes, "what": "arc log-posteriors + alpha + beta to pageable host memory"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ns, fo, args.seed, scales)
        print(json.dumps(out))
    batch.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
