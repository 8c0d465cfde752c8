"""N>1 path on the CPU: utterance sharding by cumulative arcs and the gather of index records to
rank 0, run as two real processes over gloo (world_size 2).  The per-rank compute is replaced by
the oracle here (no GPU in this container); what is under test is the sharding/ordering/gather
logic that bench.py and the multi-GPU drivers share."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import pytest  # noqa: E402

import klat_amd  # noqa: E402
from klat_amd import dist as kdist  # noqa: E402


def _partition_numpy(arc_counts, world_size):
    """Independent restatement of klat_partition_by_arcs (what round 1 shipped in Python)."""
    arc_counts = np.asarray(arc_counts, dtype=np.int64)
    n = len(arc_counts)
    csum = np.concatenate([[0], np.cumsum(arc_counts + 1)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = int(total) * r // world_size
        b = int(np.searchsorted(csum, target, side="left"))
        if 0 < b <= n and target - int(csum[b - 1]) < int(csum[b]) - target:  # the boundary nearer the target
            b -= 1
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


def test_partition_matches_independent_restatement():
    rng = np.random.RandomState(3)
    for trial in range(50):
        n = int(rng.randint(0, 200))
        arcs = rng.randint(0, 50000, size=n).astype(np.int64) if trial % 3 else np.zeros(n, np.int64)
        for world in (1, 2, 3, 5, 8, 16):
            assert kdist.partition_by_arcs(arcs, world).tolist() == _partition_numpy(arcs, world).tolist()


def test_partition_rejects_bad_arguments():
    with pytest.raises(klat_amd.KlatError):
        kdist.partition_by_arcs([1, -2, 3], 2)
    with pytest.raises(klat_amd.KlatError):
        kdist.partition_by_arcs([1, 2, 3], 0)


def test_partition_by_arcs_balances_and_keeps_order():
    rng = np.random.RandomState(0)
    arcs = np.exp(rng.uniform(np.log(200), np.log(60000), size=1000)).astype(np.int64)  # config-5 style skew
    for world in (1, 2, 3, 8):
        b = kdist.partition_by_arcs(arcs, world)
        assert b[0] == 0 and b[-1] == len(arcs) and np.all(np.diff(b) >= 0) and len(b) == world + 1
        loads = np.array([arcs[b[r]:b[r + 1]].sum() for r in range(world)])
        assert loads.max() <= arcs.sum() / world + arcs.max() + 1
    # a heavy trailing utterance must not starve the ranks behind it of everything before it (the
    # boundary is the prefix NEAREST the target, not the first one at or above it)
    assert kdist.partition_by_arcs([50, 50, 50, 50, 400], 2).tolist() == [0, 4, 5]
    b = kdist.partition_by_arcs([10, 10, 10, 1000], 4)
    assert b.tolist() == [0, 3, 3, 3, 4] or b[1] < 4, b.tolist()
    # more ranks than utterances, and empty lattices
    b = kdist.partition_by_arcs([5, 0, 0], 8)
    assert b[0] == 0 and b[-1] == 3 and np.all(np.diff(b) >= 0)
    assert kdist.partition_by_arcs([], 4).tolist() == [0, 0, 0, 0, 0]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%(root)r, "kaldi-lattice-utils_amd", "python"))
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import klat_amd
    from klat_amd import dist as kdist
    from oracle import oracle as orc
    from test_gpu_accum import _split

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ns = np.array([40, 3, 90, 1, 25, 60, 7, 33, 12], np.int32)
    fo = np.array([3, 1, 2, 1, 4, 3, 2, 3, 1], np.int32)
    arrays = klat_amd.synth(777, ns, fo)          # every rank sees the same utterance list
    bounds = kdist.partition_by_arcs(np.diff(arrays["arc_off"]), world)
    lats = _split(arrays)

    def index(u):                                  # stand-in for the GPU path: the oracle
        lat = orc.scale_and_penalize(lats[u], 1.0, 0.1, 0.0)
        t = arrays["state_time"][arrays["state_off"][u]:arrays["state_off"][u + 1]]
        a, b, tot, _, _ = orc.alphas_betas(lat)
        return orc.segment_index(lat, t, a, b, tot)

    mine = [index(u) for u in range(bounds[rank], bounds[rank + 1])]
    rec = np.zeros(sum(len(m) for m in mine), klat_amd.SEGMENT_DTYPE)
    off = np.zeros(len(mine) + 1, np.int64)
    k = 0
    for i, m in enumerate(mine):
        for (l, t0, t1, lp) in m:
            rec[k] = (l, t0, t1, 0, lp)
            k += 1
        off[i + 1] = k
    ROOT_RANK = %(root_rank)d
    comm = kdist.Comm.over_group()
    assert comm.info() == {"rank": rank, "world": world, "transport_ranks": world}
    all_rec, all_off = comm.gather_records(rec, off, root=ROOT_RANK)
    again_rec, again_off = comm.gather_records(rec, off, root=ROOT_RANK)   # the communicator is reusable
    if rank == ROOT_RANK:
        assert again_off.tolist() == all_off.tolist() and again_rec.tobytes() == all_rec.tobytes()
        assert comm.stats()["payload_bytes"] == (len(all_rec) - len(rec)) * 24 + 8 * (len(ns) - len(mine))
    comm.close()
    if rank == ROOT_RANK:
        want = [index(u) for u in range(len(ns))]
        assert len(all_off) == len(ns) + 1, all_off
        for u, w in enumerate(want):
            got = [(int(r["label"]), int(r["t0"]), int(r["t1"]), float(r["logp"])) for r in all_rec[all_off[u]:all_off[u + 1]]]
            assert got == w, (u, got[:3], w[:3])
        print("GATHER_OK", len(all_rec))
    else:
        assert all_rec is None
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world,root_rank,port", [(2, 0, 29611), (3, 2, 29613)])
def test_sharded_index_gather(tmp_path, world, root_rank, port):
    """The C++ gather (klat_gather_records over the host transport) as `world` real processes; with
    three ranks the writing rank is not rank 0 and the middle rank's share can be tiny."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "root_rank": root_rank})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "GATHER_OK" in p.stdout


def test_bench_gpus_flag_starts_the_ranks(tmp_path):
    """`python bench.py --gpus 2` must start two ranks itself (the driver's N>1 invocation when it
    does not go through torchrun).  KLAT_BENCH_DRYRUN=1 keeps it off the GPU: the ranks rendezvous
    over gloo and run the sharding + C++ gather plumbing."""
    import json
    env = dict(os.environ, KLAT_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert {k: out["index_gather"][k] for k in ("ranks", "order_restored")} == {"ranks": 2, "order_restored": True}
    # at N > 1 the fields the driver parses are the strong-scaling step on configs[3] (north_star's question), the weak
    # configs[1] line moves aside
    assert out["scaling"] == "strong" and "configs[3]" in out["config"]["workload"] and out["config"]["ranks"] == 2
    assert out["promoted"].startswith("strong_c4.node_arcs_per_s_step") and out["weak_c2"]["scaling"] == "weak"
    assert out["transport_ranks"] == 2
    # the strong-scaling leg (north_star's claim: the SAME configs[3] list at every N): the two ranks' ranges tile one list
    assert out["strong_c4"]["scaling"] == "strong" and out["strong_c4"]["ranks"] == 2 and out["strong_c4"]["ranges_tile_the_list"]
    assert out["strong_c4"]["lattices"] == 96 and out["strong_c4"]["total_arcs"] > 0


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, KLAT_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
