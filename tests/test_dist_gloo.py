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
from klat_amd import dist as kdist  # noqa: E402


def test_partition_by_arcs_balances_and_keeps_order():
    rng = np.random.RandomState(0)
    arcs = np.exp(rng.uniform(np.log(200), np.log(60000), size=1000)).astype(np.int64)  # config-5 style skew
    for world in (1, 2, 3, 8):
        b = kdist.partition_by_arcs(arcs, world)
        assert b[0] == 0 and b[-1] == len(arcs) and np.all(np.diff(b) >= 0) and len(b) == world + 1
        loads = np.array([arcs[b[r]:b[r + 1]].sum() for r in range(world)])
        assert loads.max() <= arcs.sum() / world + arcs.max() + 1
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
    all_rec, all_off = kdist.gather_records(rec, off, dst=0)
    if rank == 0:
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


def test_sharded_index_gather_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "GATHER_OK" in p.stdout
