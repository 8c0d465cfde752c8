"""Step time of the hot path on the other BASELINE shapes, at the shard one GPU of
an 8-GPU node would hold: C3 (50 000 x 2 000 states / 8), C4 (100 000 x 5 000 / 8), C5 (20 000
lattices, sizes log-uniform in [200, 20 000] states, fanout in {2,3,4,6}; whole and / 8)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd

ctx = klat_amd.Ctx(0)


def c5(U):
    rng = np.random.RandomState(20260004)
    ns = np.exp(rng.uniform(np.log(200), np.log(20000), U)).astype(np.int32)
    fo = rng.choice([2, 3, 4, 6], U).astype(np.int32)
    return ns, fo


cases = [("C2 10000x2000 f3", 20260001, np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32)),
         ("C3/8 6250x2000 f3", 20260002, np.full(6250, 2000, np.int32), np.full(6250, 3, np.int32)),
         ("C4/8 12500x5000 f3", 20260003, np.full(12500, 5000, np.int32), np.full(12500, 3, np.int32)),
         ("C5/8 2500 skewed", 20260004) + c5(2500),
         ("C5 20000 skewed", 20260004) + c5(20000)]
only = [x for x in sys.argv[1:] if not x.startswith('G=')]
force_g = [int(x[2:]) for x in sys.argv[1:] if x.startswith('G=')]
force_g = force_g[0] if force_g else 0
for name, seed, ns, fo in cases:
    if only and not any(o in name for o in only):
        continue
    a = klat_amd.synth(seed, ns, fo, host_threads=16)
    t0 = time.perf_counter()
    b = klat_amd.Batch(ctx, a, host_threads=16, lanes_per_lattice=force_g)
    tc = time.perf_counter() - t0
    info = b.info()
    sw, po, tot = [], [], []
    for it in range(60):
        b.forward_backward(1.0, 0.1, 0.0)
        if it >= 30:
            ctx.sync()
            t = b.timings()
            sw.append(t["sweep_ms"]); po.append(t["posterior_ms"]); tot.append(t["total_ms"])
    total, tf, tb = b.totals()
    assert np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-6
    A, S = info["num_arcs"], info["num_states"]
    sweep, step = float(np.median(sw)), float(np.median(tot))
    print("%-20s arcs %.3g states %.3g G=%d W=%d jobs %d (%.1f/CU) rounds %d exact %d | create %.0f ms | sweep %.3f ms (%.2f TB/s alg) post %.3f ms step %.3f ms -> %.1f G arcs/s, step frac %.2f" % (
        name, A, S, info["lanes_per_lattice"], info["ring_slots"], info["num_jobs"], info["num_jobs"] / 256.0, info["max_rounds"],
        info.get("exact_jobs", -1), tc * 1e3, sweep, (32.0 * A + 32.0 * S) / sweep / 1e9, float(np.median(po)), step, A / step / 1e6,
        (56.0 * A + 48.0 * S) / (step * 1e-3) / 8e12), flush=True)
    b.close()
    del a, b
