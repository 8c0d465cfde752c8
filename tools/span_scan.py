"""Sweep time against the share of long arcs: C2-shaped lattices (10 000 x 2 000 states, fanout 3)
with a fraction of the arcs stretched to spans log-uniform in [2, max_span] -- word lattices after
TopSort have such arcs (an arc that skips a whole alternative)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
base = klat_amd.synth(20260001, ns, fo, host_threads=16)
lat = np.repeat(np.arange(U), np.diff(base["arc_off"]))
rng = np.random.RandomState(1)
for frac, max_span in ((0.0, 0), (0.3, 100), (0.3, 1000), (0.1, 1900), (0.5, 1900)):
    a = dict(base)
    if frac > 0:
        src, dst = base["src"], base["dst"].copy()
        st = (rng.rand(len(dst)) < frac) & (dst != src + 1)
        span = np.exp(rng.uniform(np.log(2), np.log(max_span), size=int(st.sum()))).astype(np.int64)
        dst[st] = np.minimum(src[st] + span, ns[lat][st] - 1)
        a["dst"] = dst.astype(np.int32)
    b = klat_amd.Batch(ctx, a, host_threads=16, lanes_per_lattice=4 if W else 0, ring_slots=W)
    info = b.info()
    ts = []
    for it in range(40):
        b.forward_backward(1.0, 0.1, 0.0)
        if it >= 20:
            ctx.sync(); ts.append(b.timings()["sweep_ms"])
    total, tf, tb = b.totals()
    assert np.all(np.isfinite(total)) and float(np.max(np.abs(tf - tb))) < 1e-6
    slots = info["fwd_slots"] + info["bwd_slots"]
    print("stretched %.0f%% up to %4d states: G=%d W=%d layouts %d far terms %.2f%% of %d arcs, exact jobs %d, sweep %.3f ms" % (
        100 * frac, max_span, info["lanes_per_lattice"], info["ring_slots"], info["num_layouts"], 100.0 * info["far_terms"] / (2 * b.A), b.A,
        b.exact_sweep_jobs(), float(np.median(ts))), flush=True)
    b.close()
