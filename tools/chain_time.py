"""C5/8 shard (BASELINE configs[4] mix) through the streams alone and with the long lattices in segments:
sweep / step time (medians of 20 steps), the plan, the pack timings.  usage: chain_time.py [lattices]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
sys.path.insert(0, ROOT)
import klat_amd  # noqa: E402
from bench import c5_sizes  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
ns, fo = c5_sizes(U)
arrays = klat_amd.synth(20260004, ns, fo, host_threads=16)
ctx = klat_amd.Ctx(0)
scales = (1.0, 0.1, 0.0)
ref = None
CASES = [("streams", {"KLAT_CHAIN": "0"}), ("auto", {}), ("seg128", {"KLAT_CHAIN_SEG": "128"}), ("seg512", {"KLAT_CHAIN_SEG": "512"}), ("all", {"KLAT_CHAIN": "1"})]
if len(sys.argv) > 2:  # segment lengths to scan instead: chain_time.py 2500 64,96,128,160,192,256
    CASES = [("streams", {"KLAT_CHAIN": "0"})] + [("seg%s" % v, {"KLAT_CHAIN_SEG": v}) for v in sys.argv[2].split(",")]
for label, env in CASES:
    for k in ("KLAT_CHAIN", "KLAT_CHAIN_SEG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    b = klat_amd.Batch(ctx, arrays, host_threads=16, fuse_scales=scales)
    t1 = time.perf_counter()
    for _ in range(30):
        b.forward_backward(*scales)
    hist = ctx.timing_history(20)
    sweep = float(np.median([t["sweep_ms"] for t in hist]))
    step = float(np.median([t["total_ms"] for t in hist]))
    alpha, beta = b.alpha_beta()
    total, tf, tb = b.totals()
    info = b.info()
    A, S = info["num_arcs"], info["num_states"]
    if ref is None:
        ref = (alpha, beta, total)
    err = max(float(np.max(np.abs(alpha - ref[0])[np.isfinite(ref[0])])), float(np.max(np.abs(beta - ref[1])[np.isfinite(ref[1])])),
              float(np.max(np.abs(total - ref[2]))))
    print("%-8s sweep %.3f ms  step %.3f ms  step_frac %.3f  chain %d lattices / %d segments  classes %d jobs %d  create %.1f ms  pack %s  max|diff| %.2e  fwd=bwd %.1e"
          % (label, sweep, step, (56.0 * A + 48.0 * S) / (step * 1e-3) / 1e9 / 8000.0, info["chain_lattices"], info["chain_segments"],
             info["num_layouts"], info["num_jobs"], (t1 - t0) * 1e3, {k: round(v, 3) for k, v in b.pack_timings().items()}, err,
             float(np.max(np.abs(tf - tb)))), flush=True)
    b.close()
