"""klat_viterbi on the C5/8 shard (BASELINE configs[4] mix), segment path on / off: wall time per call (20 calls)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, ROOT)
import klat_amd
from bench import c5_sizes
ns, fo = c5_sizes(2500)
arrays = klat_amd.synth(20260004, ns, fo, host_threads=16)
ctx = klat_amd.Ctx(0)
scales = (1.0, 0.1, 0.0)
ref = None
for label, env in (("streams", {"KLAT_CHAIN": "0"}), ("segments", {})):
    os.environ.pop("KLAT_CHAIN", None)
    os.environ.update(env)
    b = klat_amd.Batch(ctx, arrays, host_threads=16, fuse_scales=scales)
    for _ in range(5):
        b.viterbi(*scales)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        b.viterbi(*scales)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    fw, bw, best = b.viterbi_costs()
    if ref is None:
        ref = (fw, bw, best)
    err = max(float(np.max(np.abs(fw - ref[0])[np.isfinite(ref[0])])), float(np.max(np.abs(bw - ref[1])[np.isfinite(ref[1])])))
    print("%-9s klat_viterbi %.3f ms per call (%d lattices in segments)  max |diff| to the streams %.2e" % (label, dt * 1e3, b.info()["chain_lattices"], err), flush=True)
    b.close()
