"""Few lattices of one size: klat_forward_backward with the segment path as klat_batch_create chooses it, and switched off
(sweep_ms medians of 20 steps): the choice must not cost small batches anything."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, ROOT)
import klat_amd
ctx = klat_amd.Ctx(0)
scales = (1.0, 0.1, 0.0)
for U, S in ((1, 600), (1, 2000), (1, 5000), (1, 20000), (8, 3000), (64, 5000), (256, 5000), (2000, 2000)):
    arrays = klat_amd.synth(7, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=8)
    row = []
    for env in ({}, {"KLAT_CHAIN": "0"}):
        os.environ.pop("KLAT_CHAIN", None)
        os.environ.update(env)
        b = klat_amd.Batch(ctx, arrays, fuse_scales=scales)
        for _ in range(30):
            b.forward_backward(*scales)
        hist = ctx.timing_history(20)
        row.append((float(np.median([t["sweep_ms"] for t in hist])), b.info()["chain_lattices"]))
        b.close()
    print("%5d lattices x %6d states: chosen %.3f ms (%d in segments)   streams %.3f ms" % (U, S, row[0][0], row[0][1], row[1][0]), flush=True)
