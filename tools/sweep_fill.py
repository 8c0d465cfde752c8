"""The C2 sweep against the number of lattices: 10 000 lattices are 1 250 jobs (two waves each) on 1 024 SIMDs -- 226 SIMDs
carry two chain waves, the others one, and the kernel lasts as long as the former.  python tools/sweep_fill.py [lattices ...]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
for U in [int(x) for x in sys.argv[1:]] or (8192, 10000, 12288, 16384, 20480, 24576, 32768):
    a = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
    b = klat_amd.Batch(ctx, a, host_threads=16, fuse_scales=(1.0, 0.1, 0.0))
    for _ in range(60):
        b.forward_backward(1.0, 0.1, 0.0)
    ctx.sync()
    h = ctx.timing_history(40)
    sw = float(np.median([t["sweep_ms"] for t in h]))
    A, S = int(a["arc_off"][-1]), int(a["state_off"][-1])
    print("%6d lattices (%5d jobs): sweep %.4f ms, %.1f G arcs/s, %.2f of the 8 TB/s roofline" % (U, 2 * ((U + 15) // 16), sw, A / sw / 1e6, (32.0 * A + 32.0 * S) / (sw * 1e-3) / 8e12), flush=True)
    b.close()
