"""wall clock per step on C2 and the C5/8 shard (300 steps behind 150 untimed ones), in a process of its own each time"""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    for rep in range(3):
        env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
        print("run", rep, subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, ROOT)
import klat_amd
from bench import c5_sizes
ctx = klat_amd.Ctx(0)
scales = (1.0, 0.1, 0.0)
out = []
for name, ns, fo, seed in (("C2", np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32), 20260001), ("C5/8",) + c5_sizes(2500) + (20260004,)):
    arrays = klat_amd.synth(seed, ns, fo, host_threads=16)
    b = klat_amd.Batch(ctx, arrays, host_threads=16, fuse_scales=scales)
    for _ in range(150):
        b.forward_backward(*scales)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(300):
        b.forward_backward(*scales)
    ctx.sync()
    out.append("%s %.4f ms" % (name, (time.perf_counter() - t0) / 300 * 1e3))
    b.close()
print("  ".join(out))
