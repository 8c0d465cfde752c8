"""C5/8 timing as bench.py's shapes leg does it (torch initialised or not: argv[1] = 1 / 0)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if len(sys.argv) > 1 and sys.argv[1] == "1":
    import torch
    torch.cuda.init(); x = torch.zeros(10, device="cuda"); torch.cuda.synchronize()
import klat_amd
import bench
ns, fo = bench.c5_sizes(2500)
ctx = klat_amd.Ctx(0)
for rep in range(3):
    r = bench.time_shape(ctx, klat_amd, "c5", 20260004, ns, fo, 16)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("sweep_ms", "step_ms", "step_frac", "lattices_in_segments", "layout_classes")}, flush=True)
