"""Host stages of klat_batch_create on device-resident C2 arrays (KLAT_PACK_PROFILE=1 prints them to stderr) and its wall time."""
import os, sys, time
os.environ["KLAT_PACK_PROFILE"] = "1"
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
arrays = klat_amd.synth(20260001, np.full(U, 2000, np.int32), np.full(U, 3, np.int32), host_threads=16)
ctx = klat_amd.Ctx(0)
dev, keep = dict(arrays), []
for name in ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_graph", "final_acoustic", "state_time"):
    t = torch.from_numpy(np.ascontiguousarray(arrays[name])).cuda()
    keep.append(t)
    dev[name] = int(t.data_ptr())
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    b = klat_amd.Batch(ctx, dev, fuse_scales=(1.0, 0.1, 0.0))
    t1 = time.perf_counter()
    print("rep %d: %.2f ms wall; %s" % (rep, (t1 - t0) * 1e3, b.pack_timings()), flush=True)
    b.close()
