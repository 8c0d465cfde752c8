"""klat_forward_backward_host on the C2 batch under the conditions of bench.py: with / without torch in the process,
GPU_MAX_HW_QUEUES from the environment.  python tools/host_pipe_probe.py [torch]"""
import os, sys, time
import numpy as np
if "torch" in sys.argv:
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U, S = 10000, 2000
ctx0 = klat_amd.Ctx(0)
arrays = klat_amd.synth(20260001, np.full(U, S, np.int32), np.full(U, 3, np.int32), host_threads=16)
compact = klat_amd.compact_arrays(arrays)
pin = {k: (klat_amd.pinned_copy(v) if k not in ("state_off", "arc_off") else v) for k, v in compact.items()}
St, A = int(arrays["state_off"][-1]), int(arrays["arc_off"][-1])
out = {"alpha": klat_amd.pinned_empty(St, np.float64), "beta": klat_amd.pinned_empty(St, np.float64),
       "total": klat_amd.pinned_empty(U, np.float64), "post": klat_amd.pinned_empty(A, np.float64)}
for which in ("own context", "the first context"):
    ctx = klat_amd.Ctx(0) if which == "own context" else ctx0
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        klat_amd.forward_backward_host(ctx, pin, (1.0, 0.1, 0.0), pieces=0, fuse=True, out=out, host_threads=16)
        t.append(time.perf_counter() - t0)
    print("%s, queues=%s, %s: %s ms" % ("torch" if "torch" in sys.argv else "no torch", os.environ.get("GPU_MAX_HW_QUEUES", "default"), which,
                                      " ".join("%.1f" % (x * 1e3) for x in t)), flush=True)
