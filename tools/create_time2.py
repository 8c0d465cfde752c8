"""klat_batch_create of the C2 batch from arrays already on the device (automatic layout)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
a = klat_amd.synth(20260001, np.full(10000, 2000, np.int32), np.full(10000, 3, np.int32), host_threads=16)
ctx = klat_amd.Ctx(0)
for rep in range(4):
    t2 = time.perf_counter(); b = klat_amd.Batch(ctx, a, host_threads=16); ctx.sync(); t3 = time.perf_counter()
    print("rep %d: klat_batch_create (host arrays in) %.1f ms, level lattices %d" % (rep, (t3 - t2) * 1e3, b.info()["level_lattices"]), flush=True)
    b.close()
