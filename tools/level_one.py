import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
ctx = klat_amd.Ctx(0)
a = klat_amd.synth(20260005, np.full(1, 2000, np.int32), np.full(1, 3, np.int32), wide=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
b = klat_amd.Batch(ctx, a)
for _ in range(3):
    b.forward_backward(1.0, 0.1, 0.0); ctx.sync()
for _ in range(3):
    b.viterbi(1.0, 0.1, 0.0); ctx.sync()
