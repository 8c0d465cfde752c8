"""Probe for the abort at exit seen with more than one context: creates N
contexts with one small batch each, runs a step on each, closes everything explicitly or leaves it
to the exit handlers (argv[2] = "leak")."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
leak = len(sys.argv) > 2 and sys.argv[2] == "leak"
ns = np.full(64, 200, np.int32); fo = np.full(64, 3, np.int32)
ctxs = [klat_amd.Ctx(0) for _ in range(n)]
batches = [klat_amd.Batch(ctxs[i], klat_amd.synth(1, ns, fo, first_utt=i * 64)) for i in range(n)]
for s in range(8):
    batches[s % n].forward_backward(1.0, 0.1, 0.0)
for c in ctxs:
    c.sync()
print("ran", n, "contexts", flush=True)
if not leak:
    for b in batches: b.close()
    for c in ctxs: c.close()
    print("closed", flush=True)
