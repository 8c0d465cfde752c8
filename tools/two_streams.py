"""Experiment: alternate two resident C2 batches on two contexts (streams) so that the sweep of
one overlaps the posterior of the other.  Prints arcs/s for 1 and 2 streams."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = 10000
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = [klat_amd.Ctx(0) for _ in range(nstreams)]
batches = [klat_amd.Batch(ctxs[i], klat_amd.synth(20260001, ns, fo, first_utt=i * U)) for i in range(nstreams)]
A = batches[0].info()["num_arcs"]
def run(steps):
    for s in range(steps):
        batches[s % nstreams].forward_backward(1.0, 0.1, 0.0)
    for c in ctxs:
        c.sync()
run(100)
t0 = time.perf_counter(); K = 400; run(K); dt = time.perf_counter() - t0
print("streams=%d  %.3f ms/step  %.2f G arcs/s" % (nstreams, 1e3 * dt / K, A * K / dt / 1e9), [b.timings() for b in batches][:1])
