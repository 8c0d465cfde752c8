"""Host-side cost of klat_batch_create on the C2 batch: stream packing vs the whole call (packing +
uploads), first and repeated calls."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
a = klat_amd.synth(20260001, ns, fo)
ctx = klat_amd.Ctx(0)
for rep in range(3):
    t0 = time.perf_counter(); p = klat_amd.pack_host(a, lanes_per_lattice=4, ring_slots=32); t1 = time.perf_counter()
    nbytes = p["sw"].nbytes * 3
    del p
    t2 = time.perf_counter(); b = klat_amd.Batch(ctx, a, lanes_per_lattice=4, ring_slots=32); ctx.sync(); t3 = time.perf_counter()
    print("rep %d: pack_host (incl. copy to numpy) %.3f s; klat_batch_create %.3f s; streams %.2f GB, batch arrays %.2f GB" % (
        rep, t1 - t0, t3 - t2, nbytes / 1e9, sum(v.nbytes for v in a.values()) / 1e9), flush=True)
    del b
