import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import klat_amd
from test_gpu_parity import _oracle_batch
import test_gpu_chain as T

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scales = (1.0, 1.0, 0.0)
ns = [3000, 700, 5000, 40, 1300, 2 * seg, 2 * seg - 1, 4097, 1]
fo = [3, 2, 4, 3, 6, 3, 3, 1, 1]
arrays = T._lattices(20260005, ns, fo)
so = arrays["state_off"]
ctx = klat_amd.Ctx(0)
b, got = T._run(ctx, arrays, scales, KLAT_CHAIN=1, KLAT_CHAIN_SEG=seg)
n = __import__("ctypes").c_int32(0); klat_amd.lib().klat_batch_chain_fallback(b.h, __import__("ctypes").byref(n)); print(b.info(), "fallback bits", n.value)
wa, wb, wt, wpost = _oracle_batch(arrays, *scales)
alpha, beta, total, tf, tb, post = got
for name, g, w in (("alpha", alpha, wa), ("beta", beta, wb)):
    for u in range(len(ns)):
        gg, ww = g[so[u]:so[u + 1]], w[so[u]:so[u + 1]]
        fin = np.isfinite(ww) & np.isfinite(gg)
        d = np.where(fin, np.abs(gg - ww), 0.0)
        d[np.isneginf(gg) != np.isneginf(ww)] = 1e9
        bad = np.nonzero(d > 1e-8)[0]
        print(name, "lattice", u, "S", ns[u], "bad", len(bad), "first", bad[:6], "last", bad[-3:], "max", d.max() if len(d) else 0)
        if len(bad):
            i = bad[-1]
            print("   around last bad:", [(int(s), float(gg[s]), float(ww[s])) for s in range(max(0, i - 2), min(ns[u], i + 3))])
print("tot", total, wt, tf, tb)
