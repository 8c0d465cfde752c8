"""(A checker script, not a pytest module: it lives with the tests because it calls the oracle.)
Stress test of the FAR path (sources older than the ring) under a full chip: 6 000 lattices with
every third arc stretched over 17-60 states, four layouts; flow properties over all arcs and oracle
parity on a sample."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
from test_gpu_parity import _flow_properties, _oracle_batch, _check
rng = np.random.RandomState(3)
U = 6000
ns = rng.randint(400, 900, size=U).astype(np.int32)
fo = rng.choice([2, 3, 4], size=U).astype(np.int32)
arrays = klat_amd.synth(123, ns, fo)
lat = np.repeat(np.arange(U), np.diff(arrays["arc_off"]))
src, dst = arrays["src"], arrays["dst"].copy()
stretch = (np.arange(len(dst)) % 3 == 1) & (dst != src + 1)
dst[stretch] = np.minimum(src[stretch] + rng.randint(17, 60, size=int(stretch.sum())), ns[lat][stretch] - 1)
arrays["dst"] = dst.astype(np.int32)
ctx = klat_amd.Ctx(0)
for G, W in ((4, 16), (4, 32), (1, 16), (2, 8)):
    b = klat_amd.Batch(ctx, arrays, lanes_per_lattice=G, ring_slots=W)
    info = b.info()
    for rep in range(3):
        b.forward_backward(1.0, 0.1, 0.0)
    alpha, beta = b.alpha_beta(); total, tf, tb = b.totals(); post = b.arc_logpost()
    print("G %d W %d: far terms %d of %d slots, exact jobs %d, sweep %.3f ms" % (G, W, info["far_terms"], info["fwd_slots"] + info["bwd_slots"], b.exact_sweep_jobs(), b.timings()["sweep_ms"]), flush=True)
    b.close()
    _flow_properties(arrays, alpha, beta, total, tf, tb, post)
    sub = {k: v for k, v in arrays.items()}
    n = 40
    ns_, na_ = int(arrays["state_off"][n]), int(arrays["arc_off"][n])
    sub["state_off"], sub["arc_off"] = arrays["state_off"][:n + 1], arrays["arc_off"][:n + 1]
    for k in ("src", "dst", "ilabel", "olabel", "graph", "acoustic"): sub[k] = arrays[k][:na_]
    for k in ("final_graph", "final_acoustic", "state_time"): sub[k] = arrays[k][:ns_]
    wa, wb, wt, wpost = _oracle_batch(sub, 1.0, 0.1, 0.0)
    _check(alpha[:ns_], wa, 1e-7); _check(beta[:ns_], wb, 1e-7); _check(post[:na_], wpost, 1e-7)
print("FAR stress ok")
