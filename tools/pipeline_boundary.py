"""Boundary-inclusive throughput (SURVEY.md 8d "include H2D/D2H separately reported"): the caller
holds HOST lattices and wants HOST results.  Serial: create (H2D + packing) -> sweep -> fetch, one
batch after the other on one context.  Pipelined: two host threads, each with its own context and
page-locked buffers, alternate over the batches, so that the upload + packing of batch k+1 and the
download of batch k-1 overlap the sweep of batch k (the C ABI is re-entrant: one klat_ctx per
thread).  Prints arcs/s for both."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd


def run(U=10000, S=2000, nbatches=12, pinned=True, threads=2, quiet=False):
    ns = np.full(U, S, np.int32); fo = np.full(U, 3, np.int32)
    host = [klat_amd.synth(20260001, ns, fo, first_utt=i * U, host_threads=16) for i in range(min(nbatches, 2))]
    if pinned:
        host = [{k: klat_amd.pinned_copy(v) for k, v in h.items()} for h in host]
    A = int(host[0]["arc_off"][-1]); St = int(host[0]["state_off"][-1])
    alloc = klat_amd.pinned_empty if pinned else (lambda shape, dt: np.empty(shape, dt))

    def one(ctx, k, post, al, be):
        b = klat_amd.Batch(ctx, host[k % len(host)], host_threads=8)
        b.forward_backward(1.0, 0.1, 0.0)
        b.arc_logpost(out=post)
        b.alpha_beta(out=(al, be))
        b.close()

    def worker(w, todo, out, start):
        ctx = klat_amd.Ctx(0)
        post, al, be = alloc(A, np.float64), alloc(St, np.float64), alloc(St, np.float64)
        for k in range(2):  # warm-up: the context's device blocks get allocated, the result pages touched
            one(ctx, k, post, al, be)
        start.wait()
        for k in todo:
            one(ctx, k, post, al, be)
        out[w] = float(post[0])
        ctx.close()

    res = {}
    for nt in sorted({1, threads}):
        out = [None] * nt
        start = threading.Barrier(nt + 1)
        ts = [threading.Thread(target=worker, args=(w, list(range(w, nbatches, nt)), out, start)) for w in range(nt)]
        for t in ts: t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        res[nt] = A * nbatches / dt
        if not quiet:
            print("%d host thread(s), %s buffers: %d batches of %d arcs in %.1f ms -> %.2f G arcs/s boundary-inclusive (%.1f ms per batch)"
                  % (nt, "page-locked" if pinned else "pageable", nbatches, A, 1e3 * dt, res[nt] / 1e9, 1e3 * dt / nbatches), flush=True)
    return res


if __name__ == "__main__":
    for pinned in (False, True):
        run(pinned=pinned)
