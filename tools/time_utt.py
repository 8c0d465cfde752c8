"""Times klat_utterance_scores (kwsbin2/lattice-word-index-utterance.cc's per-word queries) on
C2-shaped lattices: every distinct word of every lattice is one query."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python"))
import klat_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ns = np.full(U, 2000, np.int32); fo = np.full(U, 3, np.int32)
a = klat_amd.synth(20260001, ns, fo)
ctx = klat_amd.Ctx(0)
b = klat_amd.Batch(ctx, a)
b.forward_backward(1.0, 0.1, 0.0); ctx.sync()
off, words = [0], []
for u in range(U):
    w = np.unique(a["olabel"][a["arc_off"][u]:a["arc_off"][u + 1]])
    w = w[w != 0]
    words.append(w); off.append(off[-1] + len(w))
words = np.concatenate(words).astype(np.int32)
for rep in range(2):
    t0 = time.perf_counter(); lp = b.utterance_scores(off, words); dt = time.perf_counter() - t0
    print("%d lattices, %d queries (%.0f per lattice): %.1f ms -> %.2f M queries/s, %.1f G (arc x query) terms/s; mean logp %.3f" % (
        U, len(words), len(words) / U, 1e3 * dt, len(words) / dt / 1e6, float(np.sum(np.diff(off) * np.diff(a["arc_off"]))) / dt / 1e9, lp.mean()))
