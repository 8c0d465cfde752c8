"""KLAT_CHAIN_DEBUG dump -> when and where the segment path's waves ran (durations, concurrency, waves per CU / SIMD)"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
stride, words = int(raw[0]), int(raw[1])
da = raw[2:2 + stride].reshape(-1, 4)
dc = raw[2 + stride:2 + words].reshape(-1, 8)
for name, part in (("transfer", da), ("sweep", dc)):
    part = part[part[:, 1] > 0]
    if not len(part):
        continue
    t0 = part[:, 0].min()
    beg, end = (part[:, 0] - t0).astype(np.float64) / 100.0, (part[:, 1] - t0).astype(np.float64) / 100.0  # us
    dur = end - beg
    hw = part[:, 2].astype(np.int64)
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, part[:, 3].astype(np.int64) & 15
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("%s: %d waves, kernel span %.0f us, wave duration min %.0f median %.0f max %.0f us; starts: %.0f%% within 20 us, last start at %.0f us"
          % (name, len(part), end.max(), dur.min(), np.median(dur), dur.max(), 100.0 * np.mean(beg < 20), beg.max()))
    ts = np.linspace(0, end.max(), 9)[1:-1]
    print("   waves running at", ", ".join("%.0f us: %d" % (t, int(np.sum((beg <= t) & (end > t)))) for t in ts))
    ucu, ncu = np.unique(cuid, return_counts=True)
    print("   distinct CUs %d, waves per CU min %d median %d max %d; SIMD histogram %s" % (len(ucu), ncu.min(), int(np.median(ncu)), ncu.max(), np.bincount(simd, minlength=4)))
    if part.shape[1] == 8:
        fl, wt, tot = part[:, 4].astype(np.float64), part[:, 5].astype(np.float64), part[:, 6].astype(np.float64)
        print("   cycle counter: total median %.0f, in flush %.1f%%, waiting for the next batch %.1f%%" % (np.median(tot), 100 * fl.sum() / tot.sum(), 100 * wt.sum() / tot.sum()))
