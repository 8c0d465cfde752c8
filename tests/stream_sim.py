"""Pure-Python interpreter of the packed lockstep arc streams (test infrastructure).

Mirrors the producer/consumer pair of sweep_kernel (kaldi-lattice-utils_amd/csrc/klat_sweep.hip)
term by term -- ring cells, register bypass, constant-one and FAR sources, per-round commit --
so the host packer and the scaled-linear arithmetic can be checked against the oracle on the
CPU, before GPU time is spent.  Never used by the product."""
import math

import numpy as np

CELL_MASK, WR_SHIFT = 0xFFF, 12
COMMIT, NIL, ONE = 1 << 24, 1 << 25, 1 << 26
REL_MASK = (1 << 27) - 1
FAR, LAST, FIRST, WORD, BYPASS = 1 << 27, 1 << 28, 1 << 29, 1 << 30, 1 << 31
EXP_MIN = -(1 << 29)
LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def _cost(g, a, word, gs, as_, pen):
    g = np.float32(g)
    a = np.float32(a)
    if gs != 1.0 or as_ != 1.0:
        if np.isinf(g) and g > 0:
            a = np.float32(np.inf)
        else:
            g = np.float32(float(np.float32(gs)) * float(g))
            a = np.float32(float(np.float32(as_)) * float(a))
    if pen != 0.0 and word:
        g = np.float32(g + np.float32(pen))
    return float(g) + float(a)


def _acc(M, E, tm, te):
    hi = max(E, te)
    return math.ldexp(M, max(E - hi, -3000)) + math.ldexp(tm, max(te - hi, -3000)), hi


def _ln(m, e):
    return -math.inf if m == 0.0 else e * LN2 + math.log(m)


def run(packed, state_off, graph_scale=1.0, acoustic_scale=1.0, insertion_penalty=0.0):
    """Returns alpha, beta (natural log, per state), tot_fwd, tot_bwd per lattice."""
    sw = packed["sw"].view(np.uint32)
    sg, sa = packed["sg"], packed["sa"]
    S_tot = int(state_off[-1])
    U = len(state_off) - 1
    val = [np.full(S_tot, np.nan), np.full(S_tot, np.nan)]
    vm = [np.zeros(S_tot), np.zeros(S_tot)]
    ve = [np.zeros(S_tot, np.int64), np.zeros(S_tot, np.int64)]
    totf = np.full(U, np.nan)
    for job in packed["jobs"]:
        d = int(job["dir"])
        G, W = (int(job["tag"]) >> 8) & 0xFF, (int(job["tag"]) >> 16) & 0xFFF  # the job's layout class
        LPW = 64 // G
        assert W * LPW + LPW <= CELL_MASK + 1
        ring = {}  # cell index -> (m, e); shared by the job's lattices like the LDS ring
        for slot in range(LPW):
            li = packed["lanes"][int(job["lane_tab"]) + slot]
            base, S, u = int(li["state_base"]), int(li["num_states"]), int(li["lattice"])
            npos = S if d else S + 1
            p = 0
            accs = [(0.0, EXP_MIN)] * G
            last_val = (0.0, EXP_MIN)
            nround = 0
            for r in range(int(job["nchunks"]) * 4):
                last = False
                commit = None
                for sub in range(G):
                    lane = slot * G + sub
                    AL = ((int(job["tag"]) >> 2) & 0x3F) + 1  # lanes per lattice whose units are stored
                    if sub >= AL:
                        # not stored: the kernel makes up a padding word with the FIRST / LAST bits of
                        # the lattice's first lane; nothing to interpret here
                        continue
                    ix = (int(job["chunk_off"]) + (r // 4) * (AL * LPW) + sub * LPW + slot) * 4 + r % 4
                    w = int(sw[ix])
                    if base < 0 or p >= npos:
                        assert w & NIL and not (w & (LAST | FIRST | COMMIT | FAR)), "padding must be plain NIL"
                        assert sg[ix] == np.inf, "NIL terms carry an infinite graph cost"
                        assert (w >> WR_SHIFT) & CELL_MASK == W * LPW + lane // G
                        continue
                    assert bool(w & FIRST) == (nround == 0), "FIRST marks exactly the first round of a state"
                    c = _cost(sg[ix], sa[ix], bool(w & WORD), graph_scale, acoustic_scale, insertion_penalty)
                    dead = not (c < 1e30)
                    if not dead:
                        t = -c * LOG2E
                        k = round(t)
                        wm = 2.0 ** (t - k)
                    else:
                        k, wm = 0, 0.0
                    if w & NIL:
                        assert sg[ix] == np.inf and dead, "NIL terms carry an infinite graph cost"
                        sm, se = 0.0, EXP_MIN
                        assert not (w & (ONE | BYPASS | FAR))
                    elif w & FAR:
                        rel = w & REL_MASK
                        assert rel >= W and not (w & (ONE | BYPASS))
                        q = p - rel
                        idx = base + (S - 1 - q if d else q)
                        sm, se = vm[d][idx], int(ve[d][idx])
                    elif w & ONE:
                        sm, se = 0.5, 1
                    elif w & BYPASS:
                        assert nround == 0 and p > 0, "BYPASS only in the first round of a state"
                        sm, se = last_val
                        assert ring[w & CELL_MASK] == last_val, "bypassed source must equal the ring cell"
                    else:
                        cell = w & CELL_MASK
                        assert cell % LPW == slot, "ring cell belongs to another lattice"
                        sm, se = ring[cell]
                    zero = bool(w & NIL) or dead or sm == 0.0
                    tm = 0.0 if zero else sm * wm
                    te = EXP_MIN if zero else se + k
                    accs[sub] = (tm, te) if nround == 0 else _acc(accs[sub][0], accs[sub][1], tm, te)
                    if w & LAST:
                        last = True
                    else:
                        assert not last, "LAST must be uniform over the G lanes of a lattice"
                    if w & COMMIT:
                        assert sub == 0 and (w & LAST)
                        commit = (w >> WR_SHIFT) & CELL_MASK
                    elif not (w & FAR):
                        assert (w >> WR_SHIFT) & CELL_MASK == W * LPW + lane // G, "non-commit lanes write their lattice's dummy cell"
                    elif (w & LAST) and sub == 0:
                        assert w & FAR, "the first lane of a LAST round commits (FAR terms: resolved by the kernel)"
                        commit = (p & (W - 1)) * LPW + slot
                if base < 0 or p >= npos:
                    continue
                nround += 1
                if not last:
                    continue
                nround = 0
                M, E = accs[0]
                for sub in range(1, G):
                    M, E = _acc(M, E, accs[sub][0], accs[sub][1])
                m, ex = math.frexp(M)
                e = EXP_MIN if M == 0.0 else E + ex
                assert commit == (p & (W - 1)) * LPW + slot, "commit cell must be the target state's ring cell"
                ring[commit] = (m, e)
                last_val = (m, e)
                if d == 0 and p == S:
                    totf[u] = _ln(m, e)
                else:
                    idx = base + (S - 1 - p if d else p)
                    assert np.isnan(val[d][idx]), "state finalised twice"
                    val[d][idx] = _ln(m, e)
                    vm[d][idx], ve[d][idx] = m, e
                p += 1
            if base >= 0:
                assert p == npos, (p, S, d)
    totb = np.array([val[1][int(state_off[u])] if state_off[u + 1] > state_off[u] else 0.0 for u in range(U)])
    return val[0], val[1], totf, totb
