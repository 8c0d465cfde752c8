"""Pure-Python interpreter of the packed lockstep arc streams (test infrastructure).

Mirrors sweep_kernel in kaldi-lattice-utils_amd/csrc/klat_kernels.hip term by term, so the
host packer and the scaled-linear arithmetic can be checked against the oracle on the CPU,
before GPU time is spent.  Never used by the product."""
import math

import numpy as np

REL_MASK = (1 << 27) - 1
FAR, LAST, WORD, NIL, FIRST = 1 << 27, 1 << 28, 1 << 29, 1 << 30, 1 << 31
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
    info = packed["info"]
    G, W = info["lanes_per_lattice"], info["ring_slots"]
    LPW = 64 // G
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
        for slot in range(LPW):
            li = packed["lanes"][int(job["lane_tab"]) + slot]
            base, S, u = int(li["state_base"]), int(li["num_states"]), int(li["lattice"])
            ring = {}
            p = 0
            accs = [(0.0, EXP_MIN)] * G
            nround = 0
            for r in range(int(job["nchunks"]) * 4):
                last = False
                for sub in range(G):
                    lane = slot * G + sub
                    ix = ((int(job["chunk_off"]) + r // 4) * 64 + lane) * 4 + r % 4
                    w = int(sw[ix])
                    if base < 0:
                        assert w == NIL, "empty slot must hold NIL terms only"
                        continue
                    if p >= (S if d else S + 1):
                        assert w == NIL, "padding after the last state must be plain NIL"
                        continue
                    assert bool(w & FIRST) == (nround == 0), "FIRST marks exactly the first round of a state"
                    c = _cost(sg[ix], sa[ix], bool(w & WORD), graph_scale, acoustic_scale, insertion_penalty)
                    rel = w & REL_MASK
                    dead = not (c < 1e30)
                    if not dead:
                        t = -c * LOG2E
                        k = round(t)
                        wm = 2.0 ** (t - k)
                    else:
                        k, wm = 0, 0.0
                    if w & NIL:
                        sm, se = 0.0, EXP_MIN
                    elif rel == 0:
                        sm, se = 0.5, 1
                    elif w & FAR:
                        assert rel >= W
                        q = p - rel
                        idx = base + (S - 1 - q if d else q)
                        sm, se = vm[d][idx], int(ve[d][idx])
                    else:
                        assert 1 <= rel <= W - 1, (rel, W)
                        sm, se = ring[(p - rel) & (W - 1)]
                    zero = bool(w & NIL) or dead or sm == 0.0
                    tm = 0.0 if zero else sm * wm
                    te = EXP_MIN if zero else se + k
                    accs[sub] = _acc(accs[sub][0], accs[sub][1], tm, te)
                    if w & LAST:
                        last = True
                    else:
                        assert not last, "LAST must be uniform over the G lanes of a lattice"
                nround += 1
                if base < 0 or not last:
                    continue
                nround = 0
                M, E = accs[0]
                for sub in range(1, G):
                    M, E = _acc(M, E, accs[sub][0], accs[sub][1])
                m, ex = math.frexp(M)
                e = EXP_MIN if M == 0.0 else E + ex
                ring[p & (W - 1)] = (m, e)
                if d == 0 and p == S:
                    totf[u] = _ln(m, e)
                else:
                    idx = base + (S - 1 - p if d else p)
                    assert np.isnan(val[d][idx]), "state finalised twice"
                    val[d][idx] = _ln(m, e)
                    vm[d][idx], ve[d][idx] = m, e
                p += 1
                accs = [(0.0, EXP_MIN)] * G
            if base >= 0:
                assert p == (S + 1 if d == 0 else S), (p, S, d)
    totb = np.array([val[1][int(state_off[u])] if state_off[u + 1] > state_off[u] else 0.0 for u in range(U)])
    return val[0], val[1], totf, totb
