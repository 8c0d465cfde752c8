// klat_pack_lean.hip -- the stream packer's passes for lattices whose arcs are short (every arc at most
// kPackLeanMaxSpan states long, at most 127 terms a position): the same streams as klat_pack_device.hip writes, bit
// for bit, from fewer bytes.  What makes the general passes expensive is the order of the FORWARD stream: its terms
// are grouped by destination while the caller's arcs are grouped by source, so the general count pass leaves the
// arcs sorted by destination in HBM and the general fill pass gathers every arc through that table, once per
// direction, next to a second table for the backward order.  With short arcs none of it is needed: the in-arcs of
// the positions of one tile lie within `span` states before them in the caller's own order, so a tile reads ONE
// contiguous run of arcs, and all that has to be known about an arc beyond its own fields is its place among the
// in-arcs of its destination and among the out-arcs of its source -- a byte each.
//
//   pack_count_lean_kernel  one workgroup per lattice, its tables in LDS.  One pass over the arcs (16-byte loads):
//                           validation, in-degrees (LDS atomics), the first out-arc of every state from the arcs'
//                           own order, the longest arc, the batch's largest cost (what pack_maxcost_kernel computed
//                           in a pass of its own).  Prefix sums over the states: the first round of every position
//                           for the lanes the plan is EXPECTED to give the lattice (a.lat_g_pred: known from the
//                           offsets alone; what pack_rounds_kernel computed in a pass of its own after the plan), with
//                           the position's number of terms in the upper byte, the rounds for every lane count (the
//                           planner's input), and which position holds the first round of every tile (the fill
//                           pass's binary search).  A counting sort in LDS only -- it yields every arc's place among
//                           its destination's in-arcs; its place among its source's out-arcs (farthest first) comes
//                           from the neighbouring arcs.  Out: 4 bytes an arc (`meta`: length, the two places, the
//                           word bit) and 12 bytes a state.
//   pack_rounds_lean_kernel the same prefix sums again for the lattices whose lanes the plan chose otherwise.
//   pack_fill_lean_kernel   one workgroup per (job, run of tiles), a tile assembled in LDS and stored as whole lines
//                           like pack_fill_kernel's; a group's forward and backward job (the same lattices) side by
//                           side in one launch, so that the group's arcs come from HBM once.  Per tile one round trip for the arcs (destination or source,
//                           meta, the two costs: 16 bytes an arc, 16-byte loads of consecutive arcs) behind one for
//                           the positions' first rounds, and the second is asked for a tile ahead.
//
// A lattice the count pass finds unfit (a long arc, a position with more than 127 terms, a candidate of the level
// path) is left to the general passes: a.lean_ok[u] stays 0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <stdint.h>

#include "klat_device.h"
#include "klat_level.h"
#include "klat_math.h"
#include "klat_pack_common.h"
#include "klat_pack_device.h"

namespace klat {

namespace {

constexpr int kTL = 256;  // threads of every kernel here
constexpr uint32_t kRoundMask = 0xFFFFFFu;  // a position's first round in pf / pb; bits 24-30 its terms, bit 31 (pb) final state

// exclusive prefix sums of N ints per thread with one pair of barriers
template <int T, int N>
__device__ void block_scan_n(const int (&v)[N], int (&ex)[N], int (&tot)[N], int* sh /* [N * T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl[N];
#pragma unroll
  for (int c = 0; c < N; ++c) incl[c] = v[c];
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const int t = __shfl_up(incl[c], d, 64);
      if (lane >= d) incl[c] += t;
    }
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) {
#pragma unroll
    for (int c = 0; c < N; ++c) sh[c * (T / 64) + wave] = incl[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < N; ++c) {
    int off = 0, t = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
      const int x = sh[c * (T / 64) + w];
      if (w < wave) off += x;
      t += x;
    }
    tot[c] = t;
    ex[c] = off + incl[c] - v[c];
  }
}

// four consecutive elements of an array, quad q of it: one 16-byte load (the arrays start on 256-byte boundaries and are
// allocated four elements longer than the batch has arcs: the batch's last quad may reach past the last arc)
template <class V, class T>
__device__ __forceinline__ V load_quad(const T* base, int64_t q, int64_t) {
  return reinterpret_cast<const V*>(base)[q];
}

__device__ __forceinline__ int elem(const int4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ float elem(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ uint32_t elem(const uint4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// first entry of lattice u's rows in the tile tables (tpf / tpb): a lattice of S states owns S / 8 + 2 entries at
// least, one per 32 rounds -- the count pass leaves a lattice with more than 4 S rounds to the general passes
__device__ __forceinline__ int64_t tile_row(int64_t state_base, int64_t u) { return (state_base >> 3) + 2 * u; }
__device__ __forceinline__ int tile_cap(int S) { return (S >> 3) + 2; }

constexpr int kStatesPerThread = 4;

// The prefix sums of one lattice over its positions, for k = log2(lanes): forward (pf, tpf) from the in-degrees,
// backward (pb, tpb) from the out-degrees and the final flags.  nterm_f(p) / nterm_b(q) give the terms of forward
// position p < S and backward position q < S (and bit 31 of nterm_b: the position's state is final); nfinal the terms
// of forward position S.  Returns false when a lattice does not fit the tables (more than 4 S rounds).
// Also what the planner wants to know: rounds for every lane count and the most terms of a position, when `sums`.
template <int T, class FT, class FB, class PerF>
__device__ bool lean_prefix(const PackDevArgs& a, int u, int64_t s0, int S, int k, FT nterm_f, FB nterm_b, int nfinal, PerF per_forward,
                            int* sh, int (*fsum)[kNumLaneChoices], int (*bsum)[kNumLaneChoices], int* mtf_out, int* mtb_out) {
  const int tid = threadIdx.x;
  const int64_t sb = s0 + 2 * (int64_t)u, tb = tile_row(s0, u);
  const int cap = tile_cap(S);
  int32_t* pf = a.r0f + sb;
  int32_t* pb = a.r0b + sb;
  int32_t* tpf = a.tpf + tb;
  int32_t* tpb = a.tpb + tb;
  int mtf = 1, mtb = 1;
  bool fits = true;
  auto tiles = [&](int32_t* tp, int p, int r, int nr) {  // the tiles whose first round lies in position p (none, or one)
#pragma clang loop vectorize(disable) unroll(disable)
    for (int m = (r + kPackTileRounds - 1) / kPackTileRounds; m * kPackTileRounds < r + nr; ++m) {
      if (m < cap)
        tp[m] = p;
      else
        fits = false;
    }
  };
  // forward: positions 0 .. S - 1 (position S, the final weights, behind the loop)
  int carry[2] = {0, 0};
  for (int base = 0; base < S; base += T * kStatesPerThread) {
    const int p0 = base + tid * kStatesPerThread;
    int nt[kStatesPerThread], nr[kStatesPerThread], v[2] = {0, 0}, ex[2], tot[2];
#pragma unroll
    for (int e = 0; e < kStatesPerThread; ++e) {
      const int p = p0 + e;
      nt[e] = p < S ? nterm_f(p) : 0;
      nr[e] = p < S ? (max(nt[e], 1) + (1 << k) - 1) >> k : 0;
      v[0] += nr[e];
      v[1] += p < S ? nt[e] - (p == 0 ? 1 : 0) : 0;  // the in-degree
    }
    block_scan_n<T, 2>(v, ex, tot, sh);
    int r = carry[0] + ex[0], c = carry[1] + ex[1];
#pragma unroll
    for (int e = 0; e < kStatesPerThread; ++e) {
      const int p = p0 + e;
      if (p >= S) break;
      pf[p] = r | (min(nt[e], 127) << 24);
      tiles(tpf, p, r, nr[e]);
      per_forward(p, c);
      mtf = max(mtf, nt[e]);
      if (fsum)
        for (int kk = 0; kk < kNumLaneChoices; ++kk) (*fsum)[kk] += (max(nt[e], 1) + (1 << kk) - 1) >> kk;
      r += nr[e];
      c += nt[e] - (p == 0 ? 1 : 0);
    }
    carry[0] += tot[0], carry[1] += tot[1];
  }
  if (tid == 0) {
    const int nr = (max(nfinal, 1) + (1 << k) - 1) >> k;
    pf[S] = carry[0] | (min(nfinal, 127) << 24);
    pf[S + 1] = carry[0] + nr;
    tiles(tpf, S, carry[0], nr);
    mtf = max(mtf, nfinal);
    if (fsum)
      for (int kk = 0; kk < kNumLaneChoices; ++kk) (*fsum)[kk] += (max(nfinal, 1) + (1 << kk) - 1) >> kk;
    if (carry[0] + nr > 4 * S) fits = false;
  }
  // backward: position q = the state S - 1 - q
  int carry_b = 0;
  for (int base = 0; base < S; base += T * kStatesPerThread) {
    const int q0 = base + tid * kStatesPerThread;
    int nt[kStatesPerThread], nr[kStatesPerThread], v[1] = {0}, ex[1], tot[1];
#pragma unroll
    for (int e = 0; e < kStatesPerThread; ++e) {
      const int q = q0 + e;
      nt[e] = q < S ? nterm_b(q) : 0;
      nr[e] = q < S ? (max(nt[e] & 0xFFFFFF, 1) + (1 << k) - 1) >> k : 0;
      v[0] += nr[e];
    }
    block_scan_n<T, 1>(v, ex, tot, sh);
    int r = carry_b + ex[0];
#pragma unroll
    for (int e = 0; e < kStatesPerThread; ++e) {
      const int q = q0 + e;
      if (q >= S) break;
      const int n = nt[e] & 0xFFFFFF;
      pb[q] = r | (min(n, 127) << 24) | (nt[e] & (int)0x80000000);
      tiles(tpb, q, r, nr[e]);
      mtb = max(mtb, n);
      if (bsum)
        for (int kk = 0; kk < kNumLaneChoices; ++kk) (*bsum)[kk] += (max(n, 1) + (1 << kk) - 1) >> kk;
      r += nr[e];
    }
    carry_b += tot[0];
  }
  if (tid == 0) {
    pb[S] = carry_b;
    if (carry_b > 4 * S) fits = false;
  }
  if (mtf > 127 || mtb > 127) fits = false;
  if (mtf_out) *mtf_out = mtf, *mtb_out = mtb;
  return fits;
}

// T threads: 256 for the narrowest LDS class (four workgroups a CU), 512 and 1 024 for the wider ones (two, one): sixteen
// waves a CU whatever the lattices' size
template <int T>
__global__ __launch_bounds__(T) void pack_count_lean_kernel(PackDevArgs a, int lds_lo, int lds_hi) {
  extern __shared__ int32_t lds[];
  __shared__ int sh[2 * (T / 64)];
  __shared__ int sh_bad, sh_unfit, sh_span, sh_mr, sh_mtf, sh_mtb, sh_linked, sh_nfinal;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  __shared__ unsigned long long sh_mc;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  const int need = pack_lean_lds_bytes(S, A);
  if (need <= lds_lo || need > lds_hi || (a.lean_skip && a.lean_skip[u])) return;  // every lattice belongs to one launch: the narrowest LDS class it fits
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  if (S <= 0) {
    if (tid == 0) {
      if (S < 0 || A != 0)
        report_bad(a, 1 /* KLAT_ERR_INVALID_ARG */, u);
      else
        a.lean_ok[u] = 1;  // nothing to pack
    }
    return;
  }
  unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = a.prof ? __builtin_readcyclecounter() : 0;
  auto stamp = [&](int phase) {
    if (a.prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      pc[phase] += now - pt;
      pt = now;
    }
  };
  const int nl = (S + 31) / 32 + 1;
  int32_t* indeg = lds;                                          // [S + 2] in-degrees, then the counting sort's cursors
  uint32_t* link = reinterpret_cast<uint32_t*>(lds + (S + 2));   // [nl] a bit a state: it has an arc to its successor
  uint32_t* finb = link + nl;                                    // [nl] a bit a state: it is final
  uint16_t* bucket = reinterpret_cast<uint16_t*>(finb + nl);     // [A] arcs by destination
  uint16_t* row = bucket + ((A + 1) & ~1);                       // [S + 2] first out-arc (a lean lattice has fewer than 65 536 arcs)
  uint8_t* relb = reinterpret_cast<uint8_t*>(row + ((S + 3) & ~1));  // [A] length of every arc
  const int64_t sb = s0 + 2 * (int64_t)u;
  if (tid == 0) sh_bad = 0, sh_unfit = 0, sh_span = 1, sh_mr = 1, sh_mtf = 1, sh_mtb = 1, sh_linked = 0, sh_nfinal = 0, sh_mc = 0ull;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < S + 2; s += T) indeg[s] = 0;
  for (int s = tid; s < 2 * nl; s += T) link[s] = 0;
  __syncthreads();
  stamp(0);
  // ---- the arcs, four a lane and load
  const int a0i = (int)a0;  // (a batch has fewer than 2^31 arcs: 32-bit quad numbers; with 64-bit ones every address took four instructions more)
  const int q0 = a0i >> 2, q1 = (a0i + A + 3) >> 2;
  constexpr int kUn = 3;  // quads a thread has in flight
  float mc = 0.0f;
  int span = 1, bad = 0;
  for (int qb = q0 + tid; qb < q1; qb += T * kUn) {
    int4 vs[kUn], vd[kUn], vi[kUn];
    float4 vg[kUn], va[kUn];
    int vp[kUn];
#pragma unroll
    for (int x = 0; x < kUn; ++x) {
      const int q = min(qb + x * T, q1 - 1);
      vs[x] = load_quad<int4>(a.src, q, a.num_arcs);
      vd[x] = load_quad<int4>(a.dst, q, a.num_arcs);
      vp[x] = 4 * q > a0i ? a.src[4 * q - 1] : 0;
      if (a.fold_maxcost) {
        vg[x] = load_quad<float4>(a.graph, q, a.num_arcs);
        va[x] = load_quad<float4>(a.acoustic, q, a.num_arcs);
        if (a.fsp.penalized) vi[x] = load_quad<int4>(a.ilabel, q, a.num_arcs);
      }
    }
#pragma unroll
    for (int x = 0; x < kUn; ++x) {
      const int q = qb + x * T;
      if (q >= q1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e - a0i;
        const int s = elem(vs[x], e), t = elem(vd[x], e);
        const int prev = e == 0 ? vp[x] : elem(vs[x], e - 1);
        // (the tests as masks, the verdict in a register: nested branches with an LDS atomic in each cost every arc
        // a dozen scalar instructions)
        const bool mine = (unsigned)i < (unsigned)A;  // (else: a neighbouring lattice's arc)
        const bool ok_s = (unsigned)s < (unsigned)S && (i == 0 || s >= prev);  // else: malformed CSR
        const bool ok_t = t > s && t < S;                                      // else: not topologically sorted
        if (mine && !(ok_s && ok_t)) bad = max(bad, ok_s ? 1 : 2);
        if (mine && ok_s && ok_t) {
          atomicAdd(&indeg[t], 1);
          const int rel = t - s;
          span = max(span, rel);
          relb[i] = (uint8_t)min(rel, 255);
          if (rel == 1) atomicOr(&link[s >> 5], 1u << (s & 31));
          // arc i is the first out-arc of every state behind its predecessor's source up to its own: its own state, and
          // -- rarely -- states without arcs before it (the predecessor may be the malformed arc this pass exists to
          // reject: its source below zero must not move the fill before the table).  (Written as one loop hipcc made a
          // vectorised 128-bit fill of it with three levels of remainder loops: forty instructions for a trip count of one.)
          if (i == 0 || s != prev) {
            row[s] = (uint16_t)i;
#pragma clang loop vectorize(disable) unroll(disable)
            for (int qq = (i > 0 ? max(prev, -1) + 1 : 0); qq < s; ++qq) row[qq] = (uint16_t)i;
          }
          if (a.fold_maxcost) {
            const float c = arc_cost_abs_f(elem(vg[x], e), elem(va[x], e), a.fsp.penalized ? elem(vi[x], e) != 0 : false, a.fsp);
            if (c < 1.0e30f) mc = fmaxf(mc, c);
          }
        }
      }
    }
  }
  // the final weights: a bit a state, the largest cost (eight states a thread in flight: one after the other, a thread's
  // trips to memory were a fifth of this pass)
  for (int sbase = 0; sbase < S; sbase += 8 * T) {
    float fg[8], fa[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int s = min(sbase + x * T + tid, S - 1);
      fg[x] = a.final_graph[s0 + s], fa[x] = a.final_acoustic[s0 + s];
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int s = sbase + x * T + tid;
      if (s >= S) continue;
      if (!(fg[x] == INFINITY && fa[x] == INFINITY)) atomicOr(&finb[s >> 5], 1u << (s & 31));
      if (a.fold_maxcost) {
        const float c = arc_cost_abs_f(fg[x], fa[x], false, a.fsp);
        if (c < 1.0e30f) mc = fmaxf(mc, c);
      }
    }
  }
  if (bad) atomicMax(&sh_bad, bad);
  for (int d = 32; d > 0; d >>= 1) span = max(span, __shfl_xor(span, d, 64));
  if ((tid & 63) == 0) atomicMax(&sh_span, span);
  if (a.fold_maxcost) {  // (whatever becomes of the lattice below: the batch's largest cost covers every lattice)
    for (int d = 32; d > 0; d >>= 1) mc = fmaxf(mc, __shfl_xor(mc, d, 64));
    if ((tid & 63) == 0 && mc > 0.0f) atomicMax(&sh_mc, (unsigned long long)__double_as_longlong((double)mc));
  }
  stamp(1);
  __syncthreads();
  stamp(2);
  if (a.fold_maxcost && tid == 0 && sh_mc != 0ull) atomicMax(a.maxcost, sh_mc);  // (positive doubles order like their bit patterns)
  if (sh_bad) {
    if (tid == 0) report_bad(a, sh_bad == 2 ? 1 /* INVALID_ARG */ : 2 /* BAD_TOPOLOGY */, u);
    return;
  }
  {  // the states behind the last arc's source; the states linked to their successor
    const int last = A > 0 ? a.src[a0 + A - 1] : -1;
    for (int s = last + 1 + tid; s <= S + 1; s += T) row[s] = (uint16_t)A;
    int linked = 0;
    for (int w = tid; w < nl; w += T) linked += __popc(link[w]);
    int nfin = 0;
    for (int w = tid; w < nl; w += T) nfin += __popc(finb[w]);
    for (int d = 32; d > 0; d >>= 1) linked += __shfl_xor(linked, d, 64), nfin += __shfl_xor(nfin, d, 64);
    if ((tid & 63) == 0) atomicAdd(&sh_linked, linked), atomicAdd(&sh_nfinal, nfin);
  }
  __syncthreads();
  const int linked = sh_linked, nfinal = sh_nfinal;
  // unfit for these passes: a long arc; a lattice the level path may want (it reads the general pass's tables)
  if (sh_span > kPackLeanMaxSpan) return;
  if (a.level_probe && S >= kLevelMinStates && S <= kLevelMaxStates && !(linked > (S - linked + 1) * (S / kLevelMinWidth))) return;
  stamp(3);
  // ---- prefix sums over the positions
  const int k = a.lat_g_pred[u];
  int fsum[kNumLaneChoices] = {0}, bsum[kNumLaneChoices] = {0};
  int mtf, mtb, mr = sh_span;
  auto is_final = [&](int s) { return (finb[s >> 5] >> (s & 31)) & 1u; };
  const bool fits = lean_prefix<T>(
      a, u, s0, S, max(k, 0), [&](int p) { return indeg[p] + (p == 0 ? 1 : 0); },
      [&](int q) {
        const int s = S - 1 - q;
        const int fin = (int)is_final(s);
        return (row[s + 1] - row[s] + fin) | (fin << 31);
      },
      nfinal,
      [&](int p, int c) {
        indeg[p] = c;  // (the counting sort's cursor: the bucket's first entry)
        a.row[sb + p] = row[p];
        if (is_final(p)) mr = max(mr, S - p);
      },
      sh, &fsum, &bsum, &mtf, &mtb);
  // the final states in state order (forward position S holds one term for each)
  {
    int carry = 0;
    for (int base = 0; base < nl; base += T) {
      const int w = base + tid;
      const uint32_t bits = w < nl ? finb[w] : 0u;
      int v[1] = {(int)__popc(bits)}, ex[1], tot[1];
      block_scan_n<T, 1>(v, ex, tot, sh);
      int at = carry + ex[0];
      for (uint32_t m = bits; m; m &= m - 1) a.flist[s0 + at++] = w * 32 + __ffs(m) - 1;
      carry += tot[0];
    }
  }
  if (!fits) atomicMax(&sh_unfit, 1);
  for (int d = 32; d > 0; d >>= 1) {
    mr = max(mr, __shfl_xor(mr, d, 64));
    mtf = max(mtf, __shfl_xor(mtf, d, 64));
    mtb = max(mtb, __shfl_xor(mtb, d, 64));
#pragma unroll
    for (int kk = 0; kk < kNumLaneChoices; ++kk) fsum[kk] += __shfl_xor(fsum[kk], d, 64), bsum[kk] += __shfl_xor(bsum[kk], d, 64);
  }
  if ((tid & 63) == 0) {
    atomicMax(&sh_mr, mr);
    atomicMax(&sh_mtf, mtf);
    atomicMax(&sh_mtb, mtb);
    for (int kk = 0; kk < kNumLaneChoices; ++kk) atomicAdd(&sh_rounds[kk], fsum[kk]), atomicAdd(&sh_rounds[kNumLaneChoices + kk], bsum[kk]);
  }
  __syncthreads();
  if (sh_unfit) return;
  stamp(4);
  // ---- counting sort of the arcs by destination, in LDS: where an arc stands among the in-arcs of its destination
  for (int qb = q0 + tid; qb < q1; qb += T * kUn) {
    int4 vd[kUn];
#pragma unroll
    for (int x = 0; x < kUn; ++x) vd[x] = load_quad<int4>(a.dst, min(qb + x * T, q1 - 1), a.num_arcs);
#pragma unroll
    for (int x = 0; x < kUn; ++x) {
      const int q = qb + x * T;
      if (q >= q1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e - a0i;
        if (i < 0 || i >= A) continue;
        bucket[atomicAdd(&indeg[elem(vd[x], e)], 1)] = (uint16_t)i;
      }
    }
  }
  __syncthreads();
  stamp(5);
  // ---- meta: length | place among the destination's in-arcs (by arc number) << 8 | place among the source's out-arcs
  // (farthest destination first, equal destinations by arc number) << 16 | word << 24
  for (int qb = q0 + tid; qb < q1; qb += T * kUn) {
    int4 vs[kUn], vd[kUn], vi[kUn];
#pragma unroll
    for (int x = 0; x < kUn; ++x) {
      const int q = min(qb + x * T, q1 - 1);
      vs[x] = load_quad<int4>(a.src, q, a.num_arcs);
      vd[x] = load_quad<int4>(a.dst, q, a.num_arcs);
      vi[x] = load_quad<int4>(a.ilabel, q, a.num_arcs);
    }
#pragma unroll
    for (int x = 0; x < kUn; ++x) {
      const int q = qb + x * T;
      if (q >= q1) continue;
      uint32_t m[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e - a0i;
        m[e] = 0;
        if (i < 0 || i >= A) continue;
        const int s = elem(vs[x], e), t = elem(vd[x], e), rel = t - s;
        int jr = 0, orank = 0;
        {  // (the first four entries of either list at once: one after the other, the LDS round trips were this pass)
          const int yb = t > 0 ? indeg[t - 1] : 0, ye = indeg[t];
          int b4[4];
#pragma unroll
          for (int z = 0; z < 4; ++z) b4[z] = bucket[min(yb + z, ye - 1)];
#pragma unroll
          for (int z = 0; z < 4; ++z) jr += (yb + z < ye && b4[z] < i) ? 1 : 0;
#pragma clang loop vectorize(disable) unroll(disable)
          for (int y = yb + 4; y < ye; ++y) jr += (int)bucket[y] < i ? 1 : 0;
          const int rb = row[s], re = row[s + 1];
          int r4[4];
#pragma unroll
          for (int z = 0; z < 4; ++z) r4[z] = relb[min(rb + z, re - 1)];
#pragma unroll
          for (int z = 0; z < 4; ++z) orank += (rb + z < re && (r4[z] > rel || (r4[z] == rel && rb + z < i))) ? 1 : 0;
#pragma clang loop vectorize(disable) unroll(disable)
          for (int y = rb + 4; y < re; ++y) {
            const int ry = relb[y];
            orank += (ry > rel || (ry == rel && y < i)) ? 1 : 0;
          }
        }
        m[e] = (uint32_t)rel | ((uint32_t)jr << 8) | ((uint32_t)orank << 16) | (elem(vi[x], e) != 0 ? 1u << 24 : 0u);
      }
      if (4 * q >= a0i && 4 * q + 4 <= a0i + A) {
        reinterpret_cast<uint4*>(a.meta)[q] = make_uint4(m[0], m[1], m[2], m[3]);
      } else {  // (a quad shared with a neighbouring lattice)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * q + e >= a0i && 4 * q + e < a0i + A) a.meta[4 * q + e] = m[e];
      }
    }
  }
  stamp(6);
  if (a.prof && tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[i], pc[i]);
  if (tid < kNumLaneChoices) {
    a.lat_rounds[u].fwd[tid] = sh_rounds[tid];
    a.lat_rounds[u].bwd[tid] = sh_rounds[kNumLaneChoices + tid];
  }
  if (tid == 0) {
    a.row[sb + S] = A;
    a.row[sb + S + 1] = linked;  // (the spare entry of the row table: the level planner's chain test)
    atomicMax(&a.status[2], sh_mr);
    a.lat_rounds[u].max_rel = sh_mr;
    a.lat_rounds[u].max_terms = min(sh_mtf, 64) | (min(sh_mtb, 64) << 16);
    a.lat_rounds[u].max_span = sh_span;
    a.lean_ok[u] = 1;
  }
}

// The prefix sums again, from the terms the count pass left in the upper bytes, for the lattices whose lanes the plan
// did not choose as expected (a.reround[u] != 0).
__global__ __launch_bounds__(kTL) void pack_rounds_lean_kernel(PackDevArgs a) {
  __shared__ int sh[2 * (kTL / 64)];
  const int u = blockIdx.x;
  if (!a.reround[u]) return;
  const int64_t s0 = a.state_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  if (S <= 0 || a.lat_g[u] < 0) return;
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* pf = a.r0f + sb;
  const int32_t* pb = a.r0b + sb;
  const int nfinal = (pf[S] >> 24) & 0x7F;
  // (every thread reads its own positions' old entries before it writes them: lean_prefix loads nterm first)
  (void)lean_prefix<kTL>(
      a, u, s0, S, a.lat_g[u], [&](int p) { return (pf[p] >> 24) & 0x7F; },
      [&](int q) { return ((pb[q] >> 24) & 0x7F) | (pb[q] & (int)0x80000000); }, nfinal, [](int, int) {}, sh, nullptr, nullptr, nullptr, nullptr);
}

// ---- fill ----
constexpr int kTR = kLeanTileRounds;  // rounds a tile
constexpr int kWinL = kTR + 2;        // positions a tile can hold, per lattice, and one behind them
constexpr int kMaxTiles = kLeanTilesPerGroup;

// Everything that depends on the lanes per lattice is a compile-time constant (one instantiation per lane count: the
// first version, with the lane count in registers, wanted 217 VGPRs and spent its time in scratch memory).
#ifndef KLAT_FILL_LEAN_Q
#define KLAT_FILL_LEAN_Q 2  // quads of arcs a thread has in flight
#endif
#ifndef KLAT_FILL_LEAN_WAVES
#define KLAT_FILL_LEAN_WAVES 4
#endif
// what a workgroup of the fill pass keeps in LDS
template <int LG, bool FUSED>
struct FillLds {
  static constexpr int LPW = 64 >> LG;
  __attribute__((aligned(16))) uint32_t tw[kTR * 64];
  __attribute__((aligned(16))) float tg[kTR * 64];
  __attribute__((aligned(16))) float ta[FUSED ? 4 : kTR * 64];
  __attribute__((aligned(16))) int16_t tl[FUSED ? kTR * 64 : 8];  // the residuals
  int32_t win[LPW * kWinL];
  int32_t tp[LPW * (kMaxTiles + 1)];
  int64_t s0[LPW];
  int q0[LPW], S[LPW], lat[LPW], span[LPW];
  int finals;  // some lattice's forward position S (its final weights) has a round in the tile
};

// one entry: tiles [tile_begin, tile_begin + tile_count) of a job
template <int LG, bool FUSED, bool FWD>
__device__ __forceinline__ void fill_lean_run(const PackDevArgs& a, const LeanEntry& e, FillLds<LG, FUSED>& lds) {
  constexpr int G = 1 << LG, LPW = 64 >> LG, GMASK = G - 1;
  constexpr int LG_TPS = LG + 2, TPS = 1 << LG_TPS;  // threads a lattice has to itself: kTL / LPW
  constexpr int kWinPerThread = (LPW * kWinL + kTL - 1) / kTL;
  uint32_t* const tw = lds.tw;
  float* const tg = lds.tg;
  float* const ta = lds.ta;
  int16_t* const tl = lds.tl;
  int32_t* const win = lds.win;
  int32_t* const sh_tp = lds.tp;
  int64_t* const sh_s0 = lds.s0;
  int* const sh_q0 = lds.q0;
  int* const sh_S = lds.S;
  int* const sh_lat = lds.lat;
  int* const sh_span = lds.span;
  int& sh_finals = lds.finals;
  const int W = job_ring(e.tag), AL = job_stored_lanes(e.tag);
  const int tid = threadIdx.x;
  constexpr bool fwd = FWD;
  const int row_units = AL * LPW;
  const uint32_t cell_mask = (uint32_t)(W - 1);
  const int32_t* rtab = fwd ? a.r0f : a.r0b;
  int nfar = 0;
#ifdef KLAT_FILL_PROFILE  // shader cycles of thread 0 per phase, summed over the workgroups (a.prof[8 ..], KLAT_PACK_PROFILE=1)
  unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter();
#define STAMP(phase)                                                \
  do {                                                              \
    const unsigned long long now = __builtin_readcyclecounter();   \
    pc[phase] += now - pt;                                          \
    pt = now;                                                       \
  } while (0)
#else
#define STAMP(phase)
#endif
  // the job's lattices; for each the position that holds the first round of every tile of the run
  if (tid < LPW) {
    const LaneInfo li = a.lanes[e.lane_tab + tid];
    int S = 0, span = 0, q0 = 0;
    int64_t s0 = 0;
    if (li.lattice >= 0) {
      S = li.num_states;
      s0 = li.state_base;
      span = a.lat_rounds[li.lattice].max_span;
      q0 = (int)a.arc_off[li.lattice];  // (a batch has fewer than 2^31 arcs)
    }
    sh_lat[tid] = li.lattice, sh_S[tid] = S, sh_span[tid] = span, sh_s0[tid] = s0, sh_q0[tid] = q0;
  }
  __syncthreads();
  for (int i = tid; i < LPW * (e.tile_count + 1); i += kTL) {
    const int k = i / (e.tile_count + 1), t = i - k * (e.tile_count + 1);
    const int u = sh_lat[k];
    int p = 0;
    if (u >= 0) {
      const int S = sh_S[k], P = fwd ? S + 1 : S;
      const int64_t sb = sh_s0[k] + 2 * (int64_t)u;
      const int total = rtab[sb + P];
      const int m = (e.tile_begin + t) * (kTR / kPackTileRounds);  // (the tables hold a position per kPackTileRounds rounds)
      p = m * kPackTileRounds < total ? (fwd ? a.tpf : a.tpb)[tile_row(sh_s0[k], u) + m] : P;
    }
    sh_tp[k * (kMaxTiles + 1) + t] = p;
  }
  __syncthreads();
  // window of tile t: thread i holds entry o = i % kWinL of lattice k = i / kWinL (the first round of position plo + o);
  // every thread also the tile's arcs of the lattice it works for (its first arc, the end of its arcs)
  const int myk = tid >> LG_TPS, gl = tid & (TPS - 1);
  const int myS = sh_S[myk], myP = fwd ? myS + 1 : myS, my_a0 = sh_q0[myk];
  const int32_t* my_row = a.row + (sh_s0[myk] + 2 * (int64_t)max(sh_lat[myk], 0));
  const bool my_lat = sh_lat[myk] >= 0;
  int wr[kWinPerThread], nalo = 0, nahi = 0;
  auto ask_window = [&](int t) {
#pragma unroll
    for (int x = 0; x < kWinPerThread; ++x) {
      const int i = tid + x * kTL;
      wr[x] = kM;
      if (i >= LPW * kWinL) continue;
      const int k = i / kWinL, o = i - k * kWinL;
      const int u = sh_lat[k];
      if (u < 0) continue;
      const int P = fwd ? sh_S[k] + 1 : sh_S[k];
      const int plo = sh_tp[k * (kMaxTiles + 1) + t];
      if (plo + o <= P) wr[x] = (rtab + (sh_s0[k] + 2 * (int64_t)u))[plo + o];
    }
    nalo = nahi = 0;
    const int plo = sh_tp[myk * (kMaxTiles + 1) + t];
    if (my_lat && plo < myP) {
      const int phi = min(sh_tp[myk * (kMaxTiles + 1) + t + 1], myP - 1);
      // forward: the in-arcs of positions plo .. phi leave the states plo - span .. phi - 1; backward: positions
      // plo .. phi are the states S - 1 - phi .. S - 1 - plo
      nalo = my_row[fwd ? max(plo - sh_span[myk], 0) : myS - 1 - phi];
      nahi = my_row[fwd ? min(phi, myS) : myS - plo];
    }
  };
  // the arcs of a tile, four consecutive ones a lane and load: kQ quads a thread in flight
  constexpr int kQ = KLAT_FILL_LEAN_Q;
  const int32_t* ends = fwd ? a.dst : a.src;
  const int last_quad = a.num_arcs > 0 ? (int)((a.num_arcs - 1) >> 2) : 0;
  int4 ve[kQ];
  uint4 vm[kQ];
  float4 vg[kQ], va[kQ];
  auto ask_arcs = [&](int qb, int qlo, int qhi) {
#pragma unroll
    for (int x = 0; x < kQ; ++x) {
      // (a lane without a quad reads the lattice's first again; never past the batch's last quad: a tile without arcs at
      // the end of the batch's last lattice starts AT the arrays' end, and borrowed arrays end there)
      const int q = min(max(min(qb + x * TPS, qhi - 1), qlo), last_quad);
      ve[x] = reinterpret_cast<const int4*>(ends)[q];
      vm[x] = reinterpret_cast<const uint4*>(a.meta)[q];
      vg[x] = reinterpret_cast<const float4*>(a.graph)[q];
      va[x] = reinterpret_cast<const float4*>(a.acoustic)[q];
    }
  };
  // a barrier for what the workgroup keeps in LDS (__syncthreads would also wait for the loads in flight: the next tile's)
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  ask_window(0);
  STAMP(0);
  for (int t = 0; t < e.tile_count; ++t) {
    const int R0 = (e.tile_begin + t) * kTR;
    const int tile_units = (min(kTR, e.rounds - R0) >> 2) * row_units;  // 16-byte units of the tile (the job's last may be a short one)
    // ---- the window into LDS; padding everywhere (zero-weight terms that park their unused result in the lattice's
    // dummy cell behind the ring)
#pragma unroll
    for (int x = 0; x < kWinPerThread; ++x) {
      const int i = tid + x * kTL;
      if (i >= LPW * kWinL) continue;
      win[i] = wr[x];
    }
    const int alo = nalo, ahi = nahi;
    if (tid == 0) sh_finals = 0;  // (the threads below are in the same wave: they set it after this)
    if (fwd && tid < LPW && sh_lat[tid] >= 0 && sh_tp[tid * (kMaxTiles + 1) + t] <= sh_S[tid] && sh_tp[tid * (kMaxTiles + 1) + t + 1] >= sh_S[tid]) sh_finals = 1;
    for (int i = tid; i < tile_units; i += kTL) {  // one 16-byte unit (4 rounds of a lane) per store
      const int slot = i & (LPW - 1);  // unit = lane * LPW + slot, a row is a whole number of lanes
      const uint32_t w = kNil | ((uint32_t)(W * LPW + slot) << kWrShift);
      reinterpret_cast<uint4*>(tw)[i] = make_uint4(w, w, w, w);
      reinterpret_cast<float4*>(tg)[i] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
      if (FUSED)
        reinterpret_cast<int2*>(tl)[i] = make_int2(0, 0);
      else
        reinterpret_cast<float4*>(ta)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    STAMP(1);
    lds_barrier();
    STAMP(2);
    // slot of (round rr of the tile, lane sub, lattice k)
    auto slot_of = [&](int k, int dr, int sub) {
      int unit;  // (the chunk's row + the lane's units + the lattice: one 24-bit multiply-add, which hipcc would not pick)
      asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(unit) : "v"(dr >> 2), "s"(row_units), "v"(sub * LPW + k));
      return (unit << 2) + (dr & 3);
    };
    auto put_cost = [&](int i, uint32_t w, float g, float ac) {
      if (FUSED) {
        // the cost as the reference forms it (ScaleLattice, AddWordInsPen, ConvertToCost), kept as a float and
        // a 16-bit residual in units of the batch's lo_unit
        const double c = arc_cost(g, ac, (w & kWord) != 0, a.fsp);
        const float hi = (float)c;
        int n = 0;
        if (c < 1.0e30 && c > -1.0e30) n = max(-32767, min(32767, (int)rint((c - (double)hi) * a.lo_inv_unit)));
        tg[i] = hi;
        tl[i] = (int16_t)n;
      } else {
        tg[i] = g;
        ta[i] = ac;
      }
    };
    // the flags and the write cell of term j of a position (first round r, nr rounds; own_cell its ring cell, dummy the lattice's dummy cell)
    auto finish_word = [&](uint32_t w, int j, int jq, int nr, uint32_t own_cell, uint32_t dummy, bool far) {
      if (!far) w |= dummy << kWrShift;
      if (jq == 0) w |= kFirst;
      if (jq == nr - 1) {
        w |= kLast;
        if ((j & GMASK) == 0 && !far) w = (w & ~(kCellMask << kWrShift)) | kCommit | (own_cell << kWrShift);
      }
      return w;
    };
    // ---- the arcs: the lattice in slot k has the threads [k * TPS, (k + 1) * TPS) to itself
    {
      const int k = myk, S = myS, P = myP, a0 = my_a0;
      const int plo = sh_tp[k * (kMaxTiles + 1) + t];
      const int n = min(sh_tp[k * (kMaxTiles + 1) + t + 1], P - 1) - plo + 1;  // positions that may have a round here
      const int32_t* wk = win + k * kWinL;
      const uint32_t dummy = (uint32_t)(W * LPW + k);
      const int qlo = (a0 + alo) >> 2, qhi = (a0 + ahi + 3) >> 2;
      ask_arcs(qlo + gl, qlo, qhi);
      if (t + 1 < e.tile_count) ask_window(t + 1);  // (behind the arcs' loads: it lands while they are worked on)
      for (int qb = qlo + gl; qb < qhi; qb += kQ * TPS) {
        if (qb != qlo + gl) ask_arcs(qb, qlo, qhi);
#pragma unroll
        for (int x = 0; x < kQ; ++x) {
          const int q = qb + x * TPS;
          const int ib = q < qhi ? 4 * q - a0 - alo : ahi;  // (a quad behind the lattice's last: no element passes the test below)
#pragma unroll
          for (int el = 0; el < 4; ++el) {
            // everything without a branch up to the three stores: the tests cost less as one mask than as four jumps
            const uint32_t m = elem(vm[x], el);
            const int rel = (int)(m & 0xFFu);
            const int p = fwd ? elem(ve[x], el) : S - 1 - elem(ve[x], el);
            const unsigned o = (unsigned)(p - plo);
            bool ok = (unsigned)(ib + el) < (unsigned)(ahi - alo) && o < (unsigned)n;
            const unsigned oc = min(o, (unsigned)kTR);  // (a window holds kTR + 2 entries)
            const int w0 = wk[oc], w1 = wk[oc + 1];
            const int r = w0 & (int)kRoundMask, nr = (w1 & (int)kRoundMask) - r;
            // the term's number within its position: behind the constant-one term of a final state (backward)
            const int j = fwd ? (int)((m >> 8) & 0xFFu) : (int)((m >> 16) & 0xFFu) + (int)((uint32_t)w0 >> 31);
            const int jq = j >> LG, dr = r + jq - R0;
            ok = ok && (unsigned)dr < (unsigned)kTR;
            const uint32_t own12 = ((((uint32_t)p & cell_mask) << (6 - LG)) + (uint32_t)k) << kWrShift;
            const bool far = rel >= W, last = jq == nr - 1;
            uint32_t w = (((uint32_t)(p - rel) & cell_mask) << (6 - LG)) + (uint32_t)k;
            w |= (rel == 1 && jq == 0) ? kBypass : 0u;
            w |= (last && (j & GMASK) == 0) ? (kCommit | own12) : (dummy << kWrShift);
            w = far ? (kFar | (uint32_t)rel) : w;
            w |= ((m & (1u << 24)) << 6) | (jq == 0 ? kFirst : 0u) | (last ? kLast : 0u);
            if (ok) {
              nfar += far ? 1 : 0;
              const int at = slot_of(k, dr, j & GMASK);
              tw[at] = w;
              put_cost(at, w, elem(vg[x], el), elem(va[x], el));
            }
          }
        }
      }
    }
    STAMP(3);
    // ---- one thread per position: what is not an arc
#pragma unroll 1
    for (int i = tid; i < LPW * (kTR + 1); i += kTL) {
      const int k = i / (kTR + 1), o = i - k * (kTR + 1);
      if (sh_lat[k] < 0) continue;
      const int S = sh_S[k], P = fwd ? S + 1 : S;
      const int p = sh_tp[k * (kMaxTiles + 1) + t] + o;
      if (p >= P || p > sh_tp[k * (kMaxTiles + 1) + t + 1]) continue;
      const int w0 = win[k * kWinL + o];
      const int r = w0 & (int)kRoundMask, nr = (win[k * kWinL + o + 1] & (int)kRoundMask) - r;
      if (r >= R0 + kTR || r + nr <= R0) continue;
      const int jlo = max(0, (R0 - r) << LG), jhi = min(nr << LG, (R0 + kTR - r) << LG);
      const int nterm = (w0 >> 24) & 0x7F;
      const bool lead = fwd ? p == 0 : ((uint32_t)w0 >> 31) != 0;  // alpha[start] = 0; a final state's weight
      const uint32_t own_cell = (((uint32_t)p & cell_mask) << (6 - LG)) + (uint32_t)k, dummy = (uint32_t)(W * LPW + k);
      if (jlo == 0 && (lead || nterm == 0)) {
        float g = 0.0f, ac = 0.0f;
        if (lead && !fwd) g = a.final_graph[sh_s0[k] + S - 1 - p], ac = a.final_acoustic[sh_s0[k] + S - 1 - p];
        const uint32_t w = finish_word((lead ? kOne : kNil) | own_cell, 0, 0, nr, own_cell, dummy, false);
        const int at = slot_of(k, r - R0, 0);
        tw[at] = w;
        if (lead) put_cost(at, w, g, ac);  // (a NIL term keeps the padding's infinite cost)
      }
      // lanes of the position's rounds that carry no term (cost: the padding's)
#pragma unroll 1
      for (int jj = max(max(nterm, 1), jlo); jj < jhi; ++jj) {
        if ((jj & GMASK) >= AL) continue;
        const int jq = jj >> LG;
        uint32_t w = kNil | (dummy << kWrShift);
        if (jq == 0) w |= kFirst;
        if (jq == nr - 1) w |= kLast;
        tw[slot_of(k, r + jq - R0, jj & GMASK)] = w;
      }
    }
    STAMP(4);
    // ---- forward position S: one term per final state (up to 127 of them), a thread each
    if (fwd && sh_finals) {
#pragma unroll 1
      for (int i = tid; i < LPW * 128; i += kTL) {
        const int k = i >> 7, jj = i & 127;
        const int S = sh_S[k], plo = sh_tp[k * (kMaxTiles + 1) + t];
        if (sh_lat[k] < 0 || plo > S || sh_tp[k * (kMaxTiles + 1) + t + 1] < S) continue;  // (position S has no round here)
        const int w0 = win[k * kWinL + S - plo];
        const int r = w0 & (int)kRoundMask, nr = (win[k * kWinL + S - plo + 1] & (int)kRoundMask) - r;
        const int jq = jj >> LG, dr = r + jq - R0;
        if (jj >= ((w0 >> 24) & 0x7F) || (unsigned)dr >= (unsigned)kTR) continue;
        const int s = a.flist[sh_s0[k] + jj];
        const int rel = S - s;
        const bool far = rel >= W;
        const uint32_t own_cell = (((uint32_t)S & cell_mask) << (6 - LG)) + (uint32_t)k, dummy = (uint32_t)(W * LPW + k);
        uint32_t w = far ? kFar | (uint32_t)rel : (((uint32_t)(S - rel) & cell_mask) << (6 - LG)) + (uint32_t)k;
        if (!far && rel == 1 && jq == 0) w |= kBypass;
        nfar += far ? 1 : 0;
        w = finish_word(w, jj, jq, nr, own_cell, dummy, far);
        const int at = slot_of(k, dr, jj & GMASK);
        tw[at] = w;
        put_cost(at, w, a.final_graph[sh_s0[k] + s], a.final_acoustic[sh_s0[k] + s]);
      }
    }
    lds_barrier();
    STAMP(5);
    // ---- the tile leaves as whole lines
    {
      const int64_t base = (e.chunk_off + (int64_t)(R0 >> 2) * row_units) << 2;  // first slot of the tile
      uint4* gw = reinterpret_cast<uint4*>(a.sw + base);
      float4* gg = reinterpret_cast<float4*>(a.sg + base);
      const uint4* lw = reinterpret_cast<const uint4*>(tw);
      const float4* lg4 = reinterpret_cast<const float4*>(tg);
      if (FUSED) {  // the residuals: four 16-bit words per unit
        int2* gl = reinterpret_cast<int2*>(a.sl + base);
        const int2* ll = reinterpret_cast<const int2*>(tl);
        for (int i = tid; i < tile_units; i += kTL) {
          gw[i] = lw[i];
          gg[i] = lg4[i];
          gl[i] = ll[i];
        }
      } else {
        float4* ga = reinterpret_cast<float4*>(a.sa + base);
        const float4* la = reinterpret_cast<const float4*>(ta);
        for (int i = tid; i < tile_units; i += kTL) {
          gw[i] = lw[i];
          gg[i] = lg4[i];
          ga[i] = la[i];
        }
      }
    }
    STAMP(6);
    lds_barrier();
    STAMP(7);
  }
#ifdef KLAT_FILL_PROFILE
  if (a.prof && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[(fwd ? 8 : 16) + i], pc[i]);
#endif
#undef STAMP
  for (int d = 32; d > 0; d >>= 1) nfar += __shfl_xor(nfar, d, 64);
  if ((tid & 63) == 0 && nfar) atomicAdd(&a.status[3], nfar);
}

// DIRS: 0 the entries are forward jobs', 1 backward jobs', 2 either (the entry says: a workgroup-uniform branch; the two
// directions of one group of lattices dispatched next to each other read the same arcs within microseconds of each other)
template <int LG, bool FUSED, int DIRS>
__global__ __launch_bounds__(kTL, KLAT_FILL_LEAN_WAVES) void pack_fill_lean_kernel(PackDevArgs a, int entry0) {
  __shared__ FillLds<LG, FUSED> lds;
  const LeanEntry e = a.lentries[entry0 + blockIdx.x];
  if (DIRS == 0 || (DIRS == 2 && e.dir == 0))
    fill_lean_run<LG, FUSED, true>(a, e, lds);
  else
    fill_lean_run<LG, FUSED, false>(a, e, lds);
}

}  // namespace

hipError_t launch_pack_count_lean(const PackDevArgs& a, int num_lattices, const int class_need[kPackLeanClasses], hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_lean_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_lean_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_lean_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
    attr_done = true;
  }
  for (int c = 0; c < kPackLeanClasses; ++c) {
    if (class_need[c] <= 0) continue;
    const int lo = c == 0 ? -1 : kPackLeanClassLds[c - 1], hi = kPackLeanClassLds[c];
    if (c == 0)
      hipLaunchKernelGGL(pack_count_lean_kernel<256>, dim3(num_lattices), dim3(256), (size_t)class_need[c], stream, a, lo, hi);
    else if (c == 1)
      hipLaunchKernelGGL(pack_count_lean_kernel<512>, dim3(num_lattices), dim3(512), (size_t)class_need[c], stream, a, lo, hi);
    else
      hipLaunchKernelGGL(pack_count_lean_kernel<1024>, dim3(num_lattices), dim3(1024), (size_t)class_need[c], stream, a, lo, hi);
  }
  return hipGetLastError();
}

hipError_t launch_pack_rounds_lean(const PackDevArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rounds_lean_kernel, dim3(num_lattices), dim3(kTL), 0, stream, a);
  return hipGetLastError();
}

namespace {
template <int LG>
void launch_fill_lean_lg(const PackDevArgs& a, int dirs, int entry0, int count, hipStream_t stream) {
  if (count <= 0) return;
#define KLAT_FILL_LAUNCH(FUSED, DIRS) hipLaunchKernelGGL((pack_fill_lean_kernel<LG, FUSED, DIRS>), dim3(count), dim3(kTL), 0, stream, a, entry0)
  if (a.fused) {
    if (dirs == 0) KLAT_FILL_LAUNCH(true, 0);
    else if (dirs == 1) KLAT_FILL_LAUNCH(true, 1);
    else KLAT_FILL_LAUNCH(true, 2);
  } else {
    if (dirs == 0) KLAT_FILL_LAUNCH(false, 0);
    else if (dirs == 1) KLAT_FILL_LAUNCH(false, 1);
    else KLAT_FILL_LAUNCH(false, 2);
  }
#undef KLAT_FILL_LAUNCH
}
}  // namespace

// a.lentries sorted by lanes per lattice; within a lane count either by direction (count[3 lg + 0 / 1]: one launch each)
// or the two directions of a group of lattices next to each other (count[3 lg + 2]: one launch)
hipError_t launch_pack_fill_lean(const PackDevArgs& a, const int count[3 * kNumLaneChoices], hipStream_t stream) {
  int at = 0;
  for (int c = 0; c < 3 * kNumLaneChoices; ++c) {
    const int dirs = c % 3;
    switch (c / 3) {
      case 0: launch_fill_lean_lg<0>(a, dirs, at, count[c], stream); break;
      case 1: launch_fill_lean_lg<1>(a, dirs, at, count[c], stream); break;
      case 2: launch_fill_lean_lg<2>(a, dirs, at, count[c], stream); break;
      case 3: launch_fill_lean_lg<3>(a, dirs, at, count[c], stream); break;
      case 4: launch_fill_lean_lg<4>(a, dirs, at, count[c], stream); break;
      case 5: launch_fill_lean_lg<5>(a, dirs, at, count[c], stream); break;
      default: launch_fill_lean_lg<6>(a, dirs, at, count[c], stream); break;
    }
    at += std::max(count[c], 0);
  }
  return hipGetLastError();
}

}  // namespace klat
