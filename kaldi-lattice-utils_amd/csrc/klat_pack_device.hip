// klat_pack_device.hip -- the lockstep arc streams built on the device from the CSR batch already
// in HBM.  Same streams, bit for bit, as the host packer (klat_pack.cc), which stays as the
// reference of the stream format for the CPU-side tests; klat_batch_create uses this one.
//
//   pack_count_kernel  one workgroup per lattice, the lattice's tables in LDS: validates the
//                      topology, counts in- and out-degrees (LDS atomics), sorts the arcs by
//                      destination (counting sort; sources ascending inside a destination), counts
//                      the rounds its forward and its backward stream would take with 1, 2 ... 64
//                      lanes, and leaves three tables in HBM as full lines: first in-arc per state,
//                      first out-arc per state, arcs by destination.  The round counts go back to
//                      the host, which picks every lattice's lanes and slices the lattices into jobs
//                      (plan_layout, build_jobs in klat_pack.cc).  A lattice whose tables exceed the
//                      LDS of a CU takes pack_count_global_kernel (same results, tables in L2).
//   pack_rounds_kernel one workgroup per lattice: prefix sums of the rounds every state takes in
//                      the forward and in the backward stream with the lanes the plan gave it.
//   pack_fill_kernel   one workgroup per (job, run of tiles).  A tile is 32 rounds of the job's
//                      stored lanes: one contiguous piece of each of the three stream arrays.  The
//                      workgroup assembles the tile in LDS -- every state that has a round in the
//                      tile writes its term words and costs (ring cells, BYPASS / ONE / FAR / WORD,
//                      FIRST / LAST / COMMIT) and the padding of its unused lanes -- and stores it
//                      as whole lines (dwordx4 per lane).  Written term by term from registers the
//                      same streams cost 5.4 GB of HBM writes for 2 GB of stream on C2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_pack_device.h"

namespace klat {

namespace {

constexpr int kT = 256;       // pack_rounds_kernel, pack_fill_kernel, pack_count_global_kernel
constexpr int kTC = 512;      // pack_count_kernel

__device__ __forceinline__ bool dev_is_final(const PackDevArgs& a, int64_t gs) {
  return !(a.final_graph[gs] == INFINITY && a.final_acoustic[gs] == INFINITY);
}

// exclusive prefix sum of one int per thread; returns the thread's prefix, *total = block sum
template <int T>
__device__ int block_scan(int v, int* total, int* sh /* [T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < T / 64; ++w) {
    if (w < wave) off += sh[w];
    tot += sh[w];
  }
  *total = tot;
  return off + incl - v;
}

__device__ __forceinline__ int lower_bound_src(const int32_t* src, int n, int s) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (src[mid] < s)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void report_bad(const PackDevArgs& a, int code, int u) {
  if (atomicCAS(&a.status[0], 0, code) == 0) a.status[1] = u;
}

// The lattice's tables in LDS; arc numbers as 16-bit words (a lattice of 65 536 arcs or more takes
// the global kernel: pack_count_lds_bytes puts it above every class).
typedef uint16_t IA;
__global__ __launch_bounds__(kTC) void pack_count_kernel(PackDevArgs a, int lds_lo, int lds_hi) {
  extern __shared__ int32_t lds[];
  __shared__ int sh[kTC / 64];
  __shared__ int sh_bad, sh_mr, sh_mtf, sh_mtb;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  const int need = pack_count_lds_bytes(S, A);
  // every lattice belongs to exactly one launch: the narrowest LDS class it fits
  if (need <= lds_lo || need > lds_hi) return;
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  if (S <= 0) {
    if (tid == 0 && (S < 0 || A != 0)) report_bad(a, 1 /* KLAT_ERR_INVALID_ARG */, u);
    return;
  }
  int32_t* cnt = lds;              // [S + 2] in-degrees, then their prefix
  int32_t* row = lds + (S + 2);    // [S + 2] out-degrees, then their prefix
  IA* in_arc = reinterpret_cast<IA*>(lds + 2 * (S + 2));  // [A]
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  if (tid == 0) sh_bad = 0, sh_mr = 1, sh_mtf = 1, sh_mtb = 1;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < 2 * (S + 2); s += kTC) lds[s] = 0;
  __syncthreads();
  // degrees + validation (the host packer's checks: ids in range, arcs grouped by source, every
  // arc goes to a higher state)
  int mr = 1;
  for (int i = tid; i < A; i += kTC) {
    const int s = src[i], t = dst[i];
    const int prev = i > 0 ? src[i - 1] : 0;
    if (s < 0 || s >= S || s < prev) {
      atomicMax(&sh_bad, 2);  // malformed CSR
    } else if (t <= s || t >= S) {
      atomicMax(&sh_bad, 1);  // not topologically sorted
    } else {
      atomicAdd(&cnt[t], 1);
      atomicAdd(&row[s], 1);
      mr = max(mr, t - s);
    }
  }
  for (int s = tid; s < S; s += kTC)
    if (dev_is_final(a, s0 + s)) mr = max(mr, S - s);
  atomicMax(&sh_mr, mr);
  __syncthreads();
  if (sh_bad) {
    if (tid == 0) report_bad(a, sh_bad == 2 ? 1 /* INVALID_ARG */ : 2 /* BAD_TOPOLOGY */, u);
    return;
  }
  if (tid == 0) atomicMax(&a.status[2], sh_mr);
  // rank of every final state among the finals (terms of the forward stream's last position),
  // backward rounds (position p = S-1-s holds the out-arcs of s + its final weight), row table
  int fsum[kNumLaneChoices] = {0}, bsum[kNumLaneChoices] = {0};
  int carry_f = 0, carry_o = 0, mtf = 1, mtb = 1;
  for (int base = 0; base <= S; base += kTC) {
    const int s = base + tid;
    const int f = (s < S && dev_is_final(a, s0 + s)) ? 1 : 0;
    const int od = s < S ? row[s] : 0;
    int tot_f, tot_o;
    const int ex_f = block_scan<kTC>(f, &tot_f, sh);
    const int ex_o = block_scan<kTC>(od, &tot_o, sh);
    if (s < S) {
      a.frank[s0 + s] = carry_f + ex_f;
      const int nb = od + f;
      mtb = max(mtb, nb);
      for (int k = 0; k < kNumLaneChoices; ++k) bsum[k] += (max(nb, 1) + (1 << k) - 1) >> k;
    }
    if (s <= S) row[s] = carry_o + ex_o;
    carry_f += tot_f;
    carry_o += tot_o;
  }
  const int nfinal = carry_f;
  // forward: rounds of every position p = 0..S (S = the virtual super-final state) for every
  // number of lanes; in-degrees -> exclusive prefix in place (cnt[p] = first in-arc of p, cnt[S] = A)
  int carry_c = 0;
  for (int base = 0; base <= S; base += kTC) {
    const int p = base + tid;
    const int c = (p < S) ? cnt[p] : 0;
    const int nterm = (p < S) ? c + (p == 0 ? 1 : 0) : nfinal;
    if (p <= S) {
      mtf = max(mtf, nterm);
      for (int k = 0; k < kNumLaneChoices; ++k) fsum[k] += (max(nterm, 1) + (1 << k) - 1) >> k;
    }
    int tot_c;
    const int ex_c = block_scan<kTC>(c, &tot_c, sh);
    if (p <= S) {
      cnt[p] = carry_c + ex_c;
      a.cnt[sb + p] = carry_c + ex_c;
      a.row[sb + p] = row[p];
    }
    carry_c += tot_c;
  }
  if (tid == 0) a.cnt[sb + S + 1] = nfinal;  // (the spare entry of the row)
  __syncthreads();
  // stable counting sort of the arcs by destination: scatter (cnt[t] moves from the first entry of
  // bucket t to the first entry of bucket t + 1), then order every bucket by arc number
  for (int i = tid; i < A; i += kTC) in_arc[atomicAdd(&cnt[dst[i]], 1)] = (IA)i;
  __syncthreads();
  for (int s = tid; s < S; s += kTC) {
    const int b = s > 0 ? cnt[s - 1] : 0, e = cnt[s];
    for (int i = b + 1; i < e; ++i) {
      const IA v = in_arc[i];
      int j = i - 1;
      while (j >= b && in_arc[j] > v) {
        in_arc[j + 1] = in_arc[j];
        --j;
      }
      in_arc[j + 1] = v;
    }
  }
  __syncthreads();
  for (int i = tid; i < A; i += kTC) a.in_arc[a0 + i] = (int32_t)in_arc[i];
  atomicMax(&sh_mtf, mtf);
  atomicMax(&sh_mtb, mtb);
  for (int k = 0; k < kNumLaneChoices; ++k) {
    if (fsum[k]) atomicAdd(&sh_rounds[k], fsum[k]);
    if (bsum[k]) atomicAdd(&sh_rounds[kNumLaneChoices + k], bsum[k]);
  }
  __syncthreads();
  if (tid < kNumLaneChoices) {
    a.lat_rounds[u].fwd[tid] = sh_rounds[tid];
    a.lat_rounds[u].bwd[tid] = sh_rounds[kNumLaneChoices + tid];
  }
  if (tid == 0) {
    a.lat_rounds[u].max_rel = sh_mr;
    a.lat_rounds[u].max_terms = min(sh_mtf, 64) | (min(sh_mtb, 64) << 16);
  }
}

// The same pass with the lattice's tables in its scratch rows in HBM (L2 atomics): for the lattices
// above `lds_hi` bytes of tables.
__global__ __launch_bounds__(kT) void pack_count_global_kernel(PackDevArgs a, int lds_hi) {
  __shared__ int sh[kT / 64];
  __shared__ int sh_bad, sh_mr, sh_mtf, sh_mtb;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  if (pack_count_lds_bytes(S, A) <= lds_hi) return;
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  const int64_t sb = s0 + 2 * (int64_t)u;
  int32_t* cnt = a.cnt + sb;
  int32_t* row = a.row + sb;
  int32_t* cur = a.r0b + sb;  // (scratch for the counting sort's cursors here)
  int32_t* frank = a.frank + s0;
  int32_t* in_arc = a.in_arc + a0;
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  if (tid == 0) sh_bad = 0, sh_mr = 1, sh_mtf = 1, sh_mtb = 1;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < S + 2; s += kT) cnt[s] = 0;
  __syncthreads();
  int mr = 1;
  for (int i = tid; i < A; i += kT) {
    const int s = src[i], t = dst[i];
    const int prev = i > 0 ? src[i - 1] : 0;
    if (s < 0 || s >= S || s < prev) {
      atomicMax(&sh_bad, 2);
    } else if (t <= s || t >= S) {
      atomicMax(&sh_bad, 1);
    } else {
      atomicAdd(&cnt[t], 1);
      mr = max(mr, t - s);
    }
  }
  for (int s = tid; s < S; s += kT)
    if (dev_is_final(a, s0 + s)) mr = max(mr, S - s);
  atomicMax(&sh_mr, mr);
  __syncthreads();
  if (sh_bad) {
    if (tid == 0) report_bad(a, sh_bad == 2 ? 1 : 2, u);
    return;
  }
  if (tid == 0) atomicMax(&a.status[2], sh_mr);
  int carry = 0;
  for (int base = 0; base < S; base += kT) {
    const int s = base + tid;
    const int f = (s < S && dev_is_final(a, s0 + s)) ? 1 : 0;
    int tot;
    const int ex = block_scan<kT>(f, &tot, sh);
    if (s < S) frank[s] = carry + ex;
    carry += tot;
  }
  const int nfinal = carry;
  int fsum[kNumLaneChoices] = {0}, bsum[kNumLaneChoices] = {0};
  int carry_c = 0, mtf = 1, mtb = 1;
  for (int base = 0; base <= S; base += kT) {
    const int p = base + tid;
    const int c = (p < S) ? cnt[p] : 0;
    const int nterm = (p < S) ? c + (p == 0 ? 1 : 0) : nfinal;
    if (p <= S) {
      mtf = max(mtf, nterm);
      for (int k = 0; k < kNumLaneChoices; ++k) fsum[k] += (max(nterm, 1) + (1 << k) - 1) >> k;
    }
    int tot_c;
    const int ex_c = block_scan<kT>(c, &tot_c, sh);
    if (p <= S) cnt[p] = carry_c + ex_c;
    carry_c += tot_c;
  }
  if (tid == 0) cnt[S + 1] = nfinal;
  __syncthreads();
  for (int s = tid; s < S; s += kT) cur[s] = cnt[s];
  for (int s = tid; s <= S; s += kT) row[s] = lower_bound_src(src, A, s);
  __syncthreads();
  for (int i = tid; i < A; i += kT) in_arc[atomicAdd(&cur[dst[i]], 1)] = i;
  __syncthreads();
  for (int s = tid; s < S; s += kT) {
    const int b = cnt[s], e = cnt[s + 1];
    for (int i = b + 1; i < e; ++i) {
      const int v = in_arc[i];
      int j = i - 1;
      while (j >= b && in_arc[j] > v) {
        in_arc[j + 1] = in_arc[j];
        --j;
      }
      in_arc[j + 1] = v;
    }
    const int nb = row[s + 1] - row[s] + (dev_is_final(a, s0 + s) ? 1 : 0);
    mtb = max(mtb, nb);
    for (int k = 0; k < kNumLaneChoices; ++k) bsum[k] += (max(nb, 1) + (1 << k) - 1) >> k;
  }
  atomicMax(&sh_mtf, mtf);
  atomicMax(&sh_mtb, mtb);
  for (int k = 0; k < kNumLaneChoices; ++k) {
    if (fsum[k]) atomicAdd(&sh_rounds[k], fsum[k]);
    if (bsum[k]) atomicAdd(&sh_rounds[kNumLaneChoices + k], bsum[k]);
  }
  __syncthreads();
  if (tid < kNumLaneChoices) {
    a.lat_rounds[u].fwd[tid] = sh_rounds[tid];
    a.lat_rounds[u].bwd[tid] = sh_rounds[kNumLaneChoices + tid];
  }
  if (tid == 0) {
    a.lat_rounds[u].max_rel = sh_mr;
    a.lat_rounds[u].max_terms = min(sh_mtf, 64) | (min(sh_mtb, 64) << 16);
  }
}

__global__ __launch_bounds__(kT) void pack_rounds_kernel(PackDevArgs a) {
  __shared__ int sh[kT / 64];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  if (S <= 0 || a.lat_g[u] < 0) return;
  const int k = a.lat_g[u];
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* cnt = a.cnt + sb;
  const int32_t* row = a.row + sb;
  int32_t* r0f = a.r0f + sb;
  int32_t* r0b = a.r0b + sb;
  const int nfinal = cnt[S + 1];
  int carry_r = 0, carry_b = 0;
  for (int base = 0; base <= S; base += kT) {
    const int p = base + tid;
    const int nterm = (p < S) ? cnt[p + 1] - cnt[p] + (p == 0 ? 1 : 0) : nfinal;
    const int nr = (p <= S) ? (max(nterm, 1) + (1 << k) - 1) >> k : 0;
    int nrb = 0;
    if (p < S) {
      const int s = S - 1 - p;
      const int nb = row[s + 1] - row[s] + (dev_is_final(a, s0 + s) ? 1 : 0);
      nrb = (max(nb, 1) + (1 << k) - 1) >> k;
    }
    int tot_r, tot_b;
    const int ex_r = block_scan<kT>(nr, &tot_r, sh);
    const int ex_b = block_scan<kT>(nrb, &tot_b, sh);
    if (p <= S) r0f[p] = carry_r + ex_r;
    if (p < S) r0b[p] = carry_b + ex_b;
    carry_r += tot_r;
    carry_b += tot_b;
  }
  if (tid == 0) r0f[S + 1] = carry_r, r0b[S] = carry_b;
}

// ---- fill ----
constexpr int kTileRounds = kPackTileRounds;     // 32: 8 chunks, what a job's length is a multiple of
constexpr int kWin = kTileRounds + 2;            // round offsets of the states a tile can hold, per lattice

__global__ __launch_bounds__(kT) void pack_fill_kernel(PackDevArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t tw[kTileRounds * 64];
  __shared__ __attribute__((aligned(16))) float tg[kTileRounds * 64];
  __shared__ __attribute__((aligned(16))) float ta[kTileRounds * 64];
  __shared__ int sh_r[64 * kWin];
  __shared__ int sh_plo[64], sh_next[64], sh_S[64], sh_lat[64], sh_fin[64];
  __shared__ int sh_far;
  const PackEntry e = a.entries[blockIdx.x];
  const SweepJob job = a.jobs[e.job];
  const int G = job_lanes(job.tag), W = job_ring(job.tag), AL = job_stored_lanes(job.tag), LPW = 64 / G;
  const int tid = threadIdx.x;
  const bool fwd = job.dir == 0;
  const int row_units = AL * LPW;                      // 16-byte units of one chunk row
  const int tile_slots = kTileRounds * row_units;      // stream slots of one tile
  int nfar = 0;
  if (tid == 0) sh_far = 0;
  // the lattices of the job and, for each, the first position with a round in this run of tiles
  if (tid < LPW) {
    const LaneInfo li = a.lanes[job.lane_tab + tid];
    int plo = 0, S = 0;
    if (li.lattice >= 0) {
      S = li.num_states;
      const int64_t sb = li.state_base + 2 * (int64_t)li.lattice;
      const int32_t* r0 = (fwd ? a.r0f : a.r0b) + sb;
      const int P = fwd ? S + 1 : S;  // positions; r0[P] = the lattice's rounds
      const int R0 = e.tile_begin * kTileRounds;
      int lo = 0, hi = P;             // first p with r0[p + 1] > R0 (P if none)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (r0[mid + 1] > R0)
          hi = mid;
        else
          lo = mid + 1;
      }
      plo = lo;
    }
    sh_lat[tid] = li.lattice;
    sh_S[tid] = S;
    sh_plo[tid] = plo;
  }
  __syncthreads();
  for (int t = 0; t < e.tile_count; ++t) {
    const int R0 = (e.tile_begin + t) * kTileRounds, R1 = R0 + kTileRounds;
    // padding everywhere: zero-weight terms that park their (unused) result in the lattice's dummy
    // cell behind the ring
    for (int i = tid; i < tile_slots; i += kT) {
      const int slot = ((i >> 2) % row_units) % LPW;
      tw[i] = kNil | ((uint32_t)(W * LPW + slot) << kWrShift);
      tg[i] = INFINITY;
      ta[i] = 0.0f;
    }
    for (int i = tid; i < LPW * kWin; i += kT) {
      const int k = i / kWin, o = i - k * kWin;
      int v = 0x7FFFFFFF;
      const int u = sh_lat[k];
      if (u >= 0) {
        const int S = sh_S[k], P = fwd ? S + 1 : S;
        const int p = sh_plo[k] + o;
        if (p <= P) v = ((fwd ? a.r0f : a.r0b) + a.state_off[u] + 2 * (int64_t)u)[p];
        if (o == 0) sh_next[k] = P, sh_fin[k] = 0;
      }
      sh_r[i] = v;
    }
    __syncthreads();
    auto put = [&](int k, int rr, int sub, uint32_t w, float g, float ac) {
      const int i = ((((rr - R0) >> 2) * row_units + sub * LPW + k) << 2) + ((rr - R0) & 3);
      tw[i] = w;
      tg[i] = g;
      ta[i] = ac;
    };
    for (int i = tid; i < LPW * (kTileRounds + 1); i += kT) {
      const int k = i / (kTileRounds + 1), o = i - k * (kTileRounds + 1);
      const int u = sh_lat[k];
      if (u < 0) continue;
      const int S = sh_S[k], P = fwd ? S + 1 : S;
      const int p = sh_plo[k] + o;
      const int r = sh_r[k * kWin + o], rn = sh_r[k * kWin + o + 1];
      if (p >= P) continue;
      if (r <= R1 && rn > R1) sh_next[k] = p;  // the first position of the next tile
      if (r >= R1 || rn <= R0) continue;
      const int nr = rn - r;
      if (fwd && p == S) {  // one term per final state: by the whole workgroup, below
        sh_fin[k] = 1;
        continue;
      }
      const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
      const int64_t sb = s0 + 2 * (int64_t)u;
      const uint32_t cell_mask = (uint32_t)(W - 1);
      const uint32_t own_cell = (uint32_t)((p & cell_mask) * LPW + k);
      const uint32_t dummy = (uint32_t)(W * LPW + k) << kWrShift;
      // term j of the position sits in round r + j / G on lane j % G of the lattice
      const int jlo = max(0, (R0 - r) * G), jhi = min(nr * G, (R1 - r) * G);
      int j = 0, nterm = 0;
      auto emit = [&](int rel, bool word, float g, float ac, bool nil) {
        if (j >= jlo && j < jhi && (j % G) < AL) {
          const int rr = r + j / G;
          uint32_t w;
          bool far = false;
          if (nil) {
            w = kNil | own_cell;
          } else if (rel == 0) {
            w = kOne | own_cell;
          } else if (rel >= W) {
            w = kFar | (uint32_t)rel;
            far = true;
            nfar++;
          } else {
            w = (uint32_t)(((p - rel) & cell_mask) * LPW + k);
            if (rel == 1 && j < G) w |= kBypass;
          }
          if (word) w |= kWord;
          if (!far) w |= dummy;
          if (rr == r) w |= kFirst;
          if (rr == r + nr - 1) {
            w |= kLast;
            if (j % G == 0 && !far) w = (w & ~(kCellMask << kWrShift)) | kCommit | (own_cell << kWrShift);
          }
          put(k, rr, j % G, w, nil ? INFINITY : g, ac);
        }
        ++j;  // (every term sits on a stored lane: AL = min(G, the most terms of one position))
      };
      if (fwd) {
        const int32_t* cnt = a.cnt + sb;
        const int32_t* in_arc = a.in_arc + a0;
        const int q0 = cnt[p], q1 = cnt[p + 1];
        nterm = q1 - q0 + (p == 0 ? 1 : 0);
        if (p == 0) emit(0, false, 0.0f, 0.0f, false);  // alpha[start] = 0
        // skip the in-arcs whose rounds lie before the tile
        int q = q0;
        if (j < jlo) {
          const int skip = min(q1 - q0, jlo - j);
          q += skip;
          j += skip;
        }
        for (; q < q1 && j < jhi; ++q) {
          const int64_t i2 = a0 + in_arc[q];
          emit(p - a.src[i2], a.ilabel[i2] != 0, a.graph[i2], a.acoustic[i2], false);
        }
        if (nterm == 0) emit(0, false, 0.0f, 0.0f, true);
      } else {
        const int s = S - 1 - p;
        const int32_t* row = a.row + sb;
        const int32_t* dst = a.dst + a0;
        const int lo = row[s], hi = row[s + 1];
        const bool fin = dev_is_final(a, s0 + s);
        nterm = hi - lo + (fin ? 1 : 0);
        if (fin) emit(0, false, a.final_graph[s0 + s], a.final_acoustic[s0 + s], false);
        // farthest destination first, nearest last; arcs with the same destination in arc order
        int last_dst = 0x7FFFFFFF, last_idx = -1;
        for (int n = lo; n < hi && j < jhi; ++n) {
          int best = -1, best_dst = -1;
          for (int i2 = lo; i2 < hi; ++i2) {
            const int d = dst[i2];
            const bool after = d < last_dst || (d == last_dst && i2 > last_idx);
            if (after && (d > best_dst)) {
              best = i2;
              best_dst = d;
            }
          }
          last_dst = best_dst;
          last_idx = best;
          if (j >= jlo)
            emit(best_dst - s, a.ilabel[a0 + best] != 0, a.graph[a0 + best], a.acoustic[a0 + best], false);
          else
            ++j;
        }
        if (nterm == 0) emit(0, false, 0.0f, 0.0f, true);
      }
      // lanes of the position's rounds that carry no term
      for (int jj = max(max(nterm, 1), jlo); jj < jhi; ++jj) {
        if (jj % G >= AL) continue;
        const int rr = r + jj / G;
        uint32_t w = kNil | dummy;
        if (rr == r) w |= kFirst;
        if (rr == r + nr - 1) w |= kLast;
        put(k, rr, jj % G, w, INFINITY, 0.0f);
      }
    }
    __syncthreads();
    if (fwd) {  // position S of the lattices that have it in this tile
      for (int k = 0; k < LPW; ++k) {
        if (!sh_fin[k]) continue;
        const int u = sh_lat[k], S = sh_S[k];
        const int64_t s0 = a.state_off[u];
        const int64_t sb = s0 + 2 * (int64_t)u;
        const int r = a.r0f[sb + S], nr = a.r0f[sb + S + 1] - r;
        const int nfinal = a.cnt[sb + S + 1];
        const int jlo = max(0, (R0 - r) * G), jhi = min(nr * G, (R1 - r) * G);
        const uint32_t own_cell = (uint32_t)((S & (W - 1)) * LPW + k);
        const uint32_t dummy = (uint32_t)(W * LPW + k) << kWrShift;
        for (int s = tid; s < S; s += kT) {
          if (!dev_is_final(a, s0 + s)) continue;
          const int j = a.frank[s0 + s];
          if (j < jlo || j >= jhi || j % G >= AL) continue;
          const int rr = r + j / G, rel = S - s;
          uint32_t w;
          bool far = false;
          if (rel >= W) {
            w = kFar | (uint32_t)rel;
            far = true;
            nfar++;
          } else {
            w = (uint32_t)(((S - rel) & (W - 1)) * LPW + k);
            if (rel == 1 && j < G) w |= kBypass;
          }
          if (!far) w |= dummy;
          if (rr == r) w |= kFirst;
          if (rr == r + nr - 1) {
            w |= kLast;
            if (j % G == 0 && !far) w = (w & ~(kCellMask << kWrShift)) | kCommit | (own_cell << kWrShift);
          }
          put(k, rr, j % G, w, a.final_graph[s0 + s], a.final_acoustic[s0 + s]);
        }
        // no final state: one NIL term; then the lanes without a term
        for (int jj = max(nfinal, jlo) + tid; jj < jhi; jj += kT) {
          if (jj % G >= AL) continue;
          const int rr = r + jj / G;
          uint32_t w = kNil | (jj == 0 ? own_cell : 0u) | dummy;  // jj == 0: no final state, one NIL term
          if (rr == r) w |= kFirst;
          if (rr == r + nr - 1) {
            w |= kLast;
            if (jj == 0) w = (w & ~(kCellMask << kWrShift)) | kCommit | (own_cell << kWrShift);
          }
          put(k, rr, jj % G, w, INFINITY, 0.0f);
        }
      }
      __syncthreads();
    }
    // the tile leaves as whole lines
    {
      const int64_t base = (job.chunk_off + (int64_t)(R0 >> 2) * row_units) << 2;  // first slot of the tile
      uint4* gw = reinterpret_cast<uint4*>(a.sw + base);
      float4* gg = reinterpret_cast<float4*>(a.sg + base);
      float4* ga = reinterpret_cast<float4*>(a.sa + base);
      const uint4* lw = reinterpret_cast<const uint4*>(tw);
      const float4* lg = reinterpret_cast<const float4*>(tg);
      const float4* la = reinterpret_cast<const float4*>(ta);
      for (int i = tid; i < (tile_slots >> 2); i += kT) {
        gw[i] = lw[i];
        gg[i] = lg[i];
        ga[i] = la[i];
      }
    }
    if (tid < LPW) sh_plo[tid] = sh_next[tid];
    __syncthreads();
  }
  if (nfar) atomicAdd(&sh_far, nfar);
  __syncthreads();
  if (tid == 0 && sh_far) atomicAdd(&a.status[3], sh_far);
}

}  // namespace

hipError_t launch_pack_count(const PackDevArgs& a, int num_lattices, const int class_need[kPackCountClasses], bool any_global,
                             hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  // LDS classes: a lattice runs in the launch of the narrowest class that holds its tables, so a
  // batch of small lattices keeps four workgroups on a CU whatever its largest lattice needs
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
    attr_done = true;
  }
  for (int c = 0; c < kPackCountClasses; ++c)
    if (class_need[c] > 0)
      hipLaunchKernelGGL(pack_count_kernel, dim3(num_lattices), dim3(kTC), (size_t)class_need[c], stream, a,
                         c == 0 ? -1 : kPackCountClassLds[c - 1], kPackCountClassLds[c]);
  if (any_global) hipLaunchKernelGGL(pack_count_global_kernel, dim3(num_lattices), dim3(kT), 0, stream, a, kPackCountMaxLds);
  return hipGetLastError();
}

hipError_t launch_pack_rounds(const PackDevArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rounds_kernel, dim3(num_lattices), dim3(kT), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pack_fill(const PackDevArgs& a, int num_entries, hipStream_t stream) {
  if (num_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_fill_kernel, dim3(num_entries), dim3(kT), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
