// klat_level.hip -- forward-backward by topological LEVELS, for lattices whose depth is far below
// their number of states (real word lattices: depth = number of words; SURVEY.md 7 "hard parts",
// BASELINE.json north_star "topological levels are processed in lockstep").
//
// The lockstep streams (klat_sweep.hip) deal a lattice state by state: S states are at least S
// dependent rounds whatever the lattice's shape.  That is the right trade for time-ordered chains
// (every state linked to its successor: depth = S, the batch provides the parallelism), and it costs
// a full batch of wide lattices nothing -- but a wide lattice alone, or a batch too small to fill
// the chip, waits S rounds for work that is depth steps deep.  Here one WAVE owns one lattice:
//
//   level_plan_kernel   Kahn's algorithm over the CSR arcs: a state joins the next level when its
//                       last in-arc has been visited (counters and the level-ordered state list in
//                       the lattice's scratch rows, L2-resident); also the row table of the arcs.
//                       Gives up (depth = -1) once the depth passes S / 8: such a lattice stays
//                       with the streams.
//   level_build_kernel  the arcs once more as the lattice's LEVEL STREAM (klat_level.h): records of
//                       up to 64 states of one level with their terms (in-arcs forward, out-arcs
//                       backward), laid out in exactly the order the sweep consumes them.
//   level_sweep_kernel  one wave per lattice.  The stream comes into a 64-KiB LDS ring by LDS-DMA
//                       (global_load_lds_dwordx4, 1 KiB per instruction, up to 56 in flight, no
//                       registers), so a record is in LDS long before its turn and the dependent
//                       chain of a level is LDS latency only.  The values of all states live in
//                       LDS (8 B per state).  Every lane folds the in-arcs (forward) or out-arcs
//                       (backward) of one state into a running log-sum-exp -- what
//                       ComputeLatticeAlphasAndBetas computes with LogAdd (call
//                       site latbin/lattice-to-word-frame-post.cc:88-90), one exp per arc and one log
//                       per state -- or with (min,+) for the Viterbi twin
//                       (latbin/lattice-prune-dyn-beam.cc:27-90).
//                       Results go to the same padded alpha / beta rows the stream kernel writes
//                       (slot S of a forward row = tot_forward), so everything downstream is shared.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "klat_level.h"
#include "klat_math.h"

namespace klat {

namespace {

__global__ __launch_bounds__(64) void level_plan_kernel(LevelPlanArgs a) {
  __shared__ int tail_sh;
  const int u = blockIdx.x, lane = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  // (launched behind the count pass without waiting for its verdict: a batch with a malformed lattice is walked by nobody)
  if (S < kLevelMinStates || S > kLevelMaxStates || (a.status && a.status[0] != 0) || (a.only && !a.only[u])) {
    if (lane == 0) a.depth[u] = -1;
    return;
  }
  const int64_t sb = s0 + 2 * (int64_t)u;  // the lattice's rows of the (S + 2)-sized arrays
  const int32_t* cnt = a.cnt + sb;          // first in-arc of every state (exclusive prefix; cnt[S] = A)
  int32_t* left = a.counters + sb;          // in-arcs not yet visited
  int32_t* order = a.order + s0;            // states, level by level
  int32_t* off = a.level_off + sb;          // first entry of every level in `order`
  const int32_t* row = a.out_begin + sb;    // first out-arc of every state (row[S] = A), from the count pass
  const int32_t* dst = a.dst + a0;
  const int max_depth = S / kLevelMinWidth;

  // A time-ordered chain (nearly every state linked to its successor) is as deep as it is long:
  // tell it from one pass over the arcs instead of from max_depth steps of Kahn's algorithm.
  // With g states lacking the link, some run of linked states is at least (S - g) / (g + 1) long.
  // (the count pass of the stream packer counted the linked states: the spare entry of the row table)
  const int linked = row[S + 1];
  if (linked > (S - linked + 1) * max_depth) {
    if (lane == 0) a.depth[u] = -1;
    return;
  }

  // level 0: the states without in-arcs, in state order
  int tail = 0;
  for (int base = 0; base < S; base += 64) {
    const int s = base + lane;
    const int c = s < S ? cnt[s + 1] - cnt[s] : -1;
    if (s < S) left[s] = c;
    const unsigned long long m = __ballot(c == 0);
    if (c == 0) order[tail + __popcll(m & ((1ull << lane) - 1ull))] = s;
    tail += __popcll(m);
  }
  if (lane == 0) {
    tail_sh = tail;
    off[0] = 0;
  }
  __threadfence_block();
  __syncthreads();
  int head = 0, level = 0, records = 0, max_terms = 0;
  while (head < tail) {
    for (int c0 = head; c0 < tail; c0 += 64) {  // one record of the level stream per 64 states
      const int idx = c0 + lane;
      const bool valid = idx < tail;
      const int s = valid ? order[idx] : 0;
      int din = valid ? cnt[s + 1] - cnt[s] : 0, dout = valid ? row[s + 1] - row[s] : 0;
      if (valid)
        for (int i = row[s]; i < row[s + 1]; ++i) {
          const int d = dst[i];
          if (atomicSub(&left[d], 1) == 1) order[atomicAdd(&tail_sh, 1)] = d;  // its last in-arc: the next level
        }
      for (int d = 32; d > 0; d >>= 1) din += __shfl_xor(din, d, 64), dout += __shfl_xor(dout, d, 64);
      max_terms = max(max_terms, max(din, dout));
      ++records;
    }
    __threadfence_block();
    __syncthreads();
    head = tail;
    tail = tail_sh;
    ++level;
    if (lane == 0) off[level] = head;
    if (level > max_depth) {  // too deep for its size: the streams serve it better
      if (lane == 0) a.depth[u] = -1;
      return;
    }
    __syncthreads();
  }
  // (a cyclic input never gets here with all states: the stream packer's validation has already
  // refused it -- every arc goes to a higher state)
  if (lane == 0) {
    a.depth[u] = (head == S && max_terms <= kLevelMaxRecordTerms) ? level : -1;
    a.records[u] = records;
  }
}

// The level stream of a lattice (layout: klat_level.h), one record per 64 states of a level.
__global__ __launch_bounds__(64) void level_build_kernel(LevelBuildArgs a) {
  const int lane = threadIdx.x;
  const int u = a.lattices[blockIdx.x];
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* cnt = a.cnt + sb;
  const int32_t* row = a.out_begin + sb;
  const int32_t* off = a.level_off + sb;
  const int32_t* order = a.order + s0;
  const int32_t* in_arc = a.in_arc + a0;
  const int depth = a.depth[u];
  int32_t* out = a.stream + a.stream_off[blockIdx.x] * kLevelBlockDwords;
  int64_t cur = 0;
  for (int dir = 0; dir < 2; ++dir)
    for (int li = 0; li < depth; ++li) {
      const int l = dir == 0 ? li : depth - 1 - li;
      const int j1 = off[l + 1];
      for (int c0 = off[l]; c0 < j1; c0 += 64) {
        const int n = min(64, j1 - c0);
        const bool valid = lane < n;
        const int s = valid ? order[c0 + lane] : 0;
        const int beg = dir == 0 ? cnt[s] : row[s];
        const int deg = valid ? (dir == 0 ? cnt[s + 1] : row[s + 1]) - beg : 0;
        int x = deg;
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(x, d, 64);
          if (lane >= d) x += t;
        }
        const int nt = __shfl(x, 63, 64), rel = x - deg;
        int32_t* rec = out + cur;
        if (lane == 0) rec[0] = n, rec[1] = nt;
        if (valid) {
          rec[2 + lane] = s;
          rec[2 + n + lane] = rel;
          if (lane == n - 1) rec[2 + 2 * n] = nt;
          if (dir == 1) {
            rec[3 + 2 * n + 2 * lane] = __float_as_int(a.final_graph[s0 + s]);
            rec[4 + 2 * n + 2 * lane] = __float_as_int(a.final_acoustic[s0 + s]);
          }
          int32_t* T = rec + 3 + (dir == 0 ? 2 : 4) * n + 3 * rel;
          for (int k = 0; k < deg; ++k) {
            // forward: the in-arcs, sources ascending and in arc order -- the order in which
            // ComputeLatticeAlphasAndBetas meets them; backward: the out-arcs in arc order
            const int64_t i = a0 + (dir == 0 ? in_arc[beg + k] : beg + k);
            T[3 * k] = (dir == 0 ? a.src[i] : a.dst[i]) | (a.ilabel[i] != 0 ? (int32_t)0x80000000 : 0);
            T[3 * k + 1] = __float_as_int(a.graph[i]);
            T[3 * k + 2] = __float_as_int(a.acoustic[i]);
          }
        }
        cur += 3 + (dir == 0 ? 2 : 4) * n + 3 * nt;
      }
    }
}

// exp(x) for x <= 0 (-inf included), relative error below 1e-14: the sweep is one wave deep, so every
// instruction of its chain counts -- 2^k * e^r with |r| <= ln2 / 2 and a degree-12 Taylor polynomial.
__device__ __forceinline__ double lv_exp(double x) {
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return x < -745.2 ? 0.0 : ldexp(p, (int)k);  // (x = -inf: r is NaN, the select drops it)
}

// A running log-sum-exp: value = m + log(s), four terms at a time without a branch per term.  (The
// stream kernel does not reproduce LogAdd's rounding either -- DESIGN.md 3; both stay within 1e-11.)
struct Lse {
  double m = -INFINITY, s = 0.0;
  __device__ __forceinline__ void start(double x) { m = x, s = x > -INFINITY ? 1.0 : 0.0; }
  __device__ __forceinline__ void add4(double x0, double x1, double x2, double x3) {  // -inf: no term
    const double mn = fmax(fmax(m, fmax(x0, x1)), fmax(x2, x3));
    if (mn > -INFINITY) {
      s = fma(s, lv_exp(m - mn), (lv_exp(x0 - mn) + lv_exp(x1 - mn)) + (lv_exp(x2 - mn) + lv_exp(x3 - mn)));
      m = mn;
    }
  }
  __device__ __forceinline__ void add(double x) {
    if (x > m) {
      s = fma(s, lv_exp(m - x), 1.0);
      m = x;
    } else if (x > -INFINITY) {
      s += lv_exp(x - m);
    }
  }
  __device__ __forceinline__ double value() const { return s > 0.0 ? m + log(s) : -INFINITY; }
};

// One 1-KiB block of the stream straight into LDS (LDS-DMA: no registers; lane i's 16 bytes land at
// lds_dst + 16 i).  hipcc does not count it: every wait on it below is an explicit s_waitcnt.
__device__ __forceinline__ void glds_block(const int32_t* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// waits until at most `n` of the wave's vector-memory operations are outstanding (they retire in
// issue order); returns the bound it actually waited for (<= n)
__device__ __forceinline__ int wait_vm(int n) {
#define KLAT_VM(N)                                   \
  if (n >= N) {                                      \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); \
    return N;                                        \
  }
  KLAT_VM(48) KLAT_VM(32) KLAT_VM(24) KLAT_VM(16) KLAT_VM(12) KLAT_VM(8) KLAT_VM(6) KLAT_VM(4) KLAT_VM(3) KLAT_VM(2) KLAT_VM(1)
#undef KLAT_VM
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return 0;
}

#ifdef KLAT_LEVEL_PROFILE  // diagnostic build: cycles of a wave in fetch (0 -> 1) and fold (1 -> 2)
#define KLAT_LV_T(k)                                                    \
  {                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
    const long long now = (long long)__builtin_amdgcn_s_memtime();      \
    if (k == 1) lv_t[0] += now - lv_last;                               \
    if (k == 2) lv_t[1] += now - lv_last;                               \
    lv_last = now;                                                      \
  }
#else
#define KLAT_LV_T(k)
#endif

template <bool TROPICAL>
__global__ __launch_bounds__(64) void level_sweep_kernel(LevelSweepArgs a, int lds_states) {
#ifdef KLAT_LEVEL_PROFILE
  long long lv_t[2] = {0, 0}, lv_last = 0;
  const long long lv_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
  constexpr int RB = kLevelRingBlocks, BD = kLevelBlockDwords, RM = RB * BD - 1;
  constexpr int kMaxInFlight = 56;  // vmcnt counts to 63
  extern __shared__ double lv_smem[];
  int32_t* ring = reinterpret_cast<int32_t*>(lv_smem);              // [RB blocks] the stream, as it arrives
  double* val = reinterpret_cast<double*>(ring + RB * BD);          // [lds_states] the values of all states
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)ring;
  const int lane = threadIdx.x;
  const int u = a.lattices[blockIdx.x];
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  const int records = a.records[u];
  const int nblocks = (int)((level_stream_dwords(records, S, A) + BD - 1) / BD);
  const int32_t* stream = a.stream + a.stream_off[blockIdx.x] * BD + lane * 4;
  const ScaleParams sp = a.sp;
  double* fwd = a.alpha + padded_state_base(s0, u);
  double* bwd = a.beta + padded_state_base(s0, u);
  const double zero = TROPICAL ? INFINITY : -INFINITY;  // the semiring's Zero as a value
  auto final_cost = [&](int s) { return arc_cost(a.final_graph[s0 + s], a.final_acoustic[s0 + s], false, sp); };
  auto is_final = [&](int s) { return !(a.final_graph[s0 + s] == INFINITY && a.final_acoustic[s0 + s] == INFINITY); };

  int issued = 0, in_flight = 0;  // blocks asked for; an upper bound of those not yet landed
  int pos = 0;                    // the dword of the stream the next record starts at
  bool broken = false;            // (wave-uniform) the stream does not parse: the results are NaN
  // asks for blocks below `limit` (ring space) while the in-flight budget lasts
  auto top_up = [&]() {
    const int limit = min(nblocks, pos / BD + RB);
    while (issued < limit && in_flight < kMaxInFlight) {
      glds_block(stream + (int64_t)issued * BD, ring_lds + (uint32_t)(issued & (RB - 1)) * (BD * 4));
      ++issued;
      ++in_flight;
    }
  };
  // block `b` has landed when this returns (b < nblocks; the record it belongs to fits the ring)
  auto need = [&](int b) {
    while (issued <= b) {  // the in-flight budget was in the way: make room
      in_flight = min(in_flight, wait_vm(kMaxInFlight / 2));
      top_up();
    }
    const int after = issued - 1 - b;  // blocks asked for after b
    if (after < in_flight) in_flight = wait_vm(after);
  };

  // the records of one direction: every lane folds the terms of its state
  auto sweep = [&](auto forward_tag) {
    constexpr bool FWD = decltype(forward_tag)::value;
    for (int r = 0; r < records; ++r) {
      KLAT_LV_T(0);
      top_up();
      need((pos + 1) / BD);
      const int n = __builtin_amdgcn_readfirstlane(ring[pos & RM]);
      const int nt = __builtin_amdgcn_readfirstlane(ring[(pos + 1) & RM]);
      const int tb = pos + 3 + (FWD ? 2 : 4) * n;  // the record's terms
      const int end = tb + 3 * nt;
      if ((unsigned)(n - 1) >= 64u || (unsigned)nt > (unsigned)kLevelMaxRecordTerms || (end - 1) / BD >= nblocks) {
        broken = true;  // not a record: never wait for blocks that do not exist
        return;
      }
      need((end - 1) / BD);
      KLAT_LV_T(1);
      if (lane < n) {
        const int s = ring[(pos + 2 + lane) & RM];
        const int t0 = ring[(pos + 2 + n + lane) & RM], t1 = ring[(pos + 3 + n + lane) & RM];
        const float fg = FWD ? 0.f : __int_as_float(ring[(pos + 3 + 2 * n + 2 * lane) & RM]);
        const float fa = FWD ? 0.f : __int_as_float(ring[(pos + 4 + 2 * n + 2 * lane) & RM]);
        Lse acc;
        double best = INFINITY;
        if (FWD ? s == 0 : !(fg == INFINITY && fa == INFINITY)) {  // the constant-one source: alpha[start] = 0 / the final weight
          const double x = FWD ? 0.0 : arc_cost(fg, fa, false, sp);
          if (TROPICAL) best = x;
          else acc.start(-x);
        }
        for (int t = t0; t < t1; t += 4) {  // four terms' loads together
          int32_t ow[4];
          float g[4], ac[4];
          double v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = tb + 3 * min(t + k, t1 - 1);
            ow[k] = ring[i & RM];
            g[k] = __int_as_float(ring[(i + 1) & RM]);
            ac[k] = __int_as_float(ring[(i + 2) & RM]);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = val[ow[k] & 0x7FFFFFFF];
          double x[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double c = arc_cost(g[k], ac[k], ow[k] < 0, sp);
            x[k] = t + k < t1 ? (TROPICAL ? v[k] + c : v[k] - c) : zero;
          }
          if (TROPICAL) best = fmin(fmin(best, fmin(x[0], x[1])), fmin(x[2], x[3]));
          else acc.add4(x[0], x[1], x[2], x[3]);
        }
        val[s] = TROPICAL ? best : acc.value();
      }
      pos = end;
      __builtin_amdgcn_wave_barrier();
      KLAT_LV_T(2);
    }
  };

  sweep(std::true_type());
  {  // tot_forward (slot S of the row): the final states in state order
    Lse tot;
    double best = INFINITY;
    for (int base = 0; base < S; base += 64) {
      const int s = base + lane;
      const bool fin = s < S && is_final(s);
      unsigned long long m = __ballot(fin);
      const double mine = fin ? (TROPICAL ? val[s] + final_cost(s) : val[s] - final_cost(s)) : zero;
      while (m) {  // wave-uniform
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        const double t = __shfl(mine, k, 64);
        if (TROPICAL) best = fmin(best, t);
        else tot.add(t);
      }
    }
    for (int s = lane; s < S; s += 64) fwd[s] = val[s];
    if (lane == 0) fwd[S] = TROPICAL ? best : tot.value();
  }
  // (the backward values replace the forward ones state by state: a state reads only deeper levels)
  in_flight = min(in_flight, 63);
  if (!broken) sweep(std::false_type());
  for (int s = lane; s < S; s += 64) bwd[s] = broken ? NAN : val[s];
  if (broken && lane == 0) fwd[S] = NAN;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA outlives the wave
#ifdef KLAT_LEVEL_PROFILE
  if (lane == 0 && blockIdx.x == 0)
    printf("level sweep: %d records per direction, fetch %lld fold %lld total %lld (s_memtime ticks)\n", records, lv_t[0], lv_t[1],
           (long long)__builtin_amdgcn_s_memtime() - lv_begin);
#endif
}

}  // namespace

hipError_t launch_level_plan(const LevelPlanArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(level_plan_kernel, dim3(num_lattices), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_level_build(const LevelBuildArgs& a, int num_level_lattices, hipStream_t stream) {
  if (num_level_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(level_build_kernel, dim3(num_level_lattices), dim3(64), 0, stream, a);
  return hipGetLastError();
}

size_t level_sweep_lds_bytes(int max_states) {
  return sizeof(int32_t) * (size_t)kLevelRingBlocks * kLevelBlockDwords + sizeof(double) * (size_t)((max_states + 1) & ~1);
}

hipError_t launch_level_sweep(const LevelSweepArgs& a, int num_level_lattices, int max_states, bool tropical, hipStream_t stream) {
  if (num_level_lattices <= 0) return hipSuccess;
  const int lds_states = (max_states + 1) & ~1;
  const size_t lds = level_sweep_lds_bytes(max_states);
  const void* fn = tropical ? reinterpret_cast<const void*>(&level_sweep_kernel<true>) : reinterpret_cast<const void*>(&level_sweep_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (tropical)
    hipLaunchKernelGGL(level_sweep_kernel<true>, dim3(num_level_lattices), dim3(64), lds, stream, a, lds_states);
  else
    hipLaunchKernelGGL(level_sweep_kernel<false>, dim3(num_level_lattices), dim3(64), lds, stream, a, lds_states);
  return hipGetLastError();
}

}  // namespace klat
