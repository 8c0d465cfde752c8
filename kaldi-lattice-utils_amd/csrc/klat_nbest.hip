// klat_nbest.hip -- the n lowest-cost complete paths of every lattice of a batch in the tropical
// semiring over caller-supplied arc costs: the n-best extension of the ShortestPath(new_fst,
// &nbest_fst, 1) call at latbin/lattice-best-path2.cc:185 (BASELINE configs[2] asks for n = 100;
// the reference's own --nbest option lives at kwsbin2/lattice-char-index-segment.cc:277,289).
//
// Costs are float32 and add left to right along the path (StdArc / TropicalWeight::Times), final
// states add nothing (:157-160), paths are distinct state/arc sequences (what ShortestPath's
// n-best returns without `unique`).
//
// One wave per lattice (single-wave workgroups on a queue).  Every state owns a list of the
// kNbestWidth = 128 best partial paths reaching it: 64-bit keys (cost bits << 32 | back pointer),
// ascending, two per lane, in an L2-resident scratch row of 1 KiB.  States are visited in
// topological (= numeric) order; a finished list is pushed along each out-arc: its keys plus the
// arc cost, re-tagged with (arc, rank), are merged into the destination's list by one bitonic
// half-cleaner against the reversed destination list followed by a 128-element bitonic merge (one
// in-register stage + six cross-lane stages).  The lists of the final states are merged the same
// way into the result; one lane per result path then walks the back pointers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_nbest.h"

namespace klat {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint64_t kInfKey = ((uint64_t)0x7F800000u << 32) | kNone;

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
  return ((uint64_t)hi << 32) | lo;
}

// (x0, x1) = elements lane and lane + 64 of an ascending list; (m0, m1) = elements 127 - lane and
// 63 - lane of another ascending list.  Returns the 128 smallest of the union, ascending.
__device__ __forceinline__ void merge128(uint64_t& x0, uint64_t& x1, uint64_t m0, uint64_t m1, int lane) {
  uint64_t c0 = x0 < m0 ? x0 : m0;  // half-cleaner: the smaller half of the union, as a bitonic sequence
  uint64_t c1 = x1 < m1 ? x1 : m1;
  {
    const uint64_t lo = c0 < c1 ? c0 : c1, hi = c0 < c1 ? c1 : c0;  // stride 64: in registers
    c0 = lo, c1 = hi;
  }
#pragma unroll
  for (int t = 32; t >= 1; t >>= 1) {
    const bool upper = (lane & t) != 0;
    const uint64_t p0 = shfl_xor64(c0, t), p1 = shfl_xor64(c1, t);
    c0 = ((p0 < c0) != upper) ? p0 : c0;  // lower lane keeps the minimum, upper lane the maximum
    c1 = ((p1 < c1) != upper) ? p1 : c1;
  }
  x0 = c0, x1 = c1;
}

__device__ __forceinline__ uint64_t add_cost(uint64_t key, float w, uint32_t tag) {
  const float c = __uint_as_float((uint32_t)(key >> 32)) + w;
  // an empty slot stays empty (inf + w = inf); a real entry takes the new back pointer
  return (c == INFINITY) ? kInfKey : (((uint64_t)__float_as_uint(c) << 32) | tag);
}

__global__ __launch_bounds__(64) void nbest_kernel(NbestArgs a) {
  const int lane = threadIdx.x;
  uint64_t* __restrict__ lists = a.scratch + (size_t)blockIdx.x * (size_t)a.scratch_states * kNbestWidth;
  for (;;) {
    int u = 0;
    if (lane == 0) u = atomicAdd(a.next_lattice, 1);
    u = __shfl(u, 0, 64);
    if (u >= a.num_lattices) break;  // every wave reaches this: the counter only grows
    const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
    const int S = (int)(a.state_off[u + 1] - s0);
    const int nA = (int)(a.arc_off[u + 1] - a0);
    int npaths = 0;
    uint64_t r0 = kInfKey, r1 = kInfKey;  // the result list
    if (S > 0) {
      for (int s = 0; s < S; ++s) {
        lists[(size_t)s * kNbestWidth + lane] = (s == 0 && lane == 0) ? (uint64_t)kNone : kInfKey;  // start: cost 0, no back pointer
        lists[(size_t)s * kNbestWidth + 64 + lane] = kInfKey;
      }
      __threadfence_block();
      int i = 0;  // arcs are grouped by source, in state order
      for (int s = 0; s < S; ++s) {
        const uint64_t l0 = lists[(size_t)s * kNbestWidth + lane], l1 = lists[(size_t)s * kNbestWidth + 64 + lane];
        const bool reachable = __shfl((int)((uint32_t)(l0 >> 32) != 0x7F800000u), 0, 64) != 0;  // entry 0 is the best
        const float fg = a.final_graph[s0 + s], fa = a.final_acoustic[s0 + s];
        if (reachable && !(fg == INFINITY && fa == INFINITY)) {
          // a final state: its list joins the result, tagged (state, rank)
          const uint64_t m0 = add_cost(shfl_xor64(l1, 63), 0.0f, ((uint32_t)s << 7) | (uint32_t)(127 - lane));
          const uint64_t m1 = add_cost(shfl_xor64(l0, 63), 0.0f, ((uint32_t)s << 7) | (uint32_t)(63 - lane));
          merge128(r0, r1, m0, m1, lane);
        }
        for (; i < nA && a.src[a0 + i] == s; ++i) {
          if (!reachable) continue;
          const float w = a.arc_cost[a0 + i];
          if (!(w >= 0.0f)) {  // negative or NaN: not a tropical cost this kernel orders correctly
            if (lane == 0) atomicExch(a.status, 1 + u);
            continue;
          }
          if (w == INFINITY) continue;
          const int d = a.dst[a0 + i];
          uint64_t x0 = add_cost(l0, w, ((uint32_t)i << 7) | (uint32_t)lane);
          uint64_t x1 = add_cost(l1, w, ((uint32_t)i << 7) | (uint32_t)(64 + lane));
          uint64_t* dl = lists + (size_t)d * kNbestWidth;
          const uint64_t m0 = dl[127 - lane], m1 = dl[63 - lane];  // the destination's list, reversed
          merge128(x0, x1, m0, m1, lane);
          dl[lane] = x0;
          dl[64 + lane] = x1;
          __threadfence_block();  // the next arc may read this list back (same wave, through L1)
        }
      }
      // ---- the result: up to k paths, walked back by one lane each (two passes: count, then write)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint64_t key = half ? r1 : r0;
        const int n = half * 64 + lane;
        const bool valid = n < a.k && (uint32_t)(key >> 32) != 0x7F800000u;
        npaths += __popcll(__ballot(valid));
        if (!valid) continue;
        const int64_t slot = (int64_t)u * a.k + n;
        a.path_cost[slot] = __uint_as_float((uint32_t)(key >> 32));
        int len = 0;
        for (int pass = 0; pass < 2; ++pass) {
          uint32_t tag = (uint32_t)key;
          int state = (int)(tag >> 7), rank = (int)(tag & 127);
          int pos = 0;
          for (;;) {
            const uint32_t back = (uint32_t)lists[(size_t)state * kNbestWidth + rank];
            if (back == kNone) break;
            const int arc = (int)(back >> 7);
            rank = (int)(back & 127);
            state = a.src[a0 + arc];
            const int32_t label = a.olabel[a0 + arc];
            if (label != 0) {
              if (pass == 1 && len - 1 - pos < a.max_labels) a.path_labels[slot * a.max_labels + (len - 1 - pos)] = label;
              ++pos;
            }
          }
          len = pos;
        }
        a.path_len[slot] = len;
        if (len > a.max_labels) atomicExch(a.status, -(1 + u));
      }
    }
    if (lane == 0) a.num_paths[u] = npaths;
    __threadfence_block();  // the next lattice reuses the scratch rows
  }
}

}  // namespace

hipError_t launch_nbest(const NbestArgs& a, int num_workgroups, hipStream_t stream) {
  if (a.num_lattices <= 0 || num_workgroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(nbest_kernel, dim3(num_workgroups), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
