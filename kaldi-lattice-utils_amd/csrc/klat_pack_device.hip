// klat_pack_device.hip -- the lockstep arc streams built on the device from the CSR batch already
// in HBM.  Same streams, bit for bit, as the host packer (klat_pack.cc), which stays as the
// reference of the stream format for the CPU-side tests; klat_batch_create uses this one.
//
//   pack_count_kernel  one workgroup per lattice: validates the topology, counts in-degrees,
//                      sorts the arcs by destination (counting sort; sources ascending inside a
//                      destination), and counts the rounds its forward and its backward stream
//                      would take with 1, 2 ... 64 lanes.  These counts go back to the host, which
//                      picks every lattice's lanes and slices the lattices into jobs
//                      (plan_layout, build_jobs in klat_pack.cc).
//   pack_rounds_kernel one workgroup per lattice: prefix sums of the rounds every state takes in
//                      the forward and in the backward stream with the lanes the plan gave it.
//   pack_fill_kernel   one workgroup per (job, lattice slot): every state writes the term words
//                      and costs of its rounds -- ring cells, BYPASS / ONE / FAR / WORD, FIRST /
//                      LAST / COMMIT -- and the padding of its unused lanes; the rounds behind the
//                      lattice's end and the empty slots of a job are padded, too.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_pack_device.h"

namespace klat {

namespace {

constexpr int kT = 256;

__device__ __forceinline__ bool dev_is_final(const PackDevArgs& a, int64_t gs) {
  return !(a.final_graph[gs] == INFINITY && a.final_acoustic[gs] == INFINITY);
}

// exclusive prefix sum of one int per thread; returns the thread's prefix, *total = block sum
__device__ int block_scan(int v, int* total, int* sh /* [kT/64] */) {
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
  for (int w = 0; w < kT / 64; ++w) {
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

__global__ __launch_bounds__(kT) void pack_count_kernel(PackDevArgs a) {
  __shared__ int sh[kT / 64];
  __shared__ int sh_bad, sh_nfinal, sh_mr, sh_mtf, sh_mtb;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  if (S <= 0) {
    if (tid == 0) {
      if (S < 0 || A != 0) {
        if (atomicCAS(&a.status[0], 0, 1 /* KLAT_ERR_INVALID_ARG */) == 0) a.status[1] = u;
      }
    }
    return;
  }
  const int64_t sb = s0 + 2 * (int64_t)u;
  int32_t* cnt = a.cnt + sb;
  int32_t* r0b = a.r0b + sb;  // (scratch for the counting sort's cursors here)
  int32_t* frank = a.frank + s0;
  int32_t* in_arc = a.in_arc + a0;
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  if (tid == 0) sh_bad = 0, sh_nfinal = 0, sh_mr = 1, sh_mtf = 1, sh_mtb = 1;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < S + 2; s += kT) cnt[s] = 0;
  __syncthreads();
  // in-degrees + validation (the host packer's checks: ids in range, arcs grouped by source,
  // every arc goes to a higher state)
  int mr = 1;
  for (int i = tid; i < A; i += kT) {
    const int s = src[i], t = dst[i];
    const int prev = i > 0 ? src[i - 1] : 0;
    if (s < 0 || s >= S || s < prev) {
      atomicMax(&sh_bad, 2);  // malformed CSR
    } else if (t <= s || t >= S) {
      atomicMax(&sh_bad, 1);  // not topologically sorted
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
    if (tid == 0) {
      if (atomicCAS(&a.status[0], 0, sh_bad == 2 ? 1 /* INVALID_ARG */ : 2 /* BAD_TOPOLOGY */) == 0) a.status[1] = u;
    }
    return;
  }
  if (tid == 0) atomicMax(&a.status[2], sh_mr);
  // rank of every final state among the finals (terms of the forward stream's last position)
  int carry = 0;
  for (int base = 0; base < S; base += kT) {
    const int s = base + tid;
    const int f = (s < S && dev_is_final(a, s0 + s)) ? 1 : 0;
    int tot;
    const int ex = block_scan(f, &tot, sh);
    if (s < S) frank[s] = carry + ex;
    carry += tot;
  }
  const int nfinal = carry;
  // forward: rounds of every position p = 0..S (S = the virtual super-final state) for every
  // number of lanes; in-degrees -> exclusive prefix in place (cnt[p] = first in-arc of p, cnt[S] = A)
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
    const int ex_c = block_scan(c, &tot_c, sh);
    if (p <= S) cnt[p] = carry_c + ex_c;
    carry_c += tot_c;
  }
  __syncthreads();
  // stable counting sort of the arcs by destination: scatter, then order every bucket by arc index
  for (int s = tid; s < S; s += kT) r0b[s] = cnt[s];  // cursors (r0b is free until the backward scan)
  __syncthreads();
  for (int i = tid; i < A; i += kT) in_arc[atomicAdd(&r0b[dst[i]], 1)] = i;
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
  }
  __syncthreads();
  // backward: position p = S-1-s holds the out-arcs of s (+ its final weight)
  for (int s = tid; s < S; s += kT) {
    const int nb = lower_bound_src(src, A, s + 1) - lower_bound_src(src, A, s) + (dev_is_final(a, s0 + s) ? 1 : 0);
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
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  if (S <= 0 || a.lat_g[u] < 0) return;
  const int k = a.lat_g[u];
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* cnt = a.cnt + sb;
  int32_t* r0f = a.r0f + sb;
  int32_t* r0b = a.r0b + sb;
  const int32_t* src = a.src + a0;
  const int nfinal = a.frank[s0 + S - 1] + (dev_is_final(a, s0 + S - 1) ? 1 : 0);
  int carry_r = 0;
  for (int base = 0; base <= S; base += kT) {
    const int p = base + tid;
    const int nterm = (p < S) ? cnt[p + 1] - cnt[p] + (p == 0 ? 1 : 0) : nfinal;
    const int nr = (p <= S) ? (max(nterm, 1) + (1 << k) - 1) >> k : 0;
    int tot_r;
    const int ex_r = block_scan(nr, &tot_r, sh);
    if (p <= S) r0f[p] = carry_r + ex_r;
    carry_r += tot_r;
  }
  if (tid == 0) r0f[S + 1] = carry_r;
  int carry_b = 0;
  for (int base = 0; base < S; base += kT) {
    const int p = base + tid;
    int nr = 0;
    if (p < S) {
      const int s = S - 1 - p;
      const int nb = lower_bound_src(src, A, s + 1) - lower_bound_src(src, A, s) + (dev_is_final(a, s0 + s) ? 1 : 0);
      nr = (max(nb, 1) + (1 << k) - 1) >> k;
    }
    int tot;
    const int ex = block_scan(nr, &tot, sh);
    if (p < S) r0b[p] = carry_b + ex;
    carry_b += tot;
  }
  if (tid == 0) r0b[S] = carry_b;
}

__global__ __launch_bounds__(kT) void pack_fill_kernel(PackDevArgs a) {
  __shared__ int sh_far;
  const PackEntry e = a.entries[blockIdx.x];
  const SweepJob job = a.jobs[e.job];
  const int G = job_lanes(job.tag), W = job_ring(job.tag), AL = job_stored_lanes(job.tag), LPW = 64 / G;
  const int tid = threadIdx.x;
  const int lane0 = e.slot * G;
  const int total_rounds = job.nchunks * kChunkRounds;
  auto slot_index = [&](int rr, int lane) -> int64_t {
    return stream_slot(job.chunk_off, G, AL, rr / kChunkRounds, lane, rr % kChunkRounds);
  };
  auto pad_word = [&](int lane) -> uint32_t { return kNil | ((uint32_t)(W * LPW + lane / G) << kWrShift); };
  auto put = [&](int rr, int lane, uint32_t w, float g, float ac) {
    if (lane - lane0 >= AL) return;  // not stored: the kernel makes this lane's padding up itself
    const int64_t ix = slot_index(rr, lane);
    a.sw[ix] = (int32_t)w;
    a.sg[ix] = g;
    a.sa[ix] = ac;
  };
  int first_pad_round = 0;
  if (e.lattice >= 0) {
    const int u = e.lattice;
    const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
    const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
    const int64_t sb = s0 + 2 * (int64_t)u;
    const int32_t* cnt = a.cnt + sb;
    const int32_t* src = a.src + a0;
    const int32_t* dst = a.dst + a0;
    if (tid == 0) sh_far = 0;
    __syncthreads();
    int nfar = 0;
    auto ring_cell = [&](int p) -> uint32_t { return (uint32_t)((p & (W - 1)) * LPW + e.slot); };
    // term j of the position that starts at round r and spans nr rounds
    auto emit = [&](int pos, int r, int nr, int j, int rel, bool word, float g, float ac, bool nil) {
      const int rr = r + j / G, lane = lane0 + j % G;
      uint32_t w;
      bool far = false;
      if (nil) {
        w = kNil | ring_cell(pos);
      } else if (rel == 0) {
        w = kOne | ring_cell(pos);
      } else if (rel >= W) {
        w = kFar | (uint32_t)rel;
        far = true;
        nfar++;
      } else {
        w = ring_cell(pos - rel);
        if (rel == 1 && j < G) w |= kBypass;
      }
      if (word) w |= kWord;
      if (!far) w |= (uint32_t)(W * LPW + lane / G) << kWrShift;
      if (rr == r) w |= kFirst;
      if (rr == r + nr - 1) {
        w |= kLast;
        if (lane == lane0 && !far) w = (w & ~(kCellMask << kWrShift)) | kCommit | (ring_cell(pos) << kWrShift);
      }
      put(rr, lane, w, nil ? INFINITY : g, ac);
    };
    // lanes of the position's rounds that carry no term
    auto pad_rest = [&](int r, int nr, int nterm) {
      for (int j = nterm; j < nr * G; ++j) {
        const int rr = r + j / G, lane = lane0 + j % G;
        uint32_t w = pad_word(lane);
        if (rr == r) w |= kFirst;
        if (rr == r + nr - 1) w |= kLast;
        put(rr, lane, w, INFINITY, 0.0f);
      }
    };
    if (job.dir == 0) {
      const int32_t* r0f = a.r0f + sb;
      const int32_t* in_arc = a.in_arc + a0;
      const int32_t* frank = a.frank + s0;
      for (int p = tid; p < S; p += kT) {
        const int r = r0f[p], nr = r0f[p + 1] - r;
        int j = 0;
        if (p == 0) emit(p, r, nr, j++, 0, false, 0.0f, 0.0f, false);  // alpha[start] = 0
        for (int q = cnt[p]; q < cnt[p + 1]; ++q) {
          const int i = in_arc[q];
          emit(p, r, nr, j++, p - src[i], a.ilabel[a0 + i] != 0, a.graph[a0 + i], a.acoustic[a0 + i], false);
        }
        if (j == 0) emit(p, r, nr, j++, 0, false, 0.0f, 0.0f, true);
        pad_rest(r, nr, j);
      }
      {  // position S: one term per final state, in state order
        const int r = r0f[S], nr = r0f[S + 1] - r;
        int nfinal = 0;
        for (int s = tid; s < S; s += kT)
          if (dev_is_final(a, s0 + s))
            emit(S, r, nr, frank[s], S - s, false, a.final_graph[s0 + s], a.final_acoustic[s0 + s], false);
        if (tid == 0) {
          for (int s = S - 1; s >= 0; --s)
            if (dev_is_final(a, s0 + s)) {
              nfinal = frank[s] + 1;
              break;
            }
          if (nfinal == 0) emit(S, r, nr, nfinal++, 0, false, 0.0f, 0.0f, true);
          pad_rest(r, nr, nfinal);
        }
      }
      first_pad_round = r0f[S + 1];
    } else {
      const int32_t* r0b = a.r0b + sb;
      for (int p = tid; p < S; p += kT) {
        const int s = S - 1 - p;
        const int r = r0b[p], nr = r0b[p + 1] - r;
        const int lo = lower_bound_src(src, A, s), hi = lower_bound_src(src, A, s + 1);
        int j = 0;
        if (dev_is_final(a, s0 + s)) emit(p, r, nr, j++, 0, false, a.final_graph[s0 + s], a.final_acoustic[s0 + s], false);
        // farthest destination first, nearest last; arcs with the same destination in arc order
        int last_dst = 0x7FFFFFFF, last_idx = -1;
        for (int k = lo; k < hi; ++k) {
          int best = -1, best_dst = -1;
          for (int i = lo; i < hi; ++i) {
            const int d = dst[i];
            const bool after = d < last_dst || (d == last_dst && i > last_idx);
            if (after && (d > best_dst)) {
              best = i;
              best_dst = d;
            }
          }
          last_dst = best_dst;
          last_idx = best;
          emit(p, r, nr, j++, best_dst - s, a.ilabel[a0 + best] != 0, a.graph[a0 + best], a.acoustic[a0 + best], false);
        }
        if (j == 0) emit(p, r, nr, j++, 0, false, 0.0f, 0.0f, true);
        pad_rest(r, nr, j);
      }
      first_pad_round = r0b[S];
    }
    if (nfar) atomicAdd(&sh_far, nfar);
    __syncthreads();
    if (tid == 0 && sh_far) atomicAdd(&a.status[3], sh_far);
  }
  // rounds behind the lattice's end (all rounds of an empty slot)
  const int npad = (total_rounds - first_pad_round) * G;
  for (int k = tid; k < npad; k += kT) {
    const int rr = first_pad_round + k / G, lane = lane0 + k % G;
    put(rr, lane, pad_word(lane), INFINITY, 0.0f);
  }
}

}  // namespace

hipError_t launch_pack_count(const PackDevArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_count_kernel, dim3(num_lattices), dim3(kT), 0, stream, a);
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
