// klat_capi.cc -- the C ABI declared in include/klat.h (context, batch, sweep, fetches).
//
// There is no CPU fallback: every entry point that computes needs the HIP runtime and a GPU,
// and fails loudly (KLAT_ERR_NO_DEVICE / KLAT_ERR_HIP) otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../../include/klat.h"
#include "klat_internal.h"
#include "klat_level.h"
#include "klat_pack_device.h"

namespace klat {

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

DevPool*& current_pool() {
  static thread_local DevPool* pool = nullptr;
  return pool;
}

int hip_fail(hipError_t e, const char* what) {
  if (current_pool()) current_pool()->failing = true;  // the rest of this call frees instead of recycling (DevPool)
  return set_error(e == hipErrorOutOfMemory ? KLAT_ERR_OOM : KLAT_ERR_HIP,
                   std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace klat

using namespace klat;

// SweepArgs::role_mask of the classes' sweeps (tools/sweep_roles.py scans the masks on the C2 batch)
static const uint32_t kSweepRoleMask = 0x100;  // (in a forward job wave 0 walks the chain, in a backward job wave 1)

// Runs the pieces of one sweep side by side: the first on the context's stream, the others on side
// streams that fork from it and join it again, so that the few long workgroups of a skewed batch run
// beside the many short ones.  The pieces are the layout classes (the class with the longest chain
// first) and, before them, the lattices swept in segments (klat_chain.hip).
// A piece names its stream (0: the context's; 1 ..: a side stream) and is launched in list order, so one stream
// may take several pieces: a class's fast sweep goes out FIRST, before the eight launches of the segment path
// (a launch is ~10 us of host time, and a class that started 200 us late finished 200 us late), its exact
// sweep after them.
static int ensure_side_streams(klat_ctx* c) {
  if (c->fork_ev) return KLAT_OK;  // (a helper context of klat_forward_backward_host is made without them)
  KLAT_HIP(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
  for (int i = 0; i < klat_ctx::kSideStreams; ++i) {
    KLAT_HIP(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
    KLAT_HIP(hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
  }
  return KLAT_OK;
}
int klat::run_side_by_side(klat_ctx* c, const std::vector<SweepPiece>& work) {
  int most = 0;
  for (const SweepPiece& w : work) most = std::max(most, w.slot);
  if (most == 0) {
    for (const SweepPiece& w : work) KLAT_HIP(w.launch(c->stream));
    return KLAT_OK;
  }
  int rc0 = ensure_side_streams(c);
  if (rc0 != KLAT_OK) return rc0;
  const int side_limit = klat_ctx::kSideStreams;
  KLAT_HIP(hipEventRecord(c->fork_ev, c->stream));
  const int used = std::min(most, side_limit);
  for (int i = 0; i < used; ++i) KLAT_HIP(hipStreamWaitEvent(c->side[i], c->fork_ev, 0));
  for (const SweepPiece& w : work) KLAT_HIP(w.launch(w.slot == 0 ? c->stream : c->side[(w.slot - 1) % side_limit]));
  for (int i = 0; i < used; ++i) {
    KLAT_HIP(hipEventRecord(c->join_ev[i], c->side[i]));
    KLAT_HIP(hipStreamWaitEvent(c->stream, c->join_ev[i], 0));
  }
  return KLAT_OK;
}

template <class F>
static int for_each_class(klat_batch* b, F launch) {
  std::vector<SweepPiece> work;
  int slot = 0;
  for (const LayoutClass& lc : b->classes) {
    const LayoutClass* lcp = &lc;
    work.push_back(SweepPiece{slot++, [&launch, lcp](hipStream_t s) { return launch(*lcp, s); }});
  }
  return run_side_by_side(b->ctx, work);
}

// Which lattices are swept in segments (klat_chain.hip), and into how many.  A lattice's segments are equally
// long and hold about 4 x KLAT_CHAIN_SEG arc records each (default 128: the length, in states, of a segment of a
// lattice of fanout 3), so that the walks' lanes finish together whatever the lattices' fanout.  Candidates: at
// least two segments long, no arc longer than kChainMaxSpan states, not on the level path.  A batch the chip is
// busy with anyway keeps its streams: the segment path walks every arc (c + 2) / 2 times as often as a stream
// does.  Otherwise the planner's own model of the streams (plan_cycles) is weighed against a model of the
// segment path with every candidate in it, and they all go if that beats the streams by a quarter.
// KLAT_CHAIN=0 / =1: never / every candidate whatever the models say.
static int chain_segment_states(int64_t S, int64_t A, int seg) {
  const double fan = S > 0 ? (double)A / (double)S : 1.0;
  return std::max(2 * kChainMaxSpan, (int)(4.0 * (double)seg / (fan + 1.0)));
}
static void choose_chain_lattices(int64_t U, const int64_t* state_off, const int64_t* arc_off, const LatticeRounds* lr,
                                  const klat_pack_options& o, const uint8_t* on_level, std::vector<uint8_t>* on_chain,
                                  std::vector<ChainLattice>* out, int* seg_len_out, int* c_out, size_t* num_records) {
  const char* env = std::getenv("KLAT_CHAIN");
  const int mode = env ? std::atoi(env) : -1;
  out->clear();
  if (mode == 0 || U <= 0) return;
  const char* seg_env = std::getenv("KLAT_CHAIN_SEG");
  int seg = seg_env ? std::atoi(seg_env) : 128;
  seg = std::max(kChainMaxSpan, std::min(seg, 1 << 20));
  std::vector<int64_t> cand;
  double sum_states = 0, longest = 0;
  for (int64_t u = 0; u < U; ++u) {
    const int64_t S = state_off[u + 1] - state_off[u];
    if (on_level[u] || S <= 0) continue;
    sum_states += (double)S;
    longest = std::max(longest, (double)S);
  }
  // cycles: a stream round of a lone workgroup ~320, a state of a full chip ~60 SIMD-cycles over 1024 SIMDs
  // (first of all: a full batch ends here, before a candidate is looked at)
  if (mode < 0 && longest * 320.0 < 1.3 * sum_states * 60.0 / 1024.0) return;
  for (int64_t u = 0; u < U; ++u) {
    const int64_t S = state_off[u + 1] - state_off[u];
    if (on_level[u] || S <= 0) continue;
    if (S >= 2 * (int64_t)chain_segment_states(S, arc_off[u + 1] - arc_off[u], seg) && lr[u].max_span <= kChainMaxSpan) cand.push_back(u);
  }
  if (cand.empty()) return;
  std::sort(cand.begin(), cand.end(), [&](int64_t x, int64_t y) {
    const int64_t sx = state_off[x + 1] - state_off[x], sy = state_off[y + 1] - state_off[y];
    return sx != sy ? sx > sy : x < y;
  });
  size_t take = cand.size();
  if (mode < 0) {
    // All the candidates or none.  (Round 3 first took the lattices above the cheapest size threshold and left the
    // others to three classes of streams beside the segment path: 0.90 ms on the C5/8 shard when the four HIP
    // streams got a hardware queue each, 1.05 - 1.4 ms when the runtime's queue lottery made one of the classes
    // start 200 us late or starved it.  With every candidate in segments the streams that are left are a few
    // hundred states long and it does not matter when they run: 0.94 ms, every time.)
    const double base = plan_cycles(U, state_off, arc_off, lr, o, on_level);
    std::vector<uint8_t> skip(on_level, on_level + U);
    double records = 0, span = 2, segs_max = 1;
    for (int64_t u : cand) {
      const int64_t S = state_off[u + 1] - state_off[u], A = arc_off[u + 1] - arc_off[u];
      skip[(size_t)u] = 1;
      records += (double)(A + S);
      span = std::max(span, (double)lr[u].max_span);
      segs_max = std::max(segs_max, (double)(S / chain_segment_states(S, A, seg)));
    }
    int cc = 2;
    while (cc < span) cc *= 2;
    // two walks of a segment (transfer, sweep: ~350 cycles a record for a wave on its own), the two-level combine
    // (~1 200 a step: a lattice's groups of 16 segments, then 2 x 16 steps), six launches (measured: one lattice of
    // 5 000 or of 20 000 states 0.18 / 0.20 ms = 440 000 / 480 000 cycles); the walks' work: ~2 SIMD-cycles per record
    // and lane, c lanes for the transfer, 2 for the sweep
    const double latency = 2.0 * 4.0 * (double)seg * 350.0 + (segs_max / kChainGroup + 2.0 * kChainGroup) * 1200.0 + 60000.0;
    const double work = records * (cc + 2) * 2.0 / 1024.0;
    const double total = std::max(plan_cycles(U, state_off, arc_off, lr, o, skip.data()), latency) + work;
    if (total >= 0.75 * base) return;
  }
  if (take == 0) return;
  int cc = 2;
  for (size_t i = 0; i < take; ++i)
    while (cc < lr[cand[i]].max_span) cc *= 2;
  std::vector<int64_t> chosen(cand.begin(), cand.begin() + (long)take);
  std::sort(chosen.begin(), chosen.end());
  uint64_t rec = 0;
  int64_t nseg = 0;
  for (int64_t u : chosen) {
    const int64_t S = state_off[u + 1] - state_off[u], A = arc_off[u + 1] - arc_off[u];
    ChainLattice cl;
    cl.lattice = (int32_t)u;
    cl.num_states = (int32_t)S;
    cl.seg_begin = (int32_t)nseg;
    cl.num_segs = (int32_t)(S / std::max(chain_segment_states(S, A, seg), 2 * cc));
    if (cl.num_segs < 2) continue;
    cl.span = std::max(1, lr[u].max_span);
    cl.grp_begin = out->empty() ? 0 : out->back().grp_begin + out->back().num_grps;
    cl.num_grps = (cl.num_segs + kChainGroup - 1) / kChainGroup;
    cl.pad = 0;
    cl.rec_f = (uint32_t)rec;
    rec += (uint64_t)(A + 2 * S + 2);  // arcs, a constant term or a zero per state, final weights
    cl.rec_b = (uint32_t)rec;
    rec += (uint64_t)(A + S + 2);
    nseg += cl.num_segs;
    if (rec >= ((uint64_t)1 << 31) || nseg >= ((int64_t)1 << 24)) {  // 32-bit record numbers: the rest keeps its streams
      break;
    }
    out->push_back(cl);
    (*on_chain)[(size_t)u] = 1;
    *num_records = (size_t)rec + 32;  // (the walks load a batch of records past a segment's last)
  }
  *seg_len_out = seg;
  *c_out = cc;
}

// the lattices on the level-synchronous path (klat_level.hip), on the context's stream
static hipError_t launch_level(klat_batch* b, double* fwd, double* bwd, const ScaleParams& sp, bool tropical) {
  LevelSweepArgs la;
  la.lattices = b->lv_lattices.p;
  la.state_off = b->state_off.p;
  la.arc_off = b->arc_off.p;
  la.final_graph = b->final_graph.p;
  la.final_acoustic = b->final_acoustic.p;
  la.records = b->lv_records.p;
  la.stream_off = b->lv_stream_off.p;
  la.stream = b->lv_stream.p;
  la.alpha = fwd;
  la.beta = bwd;
  la.sp = sp;
  return launch_level_sweep(la, b->num_level, b->level_max_states, tropical, b->ctx->stream);
}

// kernel times of step `step` of the context (its events are in the ring)
static int read_timings(klat_ctx* c, int64_t step, klat_timings* t) {
  hipEvent_t* ev = c->evring[step % klat_ctx::kTimingSlots];
  KLAT_HIP(hipEventSynchronize(ev[3]));
  KLAT_HIP(hipEventElapsedTime(&t->sweep_ms, ev[0], ev[1]));
  KLAT_HIP(hipEventElapsedTime(&t->totals_ms, ev[1], ev[2]));
  KLAT_HIP(hipEventElapsedTime(&t->posterior_ms, ev[2], ev[3]));
  KLAT_HIP(hipEventElapsedTime(&t->total_ms, ev[0], ev[3]));
  return KLAT_OK;
}

extern "C" {

static ScaleParams make_scale_params(const klat_scales& s);

const char* klat_last_error(void) { return g_last_error.c_str(); }
const char* klat_version(void) { return "klat 0.1 (gfx950)"; }

int klat_device_count(int* count) {
  if (!count) return set_error(KLAT_ERR_INVALID_ARG, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return set_error(KLAT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return KLAT_OK;
}

int klat_host_alloc(void** ptr, size_t bytes) {
  if (!ptr) return set_error(KLAT_ERR_INVALID_ARG, "ptr is null");
  *ptr = nullptr;
  KLAT_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return KLAT_OK;
}

void klat_host_free(void* ptr) {
  if (ptr) (void)hipHostFree(ptr);
}

static int ctx_create(int device, klat_ctx** out, bool side_streams);
int klat_ctx_create(int device, klat_ctx** out) { return ctx_create(device, out, true); }

static int ctx_create(int device, klat_ctx** out, bool side_streams) {
  // The pieces of a skewed batch's sweep run on four streams at once (run_side_by_side).  The HIP runtime maps a
  // process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) when it starts.  The default is the right
  // number for a process whose only streams are this context's (measured on the C5/8 shard, segment path on: 1.04 ms
  // with 4 queues, 1.24 - 1.39 ms with 8); a host with streams of its own (PyTorch holds one) exports
  // GPU_MAX_HW_QUEUES=8 before the runtime starts, as bench.py does.
  if (!out) return set_error(KLAT_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(KLAT_ERR_NO_DEVICE,
                     "no HIP device available: the lattice forward-backward path has no CPU fallback");
  if (device < 0 || device >= n) return set_error(KLAT_ERR_INVALID_ARG, "device index out of range");
  KLAT_HIP(hipSetDevice(device));
  klat_ctx* c = new (std::nothrow) klat_ctx();
  if (!c) return set_error(KLAT_ERR_OOM, "host allocation failed");
  c->device = device;
  {  // what the context's pool may keep: a quarter of the device's memory
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) c->pool.max_bytes = std::min(DevPool::kMaxBytes, total_b / 4);
  }
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipStreamCreate");
  }
  // The side streams of a skewed batch's sweep, created WITH the context's stream: the runtime deals its streams onto
  // the hardware queues in the order they are created, so four streams made together sit on four queues whatever else
  // the process creates in between.  (Made on first use they landed wherever the count stood then: after the bench's
  // host-to-host leg had created and destroyed six streams, the C5/8 shard's sweep took 1.07 - 1.15 ms instead of 0.94.)
  // (The helper contexts of klat_forward_backward_host go without: the call's three main streams should sit on three queues.)
  if (side_streams && hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess) (void)hipGetLastError(), c->fork_ev = nullptr;
  for (int i = 0; c->fork_ev && i < klat_ctx::kSideStreams; ++i)
    if (hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      klat_ctx_destroy(c);
      return set_error(KLAT_ERR_HIP, "side stream creation failed");
    }
  // (the timing events of the ring are created when a step first uses their slot)
  {
    std::vector<LnTabEntry> tab((size_t)kLnTabSize);
    fill_ln_table(tab.data());
    int rc = c->ln_table.upload(tab.data(), tab.size(), c->stream);
    if (rc == KLAT_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = set_error(KLAT_ERR_HIP, "ln table upload");
    if (rc != KLAT_OK) {
      klat_ctx_destroy(c);
      return rc;
    }
  }
  *out = c;
  return KLAT_OK;
}

static void ctx_release(klat_ctx* c) {
  if (c->refs.fetch_sub(1) != 1) return;  // batches (or a communicator) still use its stream
  for (klat_ctx*& h : c->pipe)
    if (h) ctx_release(h), h = nullptr;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < klat_ctx::kTimingSlots * 4; ++i)
    if (c->evring[i / 4][i % 4]) (void)hipEventDestroy(c->evring[i / 4][i % 4]);
  for (int i = 0; i < klat_ctx::kSideStreams; ++i) {
    if (c->side[i]) (void)hipStreamSynchronize(c->side[i]), (void)hipStreamDestroy(c->side[i]);
    if (c->join_ev[i]) (void)hipEventDestroy(c->join_ev[i]);
  }
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->pinned) (void)hipHostFree(c->pinned);
  c->pool.clear();
  delete c;
}

void klat_ctx_destroy(klat_ctx* c) {
  if (!c) return;
  ctx_release(c);
}

// for libklat_dist: a communicator keeps its context's stream alive the same way
void klat_ctx_retain(klat_ctx* c) {
  if (c) c->refs.fetch_add(1);
}

int klat_ctx_sync(klat_ctx* c) {
  if (!c) return set_error(KLAT_ERR_INVALID_ARG, "ctx is null");
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  KLAT_HIP(hipStreamSynchronize(c->stream));
  return KLAT_OK;
}

void* klat_ctx_stream(klat_ctx* c) { return c ? (void*)c->stream : nullptr; }

int klat_ctx_trim(klat_ctx* c, int64_t* released_bytes) {
  if (!c) return set_error(KLAT_ERR_INVALID_ARG, "ctx is null");
  KLAT_HIP(hipSetDevice(c->device));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  if (released_bytes) *released_bytes = (int64_t)c->pool.cached;
  c->pool.clear();
  return KLAT_OK;
}

int klat_batch_create(klat_ctx* c, const klat_batch_desc* d, const klat_pack_options* opt,
                      klat_batch** out) {
  if (!c || !d || !out) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  const bool host_prof = std::getenv("KLAT_PACK_PROFILE") != nullptr;
  auto t_last = t_begin;
  std::string host_stages;
  auto lap = [&](const char* what) {
    if (!host_prof) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    std::snprintf(buf, sizeof(buf), " %s %.3f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    host_stages += buf;
    t_last = now;
  };
  klat_pack_options o = {0, 0, 0, nullptr, 0};
  if (opt) o = *opt;
  // The streams are built on the device from the CSR arrays (klat_pack_device.hip); the host
  // packer (same streams, bit for bit) remains behind klat_pack_host and KLAT_HOST_PACK=1.
  const bool host_pack = std::getenv("KLAT_HOST_PACK") != nullptr;
  PackedBatch P;
  std::string err;
  int rc = KLAT_OK;
  const int64_t U = d->num_lattices;
  if (U < 0 || (U > 0 && (!d->state_off || !d->arc_off))) return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc: null offsets");
  if (U > 0 && (d->state_off[0] != 0 || d->arc_off[0] != 0))
    return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc: state_off[0] and arc_off[0] must be 0");
  for (int64_t u = 0; u < U; ++u)
    if (d->state_off[u + 1] < d->state_off[u] || d->arc_off[u + 1] < d->arc_off[u])
      return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc: offsets of lattice " + std::to_string(u) + " decrease");
  const int64_t S_tot = U ? d->state_off[U] : 0, A_tot = U ? d->arc_off[U] : 0;
  if (host_pack) o.fuse_scales = nullptr;
  if (host_pack) {
    // the host packer reads the arc and state arrays on the host
    for (const void* ptr : {(const void*)d->src, (const void*)d->dst, (const void*)d->ilabel, (const void*)d->graph,
                            (const void*)d->acoustic, (const void*)d->final_graph, (const void*)d->final_acoustic}) {
      hipPointerAttribute_t attr;
      if (ptr && hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.type == hipMemoryTypeDevice)
        return set_error(KLAT_ERR_INVALID_ARG, "KLAT_HOST_PACK=1 needs host arrays in klat_batch_desc (got a device pointer)");
    }
    (void)hipGetLastError();  // hipPointerGetAttributes on plain host memory reports an error: not ours
    rc = pack_batch(*d, o, &P, &err);
    if (rc != KLAT_OK) return set_error(rc, err);
  } else {
    if (A_tot > 0 && (!d->src || !d->dst || !d->graph || !d->acoustic || !d->ilabel))
      return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc: null arc arrays");
    if (S_tot > 0 && (!d->final_graph || !d->final_acoustic))
      return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc: null final-weight arrays");
    if (S_tot + 2 * U >= (int64_t)1 << 31 || A_tot >= (int64_t)1 << 31)
      return set_error(KLAT_ERR_INVALID_ARG, "batch too large for 32-bit stream positions: split it");
    for (int64_t u = 0; u < U; ++u)
      if (d->state_off[u + 1] - d->state_off[u] >= (int64_t)kRelMask)
        return set_error(KLAT_ERR_INVALID_ARG, "lattice " + std::to_string(u) + ": too many states");
    if (o.lanes_per_lattice < 0 || o.lanes_per_lattice > 64 || (o.lanes_per_lattice & (o.lanes_per_lattice - 1)))
      return set_error(KLAT_ERR_INVALID_ARG, "lanes_per_lattice must be a power of two in [1,64]");
    if (o.ring_slots != 0 && (o.ring_slots < 8 || o.ring_slots > 1024 || (o.ring_slots & (o.ring_slots - 1))))
      return set_error(KLAT_ERR_INVALID_ARG, "ring_slots must be a power of two in [8,1024]");
  }

  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  klat_batch* b = new (std::nothrow) klat_batch();
  if (!b) return set_error(KLAT_ERR_OOM, "host allocation failed");
  b->ctx = c;
  c->refs.fetch_add(1);
  b->U = U;
  b->S = S_tot;
  b->A = A_tot;
  b->h_state_off.assign(d->state_off, d->state_off + b->U + 1);
  b->h_arc_off.assign(d->arc_off, d->arc_off + b->U + 1);

#define UP(buf, ptr, n) UPP(buf, ptr, n, 0)
#define UPP(buf, ptr, n, pad)                                        \
  do {                                                               \
    rc = b->buf.upload(ptr, (size_t)(n), c->stream, pad);            \
    if (rc != KLAT_OK) {                                             \
      klat_batch_destroy(b);                                         \
      return rc;                                                     \
    }                                                                \
  } while (0)
#define AL(buf, n)                                                   \
  do {                                                               \
    rc = b->buf.alloc((size_t)(n));                                  \
    if (rc != KLAT_OK) {                                             \
      klat_batch_destroy(b);                                         \
      return rc;                                                     \
    }                                                                \
  } while (0)
#define KLAT_HIP_B(expr)                                             \
  do {                                                               \
    hipError_t _e = (expr);                                          \
    if (_e != hipSuccess) {                                          \
      klat_batch_destroy(b);                                         \
      return hip_fail(_e, #expr);                                    \
    }                                                                \
  } while (0)
#define FAIL(code, msg)                                              \
  do {                                                               \
    klat_batch_destroy(b);                                           \
    return set_error(code, msg);                                     \
  } while (0)
  UP(state_off, d->state_off, b->U + 1);
  UP(arc_off, d->arc_off, b->U + 1);
  // the arc and state arrays: copied (host arrays: uploaded), or used where they are (klat_pack_options.borrow_device_arrays)
  auto on_device = [&](const void* ptr) {
    hipPointerAttribute_t attr;
    const bool dev = ptr && ((uintptr_t)ptr & 15) == 0 && hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // (hipPointerGetAttributes on plain host memory reports an error: not ours)
    return dev;
  };
#define UPB(buf, ptr, n, pad)                                        \
  do {                                                               \
    if (o.borrow_device_arrays && !host_pack && on_device(ptr))      \
      b->buf.borrow(ptr, (size_t)(n));                               \
    else                                                             \
      UPP(buf, ptr, n, pad);                                         \
  } while (0)
  UPB(src, d->src, b->A, 4);
  UPB(dst, d->dst, b->A, 4);
  UPB(ilabel, d->ilabel, b->A, 4);
  if (d->olabel) UPB(olabel, d->olabel, b->A, 0);
  UPB(graph, d->graph, b->A, 4);
  UPB(acoustic, d->acoustic, b->A, 4);
  if (d->state_time) UPB(state_time, d->state_time, b->S, 0);
  UPB(final_graph, d->final_graph, b->S, 0);
  UPB(final_acoustic, d->final_acoustic, b->S, 0);
#undef UPB
  if (host_pack) {
    UP(sw, P.sw.data(), P.sw.size());
    UP(sg, P.sg.data(), P.sg.size());
    UP(sa, P.sa.data(), P.sa.size());
  } else {
    // pass 1 on the device: validation, the tables the fill pass reads, rounds per lattice for every lane count.
    // Lattices with short arcs take the lean passes (klat_pack_lean.hip; KLAT_PACK_LEAN=0: none does), the others
    // the general ones; which lattice is which is known for sure only when the lean count pass has read its arcs.
    DevBuf<int32_t> cnt, row, r0f, r0b, flist, in_arc, out_arc, status;
    DevBuf<LatticeRounds> d_rounds;
    DevBuf<int8_t> d_lat_g, d_pred;
    DevBuf<PackEntry> d_entries;
    DevBuf<LeanEntry> d_lentries;
    DevBuf<uint32_t> meta;
    DevBuf<int32_t> tpf, tpb;
    DevBuf<uint8_t> d_lean_ok, d_only, d_only2, d_only3, d_reround;
    const int32_t status0[4] = {0, -1, 1, 0};
    if ((rc = row.alloc((size_t)(S_tot + 2 * U))) || (rc = r0f.alloc((size_t)(S_tot + 2 * U))) || (rc = r0b.alloc((size_t)(S_tot + 2 * U))) ||
        (rc = flist.alloc((size_t)S_tot)) || (rc = d_rounds.alloc((size_t)U)) || (rc = status.upload(status0, 4, c->stream))) {
      klat_batch_destroy(b);
      return rc;
    }
    bool general_tables = false;  // cnt, in_arc, out_arc: only when some lattice takes the general passes
    auto need_general_tables = [&]() -> int {
      if (general_tables) return KLAT_OK;
      int r = KLAT_OK;
      if ((r = cnt.alloc((size_t)(S_tot + 2 * U))) || (r = in_arc.alloc((size_t)A_tot)) || (r = out_arc.alloc((size_t)A_tot))) return r;
      general_tables = true;
      return KLAT_OK;
    };
    PackDevArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.state_off = b->state_off.p;
    pa.arc_off = b->arc_off.p;
    pa.src = b->src.p;
    pa.dst = b->dst.p;
    pa.ilabel = b->ilabel.p;
    pa.graph = b->graph.p;
    pa.acoustic = b->acoustic.p;
    pa.final_graph = b->final_graph.p;
    pa.final_acoustic = b->final_acoustic.p;
    pa.row = row.p;
    pa.r0f = r0f.p;
    pa.r0b = r0b.p;
    pa.flist = flist.p;
    pa.lat_rounds = d_rounds.p;
    pa.status = status.p;
    pa.num_arcs = A_tot;
    // where the count pass's verdicts come back: page-locked memory of the context when no other call holds it
    std::unique_lock<std::mutex> pinned_lock(c->pinned_mu, std::try_to_lock);
    std::vector<LatticeRounds> h_rounds_pageable;
    std::vector<uint8_t> lean_ok_pageable;
    LatticeRounds* h_rounds_p = nullptr;
    uint8_t* lean_ok_p = nullptr;
    {
      const size_t want = sizeof(LatticeRounds) * (size_t)U + (size_t)U + 64;
      if (pinned_lock.owns_lock() && c->pinned_bytes < want) {
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr, c->pinned_bytes = 0;
        void* ptr = nullptr;
        if (hipHostMalloc(&ptr, want + want / 2, hipHostMallocDefault) == hipSuccess)
          c->pinned = ptr, c->pinned_bytes = want + want / 2;
        else
          (void)hipGetLastError();
      }
      if (pinned_lock.owns_lock() && c->pinned_bytes >= want) {
        h_rounds_p = static_cast<LatticeRounds*>(c->pinned);
        lean_ok_p = reinterpret_cast<uint8_t*>(h_rounds_p + U);
        std::memset(lean_ok_p, 0, (size_t)U);
      } else {
        h_rounds_pageable.resize((size_t)U);
        lean_ok_pageable.assign((size_t)U, 0);
        h_rounds_p = h_rounds_pageable.data();
        lean_ok_p = lean_ok_pageable.data();
      }
    }
    struct {
      LatticeRounds* p;
      size_t n;
      LatticeRounds* data() const { return p; }
      size_t size() const { return n; }
    } h_rounds = {h_rounds_p, (size_t)U};
    uint8_t* const lean_ok = lean_ok_p;
    int32_t h_status[4] = {0, -1, 1, 0};
    const char* lean_env = std::getenv("KLAT_PACK_LEAN");
    bool lean_enabled = !(lean_env && std::atoi(lean_env) == 0);
    // Lattices that will be swept in segments (klat_chain.hip) need the general tables: the lean count pass would run
    // for nothing on them (the configs[4] mix: 1.4 + 1.3 ms of count passes instead of 1.3).  Which lattices those are
    // is decided from the rounds the count pass reports -- so it is PREDICTED here from rounds estimated off the offsets
    // (states x ceil(fanout / lanes), arcs assumed short enough), and the predicted ones take the general passes from the
    // start.  A wrong guess costs time only: a lattice cut after all is counted again by the general pass (`redo`
    // below), one not cut after all has taken the slower passes.  KLAT_PACK_LEAN=1: no prediction (every lattice lean).
    std::vector<uint8_t> chain_pred((size_t)U, 0);
    if (lean_enabled && o.lanes_per_lattice == 0 && !(lean_env && std::atoi(lean_env) > 0)) {
      std::vector<LatticeRounds> est((size_t)U);
      for (int64_t u = 0; u < U; ++u) {
        const int64_t S = d->state_off[u + 1] - d->state_off[u], A = d->arc_off[u + 1] - d->arc_off[u];
        LatticeRounds& r = est[(size_t)u];
        std::memset(&r, 0, sizeof(r));
        const int64_t fan = S > 0 ? (A + S - 1) / S : 0;
        for (int k = 0; k < kNumLaneChoices; ++k) r.fwd[k] = r.bwd[k] = (int32_t)std::min<int64_t>(0x7FFFFFFF, S * std::max<int64_t>(1, (fan + (1 << k) - 1) >> k));
        r.max_rel = r.max_span = kChainMaxSpan;
        r.max_terms = (int32_t)(std::min<int64_t>(fan, 64) | (std::min<int64_t>(fan, 64) << 16));
      }
      std::vector<uint8_t> none((size_t)U, 0);
      std::vector<ChainLattice> cl;
      int seg_len = 0, cc = 0;
      size_t nrec = 0;
      choose_chain_lattices(U, d->state_off, d->arc_off, est.data(), o, none.data(), &chain_pred, &cl, &seg_len, &cc, &nrec);
      // (most of the batch cut: the few lattices left are not worth launches of their own -- the configs[4] mix: 14 lean
      // jobs beside 298 general ones cost 0.2 ms of small launches -- and the whole batch takes the general passes)
      int64_t cut_arcs = 0;
      for (int64_t u = 0; u < U; ++u)
        if (chain_pred[(size_t)u]) cut_arcs += d->arc_off[u + 1] - d->arc_off[u];
      if (2 * cut_arcs > A_tot) lean_enabled = false;
    }
    const char* level_env = std::getenv("KLAT_LEVEL");
    const int level_mode = level_env == nullptr ? -1 : std::atoi(level_env);
    const bool plan_levels = o.lanes_per_lattice == 0 && level_mode != 0 && U > 0;
    // which count pass a lattice takes, as far as its sizes tell: the LDS class of its tables
    std::vector<uint8_t> known_general((size_t)U, lean_enabled ? 0 : 1);
    if (lean_enabled) known_general = chain_pred;
    std::vector<int8_t> pred;
    int lean_need[kPackLeanClasses] = {0, 0, 0};
    size_t num_known_general = lean_enabled ? 0 : (size_t)U, num_lean = 0;
    if (lean_enabled) {
      predict_lanes(U, d->state_off, d->arc_off, o, &pred);
      for (int64_t u = 0; u < U; ++u) {
        const int need = pack_lean_lds_bytes(d->state_off[u + 1] - d->state_off[u], d->arc_off[u + 1] - d->arc_off[u]);
        int k = 0;
        while (k < kPackLeanClasses && need > kPackLeanClassLds[k]) ++k;
        if (k == kPackLeanClasses || known_general[(size_t)u])
          known_general[(size_t)u] = 1, ++num_known_general;
        else
          lean_need[k] = std::max(lean_need[k], need), ++num_lean;
      }
    }
    // the general count pass over the lattices flagged in `only` (null: all): the launch of the narrowest LDS class per lattice
    auto general_count = [&](const std::vector<uint8_t>* only, DevBuf<uint8_t>* d_flags) -> hipError_t {
      int class_need[kPackCountClasses] = {0, 0, 0};
      bool any_global = false;
      for (int64_t u = 0; u < U; ++u) {
        if (only && !(*only)[(size_t)u]) continue;
        const int need = pack_count_lds_bytes(d->state_off[u + 1] - d->state_off[u], d->arc_off[u + 1] - d->arc_off[u]);
        int k = 0;
        while (k < kPackCountClasses && need > kPackCountClassLds[k]) ++k;
        if (k == kPackCountClasses)
          any_global = true;
        else
          class_need[k] = std::max(class_need[k], need);
      }
      if (need_general_tables() != KLAT_OK) return hipErrorOutOfMemory;
      pa.cnt = cnt.p;
      pa.in_arc = in_arc.p;
      pa.out_arc = out_arc.p;
      pa.only = nullptr;
      if (only) {
        if (d_flags->upload(only->data(), only->size(), c->stream) != KLAT_OK) return hipErrorOutOfMemory;
        pa.only = d_flags->p;
      }
      const hipError_t e = launch_pack_count(pa, (int)U, class_need, any_global, c->stream);
      pa.only = nullptr;
      return e;
    };
    DevBuf<unsigned long long> d_prof;
    if (std::getenv("KLAT_PACK_PROFILE") != nullptr) {
      if ((rc = d_prof.alloc(24))) {
        klat_batch_destroy(b);
        return rc;
      }
      KLAT_HIP_B(hipMemsetAsync(d_prof.p, 0, 24 * sizeof(unsigned long long), c->stream));
      pa.prof = d_prof.p;
    }
    // fused costs: the batch's largest finite cost decides the unit of the 16-bit residuals
    DevBuf<unsigned long long> d_maxcost;
    unsigned long long h_maxcost = 0;
    if (o.fuse_scales) {
      b->fused = true;
      b->fuse_scales = *o.fuse_scales;
      pa.fused = 1;
      pa.fsp = make_scale_params(b->fuse_scales);
      if ((rc = d_maxcost.upload(&h_maxcost, 1, c->stream))) {
        klat_batch_destroy(b);
        return rc;
      }
      pa.maxcost = d_maxcost.p;
    }
    hipEvent_t* pev = b->pack_ev;
    for (int i = 0; i < klat_batch::kPackEvents; ++i) KLAT_HIP_B(hipEventCreate(&pev[i]));
    KLAT_HIP_B(hipEventRecord(pev[0], c->stream));
    hipError_t he = hipSuccess;
    if (num_lean > 0) {
      if ((rc = meta.alloc((size_t)A_tot + 4)) || (rc = tpf.alloc((size_t)((S_tot >> 3) + 2 * U + 2))) || (rc = tpb.alloc((size_t)((S_tot >> 3) + 2 * U + 2))) ||
          (rc = d_lean_ok.alloc((size_t)U)) || (rc = d_pred.upload(pred.data(), pred.size(), c->stream))) {
        klat_batch_destroy(b);
        return rc;
      }
      KLAT_HIP_B(hipMemsetAsync(d_lean_ok.p, 0, (size_t)U, c->stream));
      pa.meta = meta.p;
      pa.tpf = tpf.p;
      pa.tpb = tpb.p;
      pa.lean_ok = d_lean_ok.p;
      pa.lat_g_pred = d_pred.p;
      pa.level_probe = plan_levels ? 1 : 0;
      // (the lean count pass reads every arc's costs on its way; with lattices it does not see, the pass of its own)
      pa.fold_maxcost = b->fused && num_known_general == 0 ? 1 : 0;
    }
    if (b->fused && !pa.fold_maxcost) KLAT_HIP_B(launch_pack_maxcost(pa, A_tot, S_tot, c->stream));
    // (the flags first: the lean count pass leaves the lattices alone that are known to take the general one)
    if (num_lean > 0 && num_known_general > 0) {
      if ((rc = d_only.upload(known_general.data(), known_general.size(), c->stream))) {
        klat_batch_destroy(b);
        return rc;
      }
      pa.lean_skip = d_only.p;
    }
    if (num_lean > 0) he = launch_pack_count_lean(pa, (int)U, lean_need, c->stream);
    if (he == hipSuccess && num_known_general > 0) he = general_count(num_lean > 0 ? &known_general : nullptr, &d_only);
    if (he == hipSuccess) he = hipEventRecord(pev[1], c->stream);
    if (he == hipSuccess && b->fused) he = hipMemcpyAsync(&h_maxcost, d_maxcost.p, sizeof(h_maxcost), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess && U > 0)
      he = hipMemcpyAsync(h_rounds.data(), d_rounds.p, sizeof(LatticeRounds) * h_rounds.size(), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(h_status, status.p, sizeof(h_status), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess && num_lean > 0) he = hipMemcpyAsync(lean_ok, d_lean_ok.p, (size_t)U, hipMemcpyDeviceToHost, c->stream);
    // The level planner (below) runs behind the general count pass at once when every lattice takes that pass: its answer
    // comes back in the same wait.  With lean lattices it waits for their verdict (it reads the general tables).
    std::vector<uint8_t> on_level((size_t)U, 0);
    DevBuf<int32_t> order, lvoff, depth;  // level planning's tables, until the level streams are written
    DevBuf<int32_t> lv_counters, lv_records;
    std::vector<int32_t> h_depth((size_t)U, -1), h_records((size_t)U, 0);
    auto level_plan = [&](const uint8_t* only) -> hipError_t {
      if (lv_counters.alloc((size_t)(S_tot + 2 * U)) || order.alloc((size_t)S_tot) || lvoff.alloc((size_t)(S_tot + 2 * U)) || depth.alloc((size_t)U) ||
          lv_records.alloc((size_t)U))
        return hipErrorOutOfMemory;
      LevelPlanArgs lp;
      lp.state_off = b->state_off.p;
      lp.arc_off = b->arc_off.p;
      lp.src = b->src.p;
      lp.dst = b->dst.p;
      lp.cnt = cnt.p;
      lp.counters = lv_counters.p;
      lp.order = order.p;
      lp.level_off = lvoff.p;
      lp.out_begin = row.p;
      lp.depth = depth.p;
      lp.records = lv_records.p;
      lp.status = status.p;
      lp.only = only;
      hipError_t e = hipEventRecord(pev[2], c->stream);
      if (e == hipSuccess) e = launch_level_plan(lp, (int)U, c->stream);
      if (e == hipSuccess) e = hipEventRecord(pev[3], c->stream);
      b->pack_level_timed = true;
      if (e == hipSuccess) e = hipMemcpyAsync(h_depth.data(), depth.p, sizeof(int32_t) * (size_t)U, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(h_records.data(), lv_records.p, sizeof(int32_t) * (size_t)U, hipMemcpyDeviceToHost, c->stream);
      return e;
    };
    if (he == hipSuccess && plan_levels && num_lean == 0) he = level_plan(nullptr);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) {
      klat_batch_destroy(b);
      return hip_fail(he, "stream packing (count pass)");
    }
    lap("uploads+count+wait");
    if (pa.prof) {
      unsigned long long h[8];
      KLAT_HIP_B(hipMemcpy(h, d_prof.p, sizeof(h), hipMemcpyDeviceToHost));
      const double n = (double)std::max<int64_t>(U, 1);
      std::fprintf(stderr, "klat pack_count phases, cycles per lattice: clear %.0f  arcs %.0f  states %.0f  scatter %.0f  buckets %.0f  in_arc out %.0f  out-arcs %.0f  tail %.0f\n",
                   h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n);
    }
    auto fail_status = [&]() {
      return set_error(h_status[0] == 2 ? KLAT_ERR_BAD_TOPOLOGY : KLAT_ERR_INVALID_ARG,
                       "lattice " + std::to_string(h_status[1]) +
                           (h_status[0] == 2 ? ": arc with dst <= src or out of range (lattice is not topologically sorted)"
                                             : ": malformed CSR (state ids out of range or arcs not grouped by source)"));
    };
    if (h_status[0] != 0) {
      klat_batch_destroy(b);
      return fail_status();
    }
    // the lattices the lean count pass has left alone (a long arc, a candidate of the level path ...): the general pass now,
    // and the level planner over everything that has the general tables
    if (num_lean > 0) {
      std::vector<uint8_t> unfit((size_t)U, 0), have_general(known_general);
      size_t num_unfit = 0;
      for (int64_t u = 0; u < U; ++u)
        if (!known_general[(size_t)u] && !lean_ok[(size_t)u]) unfit[(size_t)u] = 1, have_general[(size_t)u] = 1, ++num_unfit;
      if (num_unfit > 0 || (plan_levels && num_known_general > 0)) {
        if (num_unfit > 0) he = general_count(&unfit, &d_only2);
        if (he == hipSuccess && num_unfit > 0)
          he = hipMemcpyAsync(h_rounds.data(), d_rounds.p, sizeof(LatticeRounds) * h_rounds.size(), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(h_status, status.p, sizeof(h_status), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess && plan_levels) {
          if ((rc = d_only3.upload(have_general.data(), have_general.size(), c->stream))) {
            klat_batch_destroy(b);
            return rc;
          }
          he = level_plan(d_only3.p);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) {
          klat_batch_destroy(b);
          return hip_fail(he, "stream packing (count pass of the lattices with long arcs)");
        }
        if (h_status[0] != 0) {
          klat_batch_destroy(b);
          return fail_status();
        }
      }
    }
    lap("verdicts");
    // Lattices whose topological depth is far below their states take the level-synchronous path
    // (klat_level.hip) instead of a stream: Kahn's levels on the device, the choice on the host.
    // Only with the automatic layout (a caller that fixes the lanes asks for streams).
    int64_t lv_blocks = 0;
    // KLAT_LEVEL=0 keeps every lattice on the streams, KLAT_LEVEL=1 sends every lattice that qualifies
    // to the level path; by default the level path takes them when they fit the chip in ONE round of
    // waves (its wave is faster than a stream for a lone lattice, and slower per lattice in bulk).
    if (plan_levels) {
      DevBuf<int32_t>& records = lv_records;
      std::vector<int32_t> lv;
      std::vector<int64_t> lv_first;  // first block of every level stream
      for (int64_t u = 0; u < U; ++u)
        if (h_depth[(size_t)u] > 0) {
          on_level[(size_t)u] = 1;
          lv.push_back((int32_t)u);
          b->level_max_states = std::max<int32_t>(b->level_max_states, (int32_t)(d->state_off[u + 1] - d->state_off[u]));
          lv_first.push_back(lv_blocks);
          lv_blocks += (level_stream_dwords(h_records[(size_t)u], d->state_off[u + 1] - d->state_off[u], d->arc_off[u + 1] - d->arc_off[u]) +
                        kLevelBlockDwords - 1) / kLevelBlockDwords;
        }
      if (!lv.empty() && level_mode < 0) {
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const size_t per_cu = (size_t)160 * 1024 / level_sweep_lds_bytes(b->level_max_states);
        if (lv.size() > per_cu * (size_t)cus) {  // more than one round of waves: the streams are the faster path
          lv.clear();
          std::fill(on_level.begin(), on_level.end(), 0);
          b->level_max_states = 0;
        }
      }
      if (!lv.empty()) {  // keep what the level sweep reads; the count pass's tables go with it
        b->num_level = (int32_t)lv.size();
        if ((rc = b->lv_lattices.upload(lv.data(), lv.size(), c->stream)) || (rc = b->lv_stream_off.upload(lv_first.data(), lv_first.size(), c->stream))) {
          klat_batch_destroy(b);
          return rc;
        }
        KLAT_HIP(hipStreamSynchronize(c->stream));
        b->lv_records.take(records);
      }
    }
    lap("levels");
    // Long lattices of a batch that waits for them are swept in segments (klat_chain.hip): the choice, their
    // segments.  (Their arc records are built when a sweep first needs them, for the scales it is called with:
    // ensure_chain_records.)  Only with the automatic layout.
    std::vector<uint8_t> on_chain((size_t)U, 0), skip(on_level);
    size_t nrec_chain = 0;
    {
      std::vector<ChainLattice> cl;
      int seg_len = 0, cc = 0;
      size_t nrec = 0;
      if (o.lanes_per_lattice == 0)
        choose_chain_lattices(U, d->state_off, d->arc_off, h_rounds.data(), o, on_level.data(), &on_chain, &cl, &seg_len, &cc, &nrec);
      if (!cl.empty()) {
        const size_t nseg = (size_t)cl.back().seg_begin + (size_t)cl.back().num_segs;
        const size_t ngrp = (size_t)cl.back().grp_begin + (size_t)cl.back().num_grps;
        std::vector<ChainGroup> groups;
        for (size_t i = 0; i < cl.size(); ++i)
          for (int32_t g = 0; g < cl[i].num_grps; ++g)
            groups.push_back(ChainGroup{(int32_t)i, cl[i].seg_begin + g * kChainGroup, std::min(kChainGroup, cl[i].num_segs - g * kChainGroup), 0});
        const int32_t zero[5] = {0, 0, 0, 0, 0};
        if ((rc = b->ch_lats.upload(cl.data(), cl.size(), c->stream)) || (rc = b->ch_segs.alloc(nseg)) ||
            (rc = b->ch_qm.alloc(nseg * cc * cc)) || (rc = b->ch_qe.alloc(nseg * cc)) || (rc = b->ch_qfm.alloc(nseg * cc)) ||
            (rc = b->ch_qfe.alloc(nseg * cc)) || (rc = b->ch_bfm.alloc(nseg * cc)) || (rc = b->ch_bfe.alloc(nseg)) ||
            (rc = b->ch_bbm.alloc(nseg * cc)) || (rc = b->ch_bbe.alloc(nseg)) || (rc = b->ch_status.upload(zero, 5, c->stream)) ||
            (rc = b->ch_grps.upload(groups.data(), groups.size(), c->stream)) || (rc = b->ch_gm.alloc(ngrp * cc * cc)) || (rc = b->ch_ge.alloc(ngrp * cc)) ||
            (rc = b->ch_gfm.alloc(ngrp * cc)) || (rc = b->ch_gfe.alloc(ngrp * cc)) || (rc = b->ch_gbfm.alloc(ngrp * cc)) || (rc = b->ch_gbfe.alloc(ngrp)) ||
            (rc = b->ch_gbbm.alloc(ngrp * cc)) || (rc = b->ch_gbbe.alloc(ngrp))) {
          klat_batch_destroy(b);
          return rc;
        }
        b->num_chain = (int32_t)cl.size();
        nrec_chain = nrec;
        b->chain_segs = (int32_t)nseg;
        b->chain_grps = (int32_t)ngrp;
        b->chain_c = cc;
        b->chain_seg_len = seg_len;
        for (size_t u = 0; u < (size_t)U; ++u) skip[u] = on_level[u] | on_chain[u];
        KLAT_HIP_B(hipStreamSynchronize(c->stream));  // (cl is read by the upload)
      }
    }
    lap("chain choice");
    // the plan: lanes per lattice, layout classes, jobs
    std::vector<int8_t> lat_g;
    plan_layout(U, d->state_off, d->arc_off, h_rounds.data(), o, &lat_g, skip.data());
    // (the lattices swept in segments keep their streams -- for klat_viterbi, and for the sweeps that find a lattice's
    // values outside the segment path's arithmetic -- all with four lanes, as ONE layout class: the launches that
    // stand by behind every sweep cost ~20 us apiece, and three classes were six of them)
    for (int64_t u = 0; u < U; ++u)
      if (on_chain[(size_t)u]) lat_g[(size_t)u] = 2;
    std::vector<std::vector<int64_t>> job_lats;
    int64_t chunk_total = 0;
    lap("level");
    rc = build_jobs(U, d->state_off, h_rounds.data(), lat_g, o, &P, &job_lats, &chunk_total, &err, b->num_chain > 0 ? on_chain.data() : nullptr);
    if (rc != KLAT_OK) FAIL(rc, err);
    lap("plan+jobs");
    // Which fill pass writes a job: the lean one when all its lattices have the lean tables (and none is swept in segments:
    // klat_viterbi builds those lattices' cost records from the general tables), else the general one -- whose tables the
    // job's other lattices then need too, for the job's direction (bit 0 of `redo`: forward, bit 1: backward).
    std::vector<uint8_t> job_general(P.jobs.size(), 0), redo((size_t)U, 0), reround((size_t)U, 0);
    size_t num_general_jobs = 0, num_redo = 0, num_reround = 0;
    for (size_t j = 0; j < P.jobs.size(); ++j) {
      bool g = false;
      for (int64_t u : job_lats[j]) g = g || !lean_ok[(size_t)u] || on_chain[(size_t)u];
      if (!g) continue;
      job_general[j] = 1, ++num_general_jobs;
      for (int64_t u : job_lats[j]) redo[(size_t)u] |= (uint8_t)(1 << P.jobs[j].dir);
    }
    {
      std::vector<uint8_t> recount((size_t)U, 0);  // lattices with lean tables that need the general ones after all
      size_t num_recount = 0;
      for (int64_t u = 0; u < U; ++u) {
        if (redo[(size_t)u]) ++num_redo;
        if (redo[(size_t)u] && lean_ok[(size_t)u]) recount[(size_t)u] = 1, ++num_recount;
        if (lean_ok[(size_t)u] && redo[(size_t)u] != 3 && lat_g[(size_t)u] >= 0 && lat_g[(size_t)u] != pred[(size_t)u]) reround[(size_t)u] = 1, ++num_reround;
      }
      if (num_recount > 0) {
        he = general_count(&recount, &d_only2);
        if (he != hipSuccess) {
          klat_batch_destroy(b);
          return hip_fail(he, "stream packing (count pass of the lattices that share a job with long arcs)");
        }
      }
    }
    lap("job kinds");
    if (host_prof)
      std::fprintf(stderr, "klat_batch_create lattices: %zu lean candidates, %zu general by size; jobs %zu of which %zu general; lattices redone by the general passes %zu, rounds redone for other lanes %zu\n",
                   num_lean, num_known_general, P.jobs.size(), num_general_jobs, num_redo, num_reround);
    // pass 2: the rounds of every state with its lattice's lanes; pass 3: one workgroup per (job, run of tiles)
    std::vector<PackEntry> entries;
    // The lean jobs' entries by lanes per lattice; within a lane count a group's forward and backward job side by side --
    // forward run i next to the backward run over the same states -- so that the group's arcs come from HBM once and
    // from the caches the second time (one launch); jobs without such a partner by direction (a launch each).
    std::vector<LeanEntry> lentries;
    int lean_count[3 * kNumLaneChoices] = {0};
    {
      const char* pair_env = std::getenv("KLAT_FILL_PAIR");
      const bool pairing = !(pair_env && std::atoi(pair_env) == 0);
      std::vector<int32_t> partner(P.jobs.size(), -1);
      if (pairing) {
        std::vector<int32_t> bwd_of((size_t)U, -1);
        for (size_t j = 0; j < P.jobs.size(); ++j)
          if (!job_general[j] && P.jobs[j].dir == 1 && !job_lats[j].empty()) bwd_of[(size_t)job_lats[j][0]] = (int32_t)j;
        for (size_t j = 0; j < P.jobs.size(); ++j) {
          if (job_general[j] || P.jobs[j].dir != 0 || job_lats[j].empty()) continue;
          const int32_t k = bwd_of[(size_t)job_lats[j][0]];
          if (k >= 0 && job_lats[(size_t)k] == job_lats[j] && job_lanes(P.jobs[(size_t)k].tag) == job_lanes(P.jobs[j].tag)) partner[j] = k, partner[(size_t)k] = (int32_t)j;
        }
      }
      auto tiles_of = [&](size_t j) { return (P.jobs[j].nchunks * kChunkRounds + kLeanTileRounds - 1) / kLeanTileRounds; };
      auto runs_of = [&](size_t j) { return (tiles_of(j) + kLeanTilesPerGroup - 1) / kLeanTilesPerGroup; };
      auto push = [&](size_t j, int run, int slot) {
        const int ntiles = tiles_of(j), t = run * kLeanTilesPerGroup;
        lentries.push_back(LeanEntry{P.jobs[j].chunk_off, P.jobs[j].lane_tab, P.jobs[j].tag, P.jobs[j].dir, t, std::min(kLeanTilesPerGroup, ntiles - t), (int32_t)j,
                                     P.jobs[j].nchunks * kChunkRounds, 0});
        ++lean_count[slot];
      };
      for (int lg = 0; lg < kNumLaneChoices; ++lg)
        for (int kind = 0; kind < 3; ++kind)
          for (size_t j = 0; j < P.jobs.size(); ++j) {
            if (job_general[j] || job_lanes(P.jobs[j].tag) != (1 << lg)) continue;
            if (kind < 2) {
              if (partner[j] >= 0 || P.jobs[j].dir != kind) continue;
              for (int r = 0; r < runs_of(j); ++r) push(j, r, 3 * lg + kind);
            } else {
              if (partner[j] < 0 || P.jobs[j].dir != 0) continue;
              const size_t k = (size_t)partner[j];
              const int nf = runs_of(j), nb = runs_of(k);
              for (int r = 0; r < std::max(nf, nb); ++r) {
                if (r < nf) push(j, r, 3 * lg + 2);
                if (r < nb) push(k, nb - 1 - r, 3 * lg + 2);
              }
            }
          }
    }
    for (size_t j = 0; j < P.jobs.size(); ++j) {
      if (!job_general[j]) continue;
      const int ntiles = P.jobs[j].nchunks * kChunkRounds / kPackTileRounds;
      for (int t = 0; t < ntiles; t += kPackTilesPerGroup) entries.push_back(PackEntry{(int32_t)j, t, std::min(kPackTilesPerGroup, ntiles - t), 0});
    }
    lap("entries");
    const size_t nslots = (size_t)chunk_total * kChunkRounds;  // chunk_total counts 16-byte units
    AL(sw, nslots);
    AL(sg, nslots);
    if (b->fused) {
      AL(sl, nslots);
      // a cost below 2^E is a float (24 bits: an ulp of at most 2^(E-24)) plus a residual of at most half
      // an ulp, kept as a 16-bit integer in units of 2^(E-40)
      double maxc;
      std::memcpy(&maxc, &h_maxcost, sizeof(maxc));
      int e = 0;
      (void)std::frexp(std::max(maxc, 1.0), &e);
      b->lo_unit = std::ldexp(1.0, e - 40);
      pa.sl = b->sl.p;
      pa.lo_inv_unit = 1.0 / b->lo_unit;
    } else {
      AL(sa, nslots);
    }
    UP(jobs, P.jobs.data(), P.jobs.size());
    UP(lanes, P.lanes.data(), P.lanes.size());
    if ((rc = d_entries.upload(entries.data(), entries.size(), c->stream)) || (rc = d_lentries.upload(lentries.data(), lentries.size(), c->stream)) ||
        (rc = d_lat_g.upload(lat_g.data(), lat_g.size(), c->stream)) || (num_redo > 0 && num_lean > 0 && (rc = d_only.upload(redo.data(), redo.size(), c->stream))) ||
        (num_reround > 0 && (rc = d_reround.upload(reround.data(), reround.size(), c->stream)))) {
      klat_batch_destroy(b);
      return rc;
    }
    pa.lat_g = d_lat_g.p;
    pa.entries = d_entries.p;
    pa.lentries = d_lentries.p;
    pa.jobs = b->jobs.p;
    pa.lanes = b->lanes.p;
    pa.sw = b->sw.p;
    pa.sg = b->sg.p;
    pa.sa = b->sa.p;
    lap("allocs+uploads");
    he = hipEventRecord(pev[4], c->stream);
    if (he == hipSuccess && num_redo > 0) {
      pa.only = num_lean > 0 ? d_only.p : nullptr;  // (bit 0: the forward rounds, bit 1: the backward rounds)
      he = launch_pack_rounds(pa, (int)U, c->stream);
      pa.only = nullptr;
    }
    if (he == hipSuccess && num_reround > 0) {
      pa.reround = d_reround.p;
      he = launch_pack_rounds_lean(pa, (int)U, c->stream);
    }
    if (he == hipSuccess) he = hipEventRecord(pev[5], c->stream);
    int max_lpw = 1;
    for (const LayoutClass& lc : P.classes) max_lpw = std::max(max_lpw, 64 / lc.G);
    if (he == hipSuccess) he = launch_pack_fill(pa, (int)entries.size(), max_lpw, c->stream);
    if (he == hipSuccess) he = launch_pack_fill_lean(pa, lean_count, c->stream);
    if (he == hipSuccess) he = hipEventRecord(pev[6], c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(h_status, status.p, sizeof(h_status), hipMemcpyDeviceToHost, c->stream);
    // (while the fill pass runs: the posterior kernel's work items)
    {
      size_t n_items = 0;
      for (int64_t u = 0; u < U; ++u) n_items += (size_t)std::max<int64_t>(1, (d->arc_off[u + 1] - d->arc_off[u] + kPostItemArcs - 1) / kPostItemArcs);
      P.items.reserve(n_items);
      for (int64_t u = 0; u < U; ++u) append_post_items(&P.items, d->arc_off[u], d->arc_off[u + 1], u);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);  // scratch and entries go out of scope below
    if (he != hipSuccess) {
      klat_batch_destroy(b);
      return hip_fail(he, "stream packing (fill pass)");
    }
    lap("entries+rounds+fill+wait");
    b->info.lean_jobs = (int32_t)(P.jobs.size() - num_general_jobs);
    if (pa.prof && !lentries.empty()) {  // (a build with -DKLAT_FILL_PROFILE fills these)
      unsigned long long h[24];
      KLAT_HIP_B(hipMemcpy(h, d_prof.p, sizeof(h), hipMemcpyDeviceToHost));
      size_t nf = 0;
      for (const LeanEntry& le : lentries) nf += le.dir == 0 ? 1 : 0;
      for (int dir = 0; dir < 2 && h[8] + h[16] != 0; ++dir) {
        const double n = (double)std::max<size_t>(dir ? lentries.size() - nf : nf, 1);
        const unsigned long long* q = h + 8 + 8 * dir;
        std::fprintf(stderr, "klat pack_fill_lean %s, cycles per workgroup (%.0f): setup %.0f  window+init %.0f  barrier %.0f  arcs %.0f  positions %.0f  finals+barrier %.0f  store %.0f  barrier %.0f\n",
                     dir ? "backward" : "forward", n, q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / n, q[5] / n, q[6] / n, q[7] / n);
      }
    }
    P.far_terms = h_status[3];

    if (b->num_level > 0) {  // the level streams: the arcs as the level sweep consumes them
      if ((rc = b->lv_stream.alloc((size_t)lv_blocks * kLevelBlockDwords))) {
        klat_batch_destroy(b);
        return rc;
      }
      LevelBuildArgs lb;
      lb.lattices = b->lv_lattices.p;
      lb.state_off = b->state_off.p;
      lb.arc_off = b->arc_off.p;
      lb.src = b->src.p;
      lb.dst = b->dst.p;
      lb.ilabel = b->ilabel.p;
      lb.graph = b->graph.p;
      lb.acoustic = b->acoustic.p;
      lb.final_graph = b->final_graph.p;
      lb.final_acoustic = b->final_acoustic.p;
      lb.cnt = cnt.p;
      lb.in_arc = in_arc.p;
      lb.order = order.p;
      lb.level_off = lvoff.p;
      lb.out_begin = row.p;  // (the count pass's row table)
      lb.depth = depth.p;
      lb.stream_off = b->lv_stream_off.p;
      lb.stream = b->lv_stream.p;
      he = launch_level_build(lb, b->num_level, c->stream);
      if (he == hipSuccess) he = hipStreamSynchronize(c->stream);  // the planning tables go out of scope below
      if (he != hipSuccess) {
        klat_batch_destroy(b);
        return hip_fail(he, "level streams");
      }
    }
    if (b->num_chain > 0) {  // the count pass's tables stay: klat_viterbi builds the segment path's COST records from them, when it first runs
      b->ch_cnt.take(cnt);
      b->ch_row.take(row);
      b->ch_in_arc.take(in_arc);
      b->ch_out_arc.take(out_arc);
      b->chain_records = nrec_chain;
    }
  }
  b->G = P.G;
  b->W = P.W;
  b->classes = P.classes;
  {  // six workgroups of the deep-prefetch sweep fit a CU (157 VGPRs): use it when no job would have to wait for a slot
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) cus = 256;
    b->deep_prefetch = (int64_t)P.jobs.size() <= (int64_t)6 * cus && std::getenv("KLAT_NO_DEEP_PREFETCH") == nullptr ? 1 : 0;
  }
  b->num_jobs = (int)P.jobs.size();
  b->num_items = (int)P.items.size();
  b->info.num_states = b->S;
  b->info.num_arcs = b->A;
  b->info.fwd_slots = P.fwd_slots;
  b->info.bwd_slots = P.bwd_slots;
  b->info.far_terms = P.far_terms;
  b->info.lanes_per_lattice = P.G;
  b->info.ring_slots = P.W;
  b->info.num_jobs = b->num_jobs;
  b->info.max_rounds = P.max_rounds;
  b->info.num_layouts = (int32_t)P.classes.size();
  b->info.level_lattices = b->num_level;  // lattices on the level-synchronous path
  b->info.chain_lattices = b->num_chain;
  b->info.chain_segments = b->num_chain > 0 ? b->chain_segs : 0;
  if (host_pack) {
    UP(jobs, P.jobs.data(), P.jobs.size());
    UP(lanes, P.lanes.data(), P.lanes.size());
  }
  UP(items, P.items.data(), P.items.size());
  AL(job_flags, P.jobs.size());
  if (!P.jobs.empty()) {  // (the jobs of a class the sweep leaves out never write theirs)
    hipError_t me = hipMemsetAsync(b->job_flags.p, 0, P.jobs.size() * sizeof(int32_t), c->stream);
    if (me != hipSuccess) {
      klat_batch_destroy(b);
      return hip_fail(me, "hipMemsetAsync");
    }
  }
  AL(sink, P.jobs.size() * 64);
  AL(total, b->U);
  AL(tot_fwd, b->U);
  AL(tot_bwd, b->U);
  AL(alpha, b->S + b->U);
  AL(beta, b->S + b->U);
#undef UP
#undef UPP
#undef AL
#undef FAIL
#undef KLAT_HIP_B
  hipError_t e = hipStreamSynchronize(c->stream);  // uploads read pageable host memory of P
  if (e != hipSuccess) {
    klat_batch_destroy(b);
    return hip_fail(e, "upload");
  }
  lap("tail");
  if (host_prof) std::fprintf(stderr, "klat_batch_create host stages (ms):%s\n", host_stages.c_str());
  b->pack_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  *out = b;
  return KLAT_OK;
}

// klat_batch_create_compact in two steps: the uploads (asynchronous from page-locked memory: nothing waits here), and
// everything that does wait for the device (src / dst rebuilt, the ordinary path over device pointers).
// klat_forward_backward_host asks for the next piece's uploads before it finishes this piece.
struct CompactUpload {
  DevBuf<int32_t> deg, label, src, dst, status, time;
  DevBuf<uint8_t> span;
  DevBuf<float> graph, acoustic, fg, fa;
  DevBuf<int64_t> so, ao;
};

static int compact_check(const klat_batch_desc_compact* d) {
  const int64_t U = d->num_lattices;
  if (U < 0 || (U > 0 && (!d->state_off || !d->arc_off))) return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc_compact: null offsets");
  const int64_t S = U ? d->state_off[U] : 0, A = U ? d->arc_off[U] : 0;
  if (A > 0 && (!d->span || !d->label || !d->graph || !d->acoustic)) return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc_compact: null arc arrays");
  if (S > 0 && (!d->out_degree || !d->final_graph || !d->final_acoustic)) return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc_compact: null state arrays");
  return KLAT_OK;
}

static int compact_upload(klat_ctx* c, const klat_batch_desc_compact* d, CompactUpload* up) {
  const int64_t U = d->num_lattices;
  const int64_t S = U ? d->state_off[U] : 0, A = U ? d->arc_off[U] : 0;
  PoolScope pool_scope(&c->pool);
  static const int32_t zero = 0;  // (outlives the asynchronous copy)
  int rc;
  if ((rc = up->deg.upload(d->out_degree, (size_t)S, c->stream)) || (rc = up->span.upload(d->span, (size_t)A, c->stream)) ||
      (rc = up->label.upload(d->label, (size_t)A, c->stream, 4)) || (rc = up->graph.upload(d->graph, (size_t)A, c->stream, 4)) ||
      (rc = up->acoustic.upload(d->acoustic, (size_t)A, c->stream, 4)) || (rc = up->fg.upload(d->final_graph, (size_t)S, c->stream)) ||
      (rc = up->fa.upload(d->final_acoustic, (size_t)S, c->stream)) || (rc = up->src.alloc((size_t)A + 4)) || (rc = up->dst.alloc((size_t)A + 4)) ||
      (rc = up->status.upload(&zero, 1, c->stream)) || (d->state_time && (rc = up->time.upload(d->state_time, (size_t)S, c->stream))))
    return rc;
  if ((rc = up->so.upload(d->state_off, (size_t)U + 1, c->stream)) || (rc = up->ao.upload(d->arc_off, (size_t)U + 1, c->stream))) return rc;
  return KLAT_OK;
}

static int compact_finish(klat_ctx* c, const klat_batch_desc_compact* d, CompactUpload* up, const klat_pack_options* opt, klat_batch** out) {
  const int64_t U = d->num_lattices;
  PoolScope pool_scope(&c->pool);
  // src / dst are rebuilt on the device, and the full descriptor -- device pointers -- takes the ordinary path
  // (its copies are device to device)
  ExpandArgs ea;
  ea.state_off = up->so.p;
  ea.arc_off = up->ao.p;
  ea.out_degree = up->deg.p;
  ea.span = up->span.p;
  ea.src = up->src.p;
  ea.dst = up->dst.p;
  ea.status = up->status.p;
  KLAT_HIP(launch_expand_compact(ea, (int)U, c->stream));
  int32_t h_status = 0;
  KLAT_HIP(hipMemcpyAsync(&h_status, up->status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  if (h_status != 0)
    return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc_compact: the out-degrees of lattice " + std::to_string(h_status - 1) + " do not add up to its arcs");
  klat_batch_desc full;
  std::memset(&full, 0, sizeof(full));
  full.num_lattices = d->num_lattices;
  full.state_off = d->state_off;
  full.arc_off = d->arc_off;
  full.src = up->src.p;
  full.dst = up->dst.p;
  full.ilabel = up->label.p;
  full.olabel = up->label.p;
  full.graph = up->graph.p;
  full.acoustic = up->acoustic.p;
  full.final_graph = up->fg.p;
  full.final_acoustic = up->fa.p;
  full.state_time = d->state_time ? up->time.p : nullptr;
  // the batch uses these arrays where they are, and owns them from here on (a copy of each was 28 bytes an arc of device traffic)
  klat_pack_options o = {0, 0, 0, nullptr, 0};
  if (opt) o = *opt;
  o.borrow_device_arrays = 1;
  const int rc = klat_batch_create(c, &full, &o, out);
  if (rc != KLAT_OK) return rc;
  klat_batch* b = *out;
  auto own = [](auto& mine, auto& theirs) {
    if (mine.borrowed && mine.p == theirs.p) mine.take(theirs);
  };
  own(b->src, up->src);
  own(b->dst, up->dst);
  own(b->ilabel, up->label);  // (olabel keeps pointing into it)
  own(b->graph, up->graph);
  own(b->acoustic, up->acoustic);
  own(b->final_graph, up->fg);
  own(b->final_acoustic, up->fa);
  if (d->state_time) own(b->state_time, up->time);
  return KLAT_OK;
}

int klat_batch_create_compact(klat_ctx* c, const klat_batch_desc_compact* d, const klat_pack_options* opt, klat_batch** out) {
  if (!c || !d || !out) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int rc = compact_check(d);
  if (rc != KLAT_OK) return rc;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);  // (outermost: a failing step's temporaries are freed, not recycled, until this call returns)
  CompactUpload up;
  if ((rc = compact_upload(c, d, &up)) != KLAT_OK) return rc;
  return compact_finish(c, d, &up, opt, out);
}

void klat_batch_destroy(klat_batch* b) {
  if (!b) return;
  klat_ctx* c = b->ctx;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  {
    PoolScope pool_scope(&c->pool);
    for (hipEvent_t e : {b->accum.ev[0], b->accum.ev[1], b->frame.ev[0], b->frame.ev[1]})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->pack_ev)
      if (e) (void)hipEventDestroy(e);
    delete b;
  }
  ctx_release(c);  // the context itself goes with its last batch if its handle was destroyed first
}

int klat_batch_get_info(const klat_batch* b, klat_batch_info* info) {
  if (!b || !info) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  *info = b->info;
  return KLAT_OK;
}

int klat_batch_get_pack_timings(klat_batch* b, klat_pack_timings* t) {
  if (!b || !t) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  std::memset(t, 0, sizeof(*t));
  if (!b->pack_ev[0]) return KLAT_OK;  // KLAT_HOST_PACK=1: nothing ran on the device
  KLAT_HIP(hipSetDevice(b->ctx->device));
  KLAT_HIP(hipEventElapsedTime(&t->count_ms, b->pack_ev[0], b->pack_ev[1]));
  if (b->pack_level_timed) KLAT_HIP(hipEventElapsedTime(&t->level_plan_ms, b->pack_ev[2], b->pack_ev[3]));
  KLAT_HIP(hipEventElapsedTime(&t->rounds_ms, b->pack_ev[4], b->pack_ev[5]));
  KLAT_HIP(hipEventElapsedTime(&t->fill_ms, b->pack_ev[5], b->pack_ev[6]));
  if (b->pack_prep_timed) KLAT_HIP(hipEventElapsedTime(&t->posterior_prep_ms, b->pack_ev[7], b->pack_ev[8]));
  if (b->pack_chain_timed) KLAT_HIP(hipEventElapsedTime(&t->chain_prep_ms, b->pack_ev[9], b->pack_ev[10]));
  t->device_ms = t->count_ms + t->level_plan_ms + t->rounds_ms + t->fill_ms + t->posterior_prep_ms + t->chain_prep_ms;
  t->host_ms = b->pack_host_ms;
  return KLAT_OK;
}

int klat_batch_chain_fallback(klat_batch* b, int32_t* raised) {
  if (!b || !raised) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  *raised = 0;
  if (b->num_chain == 0 || !b->swept) return KLAT_OK;
  KLAT_HIP(hipSetDevice(b->ctx->device));
  KLAT_HIP(hipMemcpyAsync(raised, b->ch_status.p + (b->chain_flag_turn ? 4 : 1), sizeof(int32_t), hipMemcpyDeviceToHost, b->ctx->stream));
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

int klat_batch_topological_depth(klat_batch* b, int32_t* depth) {
  if (!b || (b->U > 0 && !depth)) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  std::vector<int32_t> src((size_t)b->A), dst((size_t)b->A);
  if (b->A > 0) {
    KLAT_HIP(hipMemcpyAsync(src.data(), b->src.p, sizeof(int32_t) * src.size(), hipMemcpyDeviceToHost, b->ctx->stream));
    KLAT_HIP(hipMemcpyAsync(dst.data(), b->dst.p, sizeof(int32_t) * dst.size(), hipMemcpyDeviceToHost, b->ctx->stream));
    KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  }
  std::vector<int32_t> level;
  for (int64_t u = 0; u < b->U; ++u) {
    const int64_t S = b->h_state_off[u + 1] - b->h_state_off[u];
    level.assign((size_t)S, S > 0 ? 1 : 0);  // states on the longest path ending here (unreachable ones count from themselves)
    int32_t best = S > 0 ? 1 : 0;
    for (int64_t i = b->h_arc_off[u]; i < b->h_arc_off[u + 1]; ++i) {  // arcs are in source order, dst > src
      int32_t& l = level[(size_t)dst[(size_t)i]];
      l = std::max(l, level[(size_t)src[(size_t)i]] + 1);
      best = std::max(best, l);
    }
    depth[u] = best;
  }
  return KLAT_OK;
}

// The arc posteriors are made when somebody asks for them: klat_forward_backward without
// KLAT_FB_NO_ARC_POSTERIORS, or klat_get_arc_logpost after a sweep that left them out.  The first pass
// over a batch allocates `logpost` and builds the one-byte source steps and arc spans of the arcs.
static int posterior_tables_pass(klat_batch* b, bool* overflow) {
  klat_ctx* c = b->ctx;
  int rc;
  DevBuf<int32_t> over;
  if ((rc = b->src_delta.alloc((size_t)b->A)) || (rc = b->arc_span.alloc((size_t)b->A)) || (rc = b->item_src.alloc((size_t)b->num_items)) ||
      (rc = over.alloc(1)))
    return rc;
  PosteriorPrepArgs pp;
  pp.items = b->items.p;
  pp.src = b->src.p;
  pp.dst = b->dst.p;
  pp.src_delta = b->src_delta.p;
  pp.span = b->arc_span.p;
  pp.item_src = b->item_src.p;
  pp.overflow = over.p;
  int32_t h_over = 0;
  KLAT_HIP(hipMemsetAsync(over.p, 0, sizeof(int32_t), c->stream));
  if (b->pack_ev[7]) KLAT_HIP(hipEventRecord(b->pack_ev[7], c->stream));
  KLAT_HIP(launch_posterior_prep(pp, b->num_items, c->stream));
  if (b->pack_ev[8]) {
    KLAT_HIP(hipEventRecord(b->pack_ev[8], c->stream));
    b->pack_prep_timed = true;
  }
  KLAT_HIP(hipMemcpyAsync(&h_over, over.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  *overflow = h_over != 0;
  return KLAT_OK;
}

static int ensure_posterior_tables(klat_batch* b) {
  int rc;
  if (b->logpost.n < (size_t)b->A && (rc = b->logpost.alloc((size_t)b->A))) return rc;
  if (b->post_prepared) return KLAT_OK;
  if (std::getenv("KLAT_NO_SRC_DELTA") == nullptr && b->A > 0) {
    bool overflow = false;
    rc = posterior_tables_pass(b, &overflow);
    // a step or an arc above 255 states: the pass reads src and dst themselves.  A pass that failed (out of memory,
    // say: klat_ctx_trim and a second try are the caller's answer) leaves no half-built table behind either, and
    // the batch stays unprepared so that the next call builds them again.
    if (rc != KLAT_OK || overflow) b->src_delta.release(), b->arc_span.release(), b->item_src.release();
    if (rc != KLAT_OK) return rc;
  }
  b->post_prepared = true;  // only now: the tables are whole, or absent
  return KLAT_OK;
}

static void fill_posterior_args(klat_batch* b, const ScaleParams& sp, PosteriorArgs* pa) {
  pa->items = b->items.p;
  pa->state_off = b->state_off.p;
  pa->src = b->src.p;
  pa->dst = b->dst.p;
  pa->ilabel = b->ilabel.p;
  pa->graph = b->graph.p;
  pa->acoustic = b->acoustic.p;
  pa->alpha = b->alpha.p;
  pa->beta = b->beta.p;
  pa->total = b->total.p;
  pa->tot_fwd = b->tot_fwd.p;
  pa->tot_bwd = b->tot_bwd.p;
  pa->logpost = b->logpost.p;
  pa->sp = sp;
  pa->src_delta = b->src_delta.p;
  pa->item_src = b->item_src.p;
  pa->span = b->arc_span.p;
}

// The segment path's arc records for the scales of this call: exp(-cost) for the log-semiring sweep, the cost itself
// for klat_viterbi.  Built from the count pass's tables (which the batch keeps) when a sweep first asks, again when the
// scales change.  *usable = false when the log-semiring records cannot hold these costs (|cost| above 690, or costs so
// large that a lane would have to rescale more often than every four records): the lattices' streams sweep them then.
static int ensure_chain_records(klat_batch* b, const klat_scales& s, const ScaleParams& sp, bool tropical, bool* usable) {
  klat_ctx* c = b->ctx;
  bool& ready = tropical ? b->chain_trop_ready : b->chain_log_ready;
  klat_scales& have = tropical ? b->chain_trop_scales : b->chain_log_scales;
  const bool same = ready && have.graph_scale == s.graph_scale && have.acoustic_scale == s.acoustic_scale && have.insertion_penalty == s.insertion_penalty;
  if (!same) {
    int rc;
    DevBuf<int32_t> sf, sb2;
    DevBuf<uint64_t> arc_records;
    DevBuf<uint64_t>& recs = tropical ? b->ch_recs_t : b->ch_recs;
    if ((rc = recs.alloc(b->chain_records)) || (rc = sf.alloc((size_t)(b->S + 2 * b->U))) || (rc = sb2.alloc((size_t)(b->S + 2 * b->U))) ||
        (rc = arc_records.alloc((size_t)b->A)))
      return rc;
    KLAT_HIP(hipMemsetAsync(b->ch_status.p, 0, sizeof(int32_t), c->stream));       // [0]: a cost that fits no record
    KLAT_HIP(hipMemsetAsync(b->ch_status.p + 2, 0, 2 * sizeof(int32_t), c->stream));  // [2..3]: the largest |cost| (a double's bits)
    ChainPrepArgs cp;
    cp.lats = b->ch_lats.p;
    cp.segs = b->ch_segs.p;
    cp.recs = recs.p;
    cp.state_off = b->state_off.p;
    cp.arc_off = b->arc_off.p;
    cp.src = b->src.p;
    cp.dst = b->dst.p;
    cp.ilabel = b->ilabel.p;
    cp.graph = b->graph.p;
    cp.acoustic = b->acoustic.p;
    cp.final_graph = b->final_graph.p;
    cp.final_acoustic = b->final_acoustic.p;
    cp.cnt = b->ch_cnt.p;
    cp.row = b->ch_row.p;
    cp.in_arc = b->ch_in_arc.p;
    cp.out_arc = b->ch_out_arc.p;
    cp.scratch_f = sf.p;
    cp.scratch_b = sb2.p;
    cp.arc_records = arc_records.p;
    cp.sp = sp;
    cp.tropical = tropical ? 1 : 0;
    cp.status = b->ch_status.p;
    cp.maxcost = reinterpret_cast<unsigned long long*>(b->ch_status.p + 2);
    const bool timed = !b->pack_chain_timed && b->pack_ev[9] && b->pack_ev[10];
    if (timed) KLAT_HIP(hipEventRecord(b->pack_ev[9], c->stream));
    KLAT_HIP(launch_chain_prep(cp, b->num_chain, c->stream));
    if (timed) {
      KLAT_HIP(hipEventRecord(b->pack_ev[10], c->stream));
      b->pack_chain_timed = true;
    }
    int32_t h[4] = {0, 0, 0, 0};
    KLAT_HIP(hipMemcpyAsync(h, b->ch_status.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    KLAT_HIP(hipStreamSynchronize(c->stream));  // (the scratch rows go out of scope)
    have = s;
    ready = true;
    if (!tropical) {
      double maxc = 0.0;
      std::memcpy(&maxc, &h[2], sizeof(maxc));
      // between two rescalings a lane's values may move by (6 + cost / ln 2) binary orders a record (klat_chain.hip)
      const int rescale_records = (int)std::min(64.0, 600.0 / (6.0 + 1.4427 * maxc));
      b->chain_rescale = std::max(1, rescale_records / 4);
      b->chain_log_ok = h[0] == 0 && rescale_records >= 4;
    }
  }
  *usable = tropical ? true : b->chain_log_ok;
  return KLAT_OK;
}

// the segment path's argument block for the log-semiring sweep of batch b (klat_viterbi swaps records and rows)
static void fill_chain_args(klat_batch* b, ChainArgs* out) {
  ChainArgs& ca = *out;
  ca.lats = b->ch_lats.p;
  ca.segs = b->ch_segs.p;
  ca.recs = b->ch_recs.p;
  ca.num_lats = b->num_chain;
  ca.num_segs = b->chain_segs;
  ca.c = b->chain_c;
  ca.lgc = 0;
  while ((1 << ca.lgc) < ca.c) ++ca.lgc;
  ca.W = 2 * ca.c;
  ca.Qm = b->ch_qm.p;
  ca.Qe = b->ch_qe.p;
  ca.Qfm = b->ch_qfm.p;
  ca.Qfe = b->ch_qfe.p;
  ca.bnd_fm = b->ch_bfm.p;
  ca.bnd_fe = b->ch_bfe.p;
  ca.bnd_bm = b->ch_bbm.p;
  ca.bnd_be = b->ch_bbe.p;
  ca.grps = b->ch_grps.p;
  ca.num_grps = b->chain_grps;
  ca.Gm = b->ch_gm.p, ca.Ge = b->ch_ge.p, ca.Gfm = b->ch_gfm.p, ca.Gfe = b->ch_gfe.p;
  ca.gb_fm = b->ch_gbfm.p, ca.gb_fe = b->ch_gbfe.p, ca.gb_bm = b->ch_gbbm.p, ca.gb_be = b->ch_gbbe.p;
  ca.rescale_iters = b->chain_rescale;
  ca.flag = b->ch_status.p + (b->chain_flag_turn ? 4 : 1);  // (klat_forward_backward_ex passes the turn on)
  ca.state_off = b->state_off.p;
  ca.alpha = b->alpha.p;
  ca.beta = b->beta.p;
  ca.ln_table = b->ctx->ln_table.p;
}

int klat_forward_backward(klat_batch* b, const klat_scales* sc) { return klat_forward_backward_ex(b, sc, 0); }

int klat_forward_backward_ex(klat_batch* b, const klat_scales* sc, uint32_t flags) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  klat_scales s = {1.0f, 1.0f, 0.0f};
  if (sc) s = *sc;
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  ScaleParams sp;
  sp.gs_f = s.graph_scale;
  sp.as_f = s.acoustic_scale;
  sp.pen = s.insertion_penalty;
  sp.scaled = (s.acoustic_scale != 1.0f || s.graph_scale != 1.0f) ? 1 : 0;
  sp.penalized = (s.insertion_penalty != 0.0f) ? 1 : 0;
  b->scales = s;

  SweepArgs sa;
  sa.jobs = b->jobs.p;
  sa.lanes = b->lanes.p;
  sa.sw = b->sw.p;
  sa.sg = b->sg.p;
  sa.sa = b->sa.p;
  sa.alpha = b->alpha.p;
  sa.beta = b->beta.p;
  sa.job_flags = b->job_flags.p;
  sa.sink = b->sink.p;
  sa.ln_table = c->ln_table.p;
  sa.sp = sp;
  sa.sc = make_stream_cost(sp);
  sa.ring_slots = 0;  // per layout class below
  sa.far_prefetch = 0;
  if (b->fused) {
    if (s.graph_scale != b->fuse_scales.graph_scale || s.acoustic_scale != b->fuse_scales.acoustic_scale ||
        s.insertion_penalty != b->fuse_scales.insertion_penalty)
      return set_error(KLAT_ERR_STATE, "this batch was created with klat_pack_options.fuse_scales: its streams hold the costs under those scales only");
    sa.sl = b->sl.p;
    sa.lo_unit = b->lo_unit;
    sa.fused = 1;
  }

  const bool want_post = (flags & KLAT_FB_NO_ARC_POSTERIORS) == 0;
  if (flags & ~(uint32_t)KLAT_FB_NO_ARC_POSTERIORS) return set_error(KLAT_ERR_INVALID_ARG, "klat_forward_backward_ex: unknown flag");
  if (want_post) {
    const int rc = ensure_posterior_tables(b);
    if (rc != KLAT_OK) return rc;
  }
  PosteriorArgs pa;
  fill_posterior_args(b, sp, &pa);

  // sweep_ms of a one-class batch is the fast sweep kernel alone (what rocprofv3 lists under its
  // name): the events sit right around its launch.  The exact sweep's launch after it (all of its
  // workgroups exit at once unless the fast sweep flagged a job) falls to totals_ms.
  const bool one_class = b->classes.size() == 1 && b->num_level == 0 && b->num_chain == 0;
  hipEvent_t* ev = c->evring[c->steps % klat_ctx::kTimingSlots];
  for (int i = 0; i < 4; ++i)
    if (!ev[i]) KLAT_HIP(hipEventCreate(&ev[i]));
  // results derived from the previous posteriors are stale from here on
  b->accum.cached_mode = b->accum.dense_mode = 0;
  b->frame.ready = false;
  if (!one_class) KLAT_HIP(hipEventRecord(ev[0], c->stream));
  if (b->num_level > 0) KLAT_HIP(launch_level(b, b->alpha.p, b->beta.p, sp, false));
  {
    std::vector<SweepPiece> work;
    ChainArgs ca;
    bool chain_ok = false;
    if (b->num_chain > 0) {  // the long lattices, in segments
      const int erc = ensure_chain_records(b, s, sp, false, &chain_ok);
      if (erc != KLAT_OK) return erc;
      if (chain_ok) {
        b->chain_flag_turn ^= 1;  // this sweep's flag was cleared by the one before it, and it clears the next one's
        fill_chain_args(b, &ca);
        ca.flag_next = b->ch_status.p + (b->chain_flag_turn ? 1 : 4);
      }
    }
    // the launches, in this order: the classes' fast sweeps (a few long workgroups each: they take their CUs before
    // the segment path's thousands of waves fill the chip), the segment path on the context's stream, the classes'
    // exact sweeps (all of their workgroups leave at once unless the fast sweep flagged a job)
    auto class_args = [&](const LayoutClass& lc) {
      SweepArgs ca2 = sa;
      ca2.jobs = sa.jobs + lc.job_begin;
      ca2.job_flags = sa.job_flags + lc.job_begin;
      ca2.sink = sa.sink + (size_t)lc.job_begin * 64;
      ca2.ring_slots = lc.W;
      ca2.far_prefetch = lc.far;
      ca2.deep_prefetch = b->deep_prefetch;
      ca2.role_mask = kSweepRoleMask;
      if (const char* rm = std::getenv("KLAT_SWEEP_ROLE_MASK")) ca2.role_mask = (uint32_t)std::strtoul(rm, nullptr, 0);
      return ca2;
    };
    const bool with_chain = chain_ok;
    for (int pass = 0; pass < 2; ++pass) {
      int slot = with_chain ? 1 : 0;
      for (const LayoutClass& lcr : b->classes) {
        if (lcr.chain && with_chain) continue;  // (their lattices are the chain piece's)
        const LayoutClass* lcp = &lcr;
        const int which = with_chain ? (pass == 0 ? kSweepFast : kSweepExact) : (pass == 0 ? kSweepFast | kSweepExact : 0);
        if (which)
          work.push_back(SweepPiece{slot, [&, lcp, which](hipStream_t stream) {
            return launch_sweep(class_args(*lcp), lcp->job_end - lcp->job_begin, lcp->G, stream, one_class ? ev[0] : nullptr, one_class ? ev[1] : nullptr, which);
          }});
        ++slot;
      }
      if (pass == 0 && with_chain) {
        // ... and behind the segment path's kernels, on the same stream, the streams its lattices keep: their
        // workgroups leave at once unless it raised its flag (values that do not fit its arithmetic)
        work.push_back(SweepPiece{0, [&](hipStream_t stream) {
          // (its flag was cleared by the sweep before this one: the two flags take turns, no memset between sweeps)
          hipError_t e = launch_chain_sweep(ca, stream);
          for (const LayoutClass& lc : b->classes) {
            if (!lc.chain || e != hipSuccess) continue;
            SweepArgs fa = class_args(lc);
            fa.deep_prefetch = 0;
            fa.run_if = ca.flag;
            e = launch_sweep(fa, lc.job_end - lc.job_begin, lc.G, stream);
          }
          return e;
        }});
      }
    }
    // KLAT_CHAIN_DEBUG=<file>: when and where (CU, SIMD) every wave of the segment path's walks ran, written after this sweep
    DevBuf<unsigned long long> dbg;
    const char* dbg_path = chain_ok ? std::getenv("KLAT_CHAIN_DEBUG") : nullptr;
    size_t dbg_words = 0;
    if (dbg_path) {
      const size_t waves_a = ((size_t)b->chain_segs * ca.c / 64 + 8) / 4 * 4 + 8, waves_c = 2 * (((size_t)b->chain_segs / 64 + 8) / 4 * 4 + 8);
      dbg_words = 4 * waves_a + 8 * waves_c;
      int drc = dbg.alloc(dbg_words);
      if (drc != KLAT_OK) return drc;
      KLAT_HIP(hipMemsetAsync(dbg.p, 0, dbg_words * sizeof(unsigned long long), c->stream));
      ca.dbg = dbg.p;
      ca.dbg_stride = 4 * waves_a;
    }
    const int rc = run_side_by_side(c, work);
    if (rc != KLAT_OK) return rc;
    if (dbg_path) {
      std::vector<unsigned long long> h(dbg_words);
      KLAT_HIP(hipStreamSynchronize(c->stream));
      KLAT_HIP(hipMemcpy(h.data(), dbg.p, dbg_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE* f = std::fopen(dbg_path, "wb")) {
        const unsigned long long head[2] = {(unsigned long long)ca.dbg_stride, (unsigned long long)dbg_words};
        std::fwrite(head, sizeof(head), 1, f);
        std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        std::fclose(f);
      }
    }
  }
  if (!one_class) KLAT_HIP(hipEventRecord(ev[1], c->stream));
  // (the totals are formed by the posterior kernel's first item of every lattice; totals_ms of the
  // timings is what is left between the two events: the exact sweep's launch)
  KLAT_HIP(hipEventRecord(ev[2], c->stream));
  if (want_post) {
    KLAT_HIP(launch_posterior(pa, b->num_items, c->stream));
  } else {  // the lattices' totals alone (the accumulators form the arcs' posteriors from alpha / beta themselves)
    TotalsArgs ta;
    ta.num_lattices = (int32_t)b->U;
    ta.state_off = b->state_off.p;
    ta.alpha = b->alpha.p;
    ta.beta = b->beta.p;
    ta.total = b->total.p;
    ta.tot_fwd = b->tot_fwd.p;
    ta.tot_bwd = b->tot_bwd.p;
    KLAT_HIP(launch_totals(ta, c->stream));
  }
  b->have_logpost = want_post;
  KLAT_HIP(hipEventRecord(ev[3], c->stream));
  b->timing_step = c->steps++;
  b->swept = true;
  return KLAT_OK;
}

static ScaleParams make_scale_params(const klat_scales& s) {
  ScaleParams sp;
  sp.gs_f = s.graph_scale;
  sp.as_f = s.acoustic_scale;
  sp.pen = s.insertion_penalty;
  sp.scaled = (s.acoustic_scale != 1.0f || s.graph_scale != 1.0f) ? 1 : 0;
  sp.penalized = (s.insertion_penalty != 0.0f) ? 1 : 0;
  return sp;
}

int klat_viterbi(klat_batch* b, const klat_scales* sc) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  klat_scales s = {1.0f, 1.0f, 0.0f};
  if (sc) s = *sc;
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  int rc;
  if (b->fwd_cost.n == 0) {
    if ((rc = b->fwd_cost.alloc((size_t)(b->S + b->U)))) return rc;
    if ((rc = b->bwd_cost.alloc((size_t)(b->S + b->U)))) return rc;
  }
  b->viterbi_scales = s;
  SweepArgs va;
  va.jobs = b->jobs.p;
  va.lanes = b->lanes.p;
  va.sw = b->sw.p;
  va.sg = b->sg.p;
  va.sa = b->sa.p;
  va.alpha = b->fwd_cost.p;
  va.beta = b->bwd_cost.p;
  va.job_flags = b->job_flags.p;  // not used by the tropical sweep
  va.sink = b->sink.p;
  va.ln_table = c->ln_table.p;    // not used
  va.sp = make_scale_params(s);
  va.sc = make_stream_cost(va.sp);
  va.ring_slots = 0;
  va.far_prefetch = 0;
  if (b->fused) {
    if (s.graph_scale != b->fuse_scales.graph_scale || s.acoustic_scale != b->fuse_scales.acoustic_scale ||
        s.insertion_penalty != b->fuse_scales.insertion_penalty)
      return set_error(KLAT_ERR_STATE, "this batch was created with klat_pack_options.fuse_scales: its streams hold the costs under those scales only");
    va.sl = b->sl.p;
    va.lo_unit = b->lo_unit;
    va.fused = 1;
  }
  if (b->num_level > 0) KLAT_HIP(launch_level(b, b->fwd_cost.p, b->bwd_cost.p, va.sp, true));
  auto class_launch = [&](const LayoutClass& lc, hipStream_t stream) {
    SweepArgs ca = va;
    ca.jobs = va.jobs + lc.job_begin;
    ca.job_flags = va.job_flags + lc.job_begin;
    ca.sink = va.sink + (size_t)lc.job_begin * 64;
    ca.ring_slots = lc.W;
    ca.far_prefetch = lc.far;
    return launch_sweep_tropical(ca, lc.job_end - lc.job_begin, lc.G, stream);
  };
  if (b->num_chain > 0 && std::getenv("KLAT_CHAIN_VITERBI_STREAMS") == nullptr) {
    // The lattices swept in segments take the (min,+) twin of that path (klat_chain.hip) over COST records, built
    // from the count pass's tables when this first runs on the batch; the others their streams, beside it.
    bool usable = true;
    if ((rc = ensure_chain_records(b, s, va.sp, true, &usable))) return rc;
    ChainArgs ca;
    fill_chain_args(b, &ca);
    ca.recs = b->ch_recs_t.p;
    ca.alpha = b->fwd_cost.p;
    ca.beta = b->bwd_cost.p;
    std::vector<SweepPiece> work;
    int slot = 1;
    for (const LayoutClass& lcr : b->classes) {
      if (lcr.chain) continue;
      const LayoutClass* lcp = &lcr;
      work.push_back(SweepPiece{slot++, [&, lcp](hipStream_t stream) { return class_launch(*lcp, stream); }});
    }
    work.push_back(SweepPiece{0, [&](hipStream_t stream) { return launch_chain_sweep(ca, stream, true); }});
    rc = run_side_by_side(c, work);
  } else {
    rc = for_each_class(b, class_launch);
  }
  if (rc != KLAT_OK) return rc;
  b->viterbi_done = true;
  return KLAT_OK;
}

int klat_get_viterbi(klat_batch* b, double* fwd_cost, double* bwd_cost, double* best) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->viterbi_done) return set_error(KLAT_ERR_STATE, "klat_viterbi has not run on this batch");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  int rc;
  if (b->dense.n < (size_t)b->S && (rc = b->dense.alloc((size_t)b->S))) return rc;
  for (int which = 0; which < 2; ++which) {
    double* host = which ? bwd_cost : fwd_cost;
    if (!host || b->S == 0) continue;
    UnpadArgs ua;
    ua.num_lattices = (int32_t)b->U;
    ua.state_off = b->state_off.p;
    ua.padded = which ? b->bwd_cost.p : b->fwd_cost.p;
    ua.dense = b->dense.p;
    KLAT_HIP(launch_unpad(ua, b->ctx->stream));
    KLAT_HIP(hipMemcpyAsync(host, b->dense.p, sizeof(double) * (size_t)b->S, hipMemcpyDeviceToHost, b->ctx->stream));
  }
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  if (best) {
    // the best complete path of lattice u: the backward cost of its start state
    std::vector<double> bw((size_t)(b->S + b->U));
    if (!bw.empty())
      KLAT_HIP(hipMemcpy(bw.data(), b->bwd_cost.p, sizeof(double) * bw.size(), hipMemcpyDeviceToHost));
    for (int64_t u = 0; u < b->U; ++u)
      best[u] = b->h_state_off[u + 1] > b->h_state_off[u] ? bw[(size_t)(b->h_state_off[u] + u)]
                                                           : std::numeric_limits<double>::infinity();
  }
  return KLAT_OK;
}

int klat_prune_mask(klat_batch* b, float beam, uint8_t* keep_arc, uint8_t* keep_final, double* lattice_beam) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->viterbi_done) return set_error(KLAT_ERR_STATE, "klat_viterbi has not run on this batch");
  if (!(beam > 0.0f)) return set_error(KLAT_ERR_INVALID_ARG, "beam must be positive");
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  DevBuf<uint8_t> d_arc, d_fin;
  DevBuf<double> d_beam;
  int rc;
  if ((rc = d_arc.alloc((size_t)b->A)) || (rc = d_fin.alloc((size_t)b->S)) || (rc = d_beam.alloc((size_t)b->U))) return rc;
  PruneArgs pa;
  pa.state_off = b->state_off.p;
  pa.arc_off = b->arc_off.p;
  pa.src = b->src.p;
  pa.dst = b->dst.p;
  pa.ilabel = b->ilabel.p;
  pa.graph = b->graph.p;
  pa.acoustic = b->acoustic.p;
  pa.final_graph = b->final_graph.p;
  pa.final_acoustic = b->final_acoustic.p;
  pa.fwd_cost = b->fwd_cost.p;
  pa.bwd_cost = b->bwd_cost.p;
  pa.beam = (double)beam;
  pa.keep_arc = d_arc.p;
  pa.keep_final = d_fin.p;
  pa.lattice_beam = d_beam.p;
  pa.sp = make_scale_params(b->viterbi_scales);
  KLAT_HIP(launch_prune(pa, (int)b->U, c->stream));
  if (keep_arc && b->A > 0) KLAT_HIP(hipMemcpyAsync(keep_arc, d_arc.p, (size_t)b->A, hipMemcpyDeviceToHost, c->stream));
  if (keep_final && b->S > 0) KLAT_HIP(hipMemcpyAsync(keep_final, d_fin.p, (size_t)b->S, hipMemcpyDeviceToHost, c->stream));
  if (lattice_beam && b->U > 0)
    KLAT_HIP(hipMemcpyAsync(lattice_beam, d_beam.p, sizeof(double) * (size_t)b->U, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  return KLAT_OK;
}

int klat_batch_get_streams(klat_batch* b, int32_t* sw, float* sg, float* sa) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  const size_t n = (size_t)(b->info.fwd_slots + b->info.bwd_slots);
  if (n == 0) return KLAT_OK;
  if (b->fused && sa) return set_error(KLAT_ERR_STATE, "a batch with fused costs has no acoustic stream");
  if (sw) KLAT_HIP(hipMemcpyAsync(sw, b->sw.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, b->ctx->stream));
  if (sg) KLAT_HIP(hipMemcpyAsync(sg, b->sg.p, n * sizeof(float), hipMemcpyDeviceToHost, b->ctx->stream));
  if (sa) KLAT_HIP(hipMemcpyAsync(sa, b->sa.p, n * sizeof(float), hipMemcpyDeviceToHost, b->ctx->stream));
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

int klat_batch_get_fused_streams(klat_batch* b, int32_t* sw, float* sg, int16_t* sl) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->fused) return set_error(KLAT_ERR_STATE, "the batch was not built with fused costs");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  const size_t n = (size_t)(b->info.fwd_slots + b->info.bwd_slots);
  if (n == 0) return KLAT_OK;
  if (sw) KLAT_HIP(hipMemcpyAsync(sw, b->sw.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, b->ctx->stream));
  if (sg) KLAT_HIP(hipMemcpyAsync(sg, b->sg.p, n * sizeof(float), hipMemcpyDeviceToHost, b->ctx->stream));
  if (sl) KLAT_HIP(hipMemcpyAsync(sl, b->sl.p, n * sizeof(int16_t), hipMemcpyDeviceToHost, b->ctx->stream));
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

int klat_get_timings(const klat_batch* b, klat_timings* t) {
  if (!b || !t) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  if (c->steps - b->timing_step > klat_ctx::kTimingSlots)
    return set_error(KLAT_ERR_STATE, "the timing events of this step have been reused by later steps");
  return read_timings(c, b->timing_step, t);
}

int klat_ctx_timing_history(klat_ctx* c, int32_t count, klat_timings* out) {
  if (!c || !out || count < 0) return set_error(KLAT_ERR_INVALID_ARG, "bad argument");
  if (count > klat_ctx::kTimingSlots || count > c->steps)
    return set_error(KLAT_ERR_INVALID_ARG, "klat_ctx_timing_history: more steps than were run or than the ring holds (512)");
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  for (int32_t i = 0; i < count; ++i) {
    const int rc = read_timings(c, c->steps - count + i, &out[i]);
    if (rc != KLAT_OK) return rc;
  }
  return KLAT_OK;
}

int klat_get_exact_sweep_jobs(klat_batch* b, int32_t* count) {
  if (!b || !count) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  std::vector<int32_t> flags((size_t)b->num_jobs);
  if (b->num_jobs > 0)
    KLAT_HIP(hipMemcpyAsync(flags.data(), b->job_flags.p, sizeof(int32_t) * flags.size(), hipMemcpyDeviceToHost,
                            b->ctx->stream));
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  int32_t n = 0;
  for (int32_t f : flags) n += f != 0;
  *count = n;
  return KLAT_OK;
}

static int fetch(klat_batch* b, void* host, const void* dev, size_t bytes) {
  if (!host || bytes == 0) return KLAT_OK;
  KLAT_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, b->ctx->stream));
  return KLAT_OK;
}

// The fetches without their wait (klat_forward_backward_host enqueues a piece's results and goes on to the next piece).
static int enqueue_alpha_beta(klat_batch* b, double* alpha, double* beta) {
  int rc;
  // the device keeps both arrays padded (one extra slot per lattice): compact, then copy
  if (b->dense.n < (size_t)b->S && (rc = b->dense.alloc((size_t)b->S))) return rc;
  for (int which = 0; which < 2; ++which) {
    double* host = which ? beta : alpha;
    if (!host || b->S == 0) continue;
    UnpadArgs ua;
    ua.num_lattices = (int32_t)b->U;
    ua.state_off = b->state_off.p;
    ua.padded = which ? b->beta.p : b->alpha.p;
    ua.dense = b->dense.p;
    KLAT_HIP(launch_unpad(ua, b->ctx->stream));
    if ((rc = fetch(b, host, b->dense.p, sizeof(double) * (size_t)b->S))) return rc;
  }
  return KLAT_OK;
}

static int enqueue_arc_logpost(klat_batch* b, double* logpost) {
  int rc;
  if (!b->have_logpost) {  // the sweep ran with KLAT_FB_NO_ARC_POSTERIORS: the pass runs now
    if ((rc = ensure_posterior_tables(b))) return rc;
    PosteriorArgs pa;
    fill_posterior_args(b, make_scale_params(b->scales), &pa);
    KLAT_HIP(launch_posterior(pa, b->num_items, b->ctx->stream));
    b->have_logpost = true;
  }
  return fetch(b, logpost, b->logpost.p, sizeof(double) * (size_t)b->A);
}

int klat_get_alpha_beta(klat_batch* b, double* alpha, double* beta) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  int rc;
  if ((rc = enqueue_alpha_beta(b, alpha, beta))) return rc;
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

int klat_get_totals(klat_batch* b, double* total, double* tot_fwd, double* tot_bwd) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  int rc;
  if ((rc = fetch(b, total, b->total.p, sizeof(double) * (size_t)b->U))) return rc;
  if ((rc = fetch(b, tot_fwd, b->tot_fwd.p, sizeof(double) * (size_t)b->U))) return rc;
  if ((rc = fetch(b, tot_bwd, b->tot_bwd.p, sizeof(double) * (size_t)b->U))) return rc;
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

int klat_get_arc_logpost(klat_batch* b, double* logpost) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  int rc;
  if ((rc = enqueue_arc_logpost(b, logpost))) return rc;
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  return KLAT_OK;
}

// Host lattices in, host results out, the batch in pieces over three contexts' streams (include/klat.h).
int klat_forward_backward_host(klat_ctx* c, const klat_batch_desc_compact* d, const klat_pack_options* opt, const klat_scales* scales,
                               int32_t pieces, double* alpha, double* beta, double* total, double* arc_logpost) {
  if (!c || !d) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  const int64_t U = d->num_lattices;
  if (U < 0 || (U > 0 && (!d->state_off || !d->arc_off))) return set_error(KLAT_ERR_INVALID_ARG, "klat_batch_desc_compact: null offsets");
  if (U == 0) return KLAT_OK;
  const int64_t A = d->arc_off[U] - d->arc_off[0];
  if (pieces <= 0) pieces = (int32_t)std::max<int64_t>(1, std::min<int64_t>(8, A / (8 << 20)));  // pieces of eight million arcs or more
  pieces = (int32_t)std::min<int64_t>(pieces, U);
  int rc = compact_check(d);  // (before anything is created for the call)
  if (rc != KLAT_OK) return rc;
  KLAT_HIP(hipSetDevice(c->device));
  // Three helper contexts made together on first use, the caller's own stream left out: the runtime deals streams onto
  // its hardware queues in the order they are created, and three made in a row sit on three queues -- with the caller's
  // stream as one of the three the call took 32 - 35 ms instead of 28 in three of four process set-ups (with and without
  // PyTorch in the process, four or eight hardware queues: tools/host_pipe_probe.py), two streams instead of three 31.6.
  // The helpers belong to the context, so two host threads that share ONE context take turns here (contexts are per host
  // thread, include/klat.h; this keeps a caller that shares one anyway from running two calls over the same streams).
  std::lock_guard<std::mutex> pipe_lock(c->pipe_mu);
  klat_ctx* ctxs[3] = {c, nullptr, nullptr};
  const int lanes = pieces >= 3 ? 3 : pieces;
  for (int i = 0; lanes > 1 && i < lanes; ++i) {
    if (!c->pipe[i]) {
      rc = ctx_create(c->device, &c->pipe[i], false);
      if (rc != KLAT_OK) return rc;
    }
    ctxs[i] = c->pipe[i];
  }
  const uint32_t flags = arc_logpost ? 0u : KLAT_FB_NO_ARC_POSTERIORS;
  klat_batch* live[3] = {nullptr, nullptr, nullptr};
  struct Piece {
    int64_t u0 = 0, u1 = 0, s0 = 0, a0 = 0;
    std::vector<int64_t> so, ao;
    klat_batch_desc_compact desc;
    CompactUpload up;
  };
  // the pieces end at the lattice boundaries nearest their share of the arcs
  std::vector<int64_t> cut((size_t)pieces + 1, 0);
  cut[(size_t)pieces] = U;
  for (int k = 1; k < pieces; ++k) {
    const int64_t want = d->arc_off[0] + A * k / pieces;
    int64_t u = std::lower_bound(d->arc_off + cut[(size_t)k - 1] + 1, d->arc_off + U, want) - d->arc_off;
    cut[(size_t)k] = std::max(cut[(size_t)k - 1] + 1, std::min<int64_t>(u, U - (pieces - k)));
  }
  // a piece's uploads are asked for one piece ahead: they cross PCIe while the piece before is packed and swept
  auto start = [&](int k, Piece* pc) -> int {
    const int slot = k % lanes;
    if (live[slot]) {  // the piece this context carried before: its results are down when its stream is idle
      klat_batch_destroy(live[slot]);
      live[slot] = nullptr;
    }
    pc->u0 = cut[(size_t)k], pc->u1 = cut[(size_t)k + 1];
    pc->s0 = d->state_off[pc->u0], pc->a0 = d->arc_off[pc->u0];
    const int64_t n = pc->u1 - pc->u0;
    pc->so.resize((size_t)n + 1), pc->ao.resize((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) pc->so[(size_t)i] = d->state_off[pc->u0 + i] - pc->s0, pc->ao[(size_t)i] = d->arc_off[pc->u0 + i] - pc->a0;
    klat_batch_desc_compact& p = pc->desc;
    p = *d;
    p.num_lattices = (int32_t)n;
    p.state_off = pc->so.data(), p.arc_off = pc->ao.data();
    p.out_degree = d->out_degree ? d->out_degree + pc->s0 : nullptr;
    p.span = d->span ? d->span + pc->a0 : nullptr;
    p.label = d->label ? d->label + pc->a0 : nullptr;
    p.graph = d->graph ? d->graph + pc->a0 : nullptr;
    p.acoustic = d->acoustic ? d->acoustic + pc->a0 : nullptr;
    p.final_graph = d->final_graph ? d->final_graph + pc->s0 : nullptr;
    p.final_acoustic = d->final_acoustic ? d->final_acoustic + pc->s0 : nullptr;
    p.state_time = nullptr;  // (no accumulator runs on the pieces)
    return compact_upload(ctxs[slot], &p, &pc->up);
  };
  const bool dbg = std::getenv("KLAT_PIPE_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  std::unique_ptr<Piece> cur(new Piece()), next;
  rc = start(0, cur.get());
  for (int k = 0; k < pieces && rc == KLAT_OK; ++k) {
    const int slot = k % lanes;
    const double t0 = now();
    if (k + 1 < pieces) {
      next.reset(new Piece());
      if ((rc = start(k + 1, next.get())) != KLAT_OK) break;
    }
    const double t1 = now();
    klat_batch* b = nullptr;
    {
      PoolScope pool_scope(&ctxs[slot]->pool);
      rc = compact_finish(ctxs[slot], &cur->desc, &cur->up, opt, &b);
    }
    if (rc != KLAT_OK) break;
    const double t2 = now();
    if (dbg) std::fprintf(stderr, "piece %d: at %.2f ms, next piece's uploads asked for in %.2f ms, finish %.2f ms\n", k, t0 - t_begin, t1 - t0, t2 - t1);
    live[slot] = b;
    if ((rc = klat_forward_backward_ex(b, scales, flags)) != KLAT_OK) break;
    {
      PoolScope pool_scope(&b->ctx->pool);
      if ((rc = enqueue_alpha_beta(b, alpha ? alpha + cur->s0 : nullptr, beta ? beta + cur->s0 : nullptr)) != KLAT_OK) break;
      if (arc_logpost && (rc = enqueue_arc_logpost(b, arc_logpost + cur->a0)) != KLAT_OK) break;
      if ((rc = fetch(b, total ? total + cur->u0 : nullptr, b->total.p, sizeof(double) * (size_t)(cur->u1 - cur->u0))) != KLAT_OK) break;
    }
    cur = std::move(next);
  }
  if (rc != KLAT_OK)  // uploads of a piece that never ran may still be in flight on its stream
    for (int i = 0; i < lanes; ++i) (void)hipStreamSynchronize(ctxs[i]->stream);
  cur.reset(), next.reset();
  for (klat_batch*& b : live)
    if (b) klat_batch_destroy(b), b = nullptr;  // (waits for the piece's stream)
  return rc;
}

/* ---- host-only view of the packed streams (no device needed; used by the CPU-side tests
 * that check the stream format against the oracle before any GPU time is spent) ---- */
struct klat_packed {
  klat::PackedBatch P;
  klat_batch_info info;
};

int klat_pack_host(const klat_batch_desc* d, const klat_pack_options* opt, klat_packed** out) {
  if (!d || !out) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  klat_pack_options o = {0, 0, 0, nullptr, 0};
  if (opt) o = *opt;
  o.fuse_scales = nullptr;  // (the host packer defines the generic stream format)
  klat_packed* p = new (std::nothrow) klat_packed();
  if (!p) return set_error(KLAT_ERR_OOM, "host allocation failed");
  std::string err;
  int rc = pack_batch(*d, o, &p->P, &err);
  if (rc != KLAT_OK) {
    delete p;
    return set_error(rc, err);
  }
  std::memset(&p->info, 0, sizeof(p->info));
  p->info.num_states = d->num_lattices ? d->state_off[d->num_lattices] : 0;
  p->info.num_arcs = d->num_lattices ? d->arc_off[d->num_lattices] : 0;
  p->info.fwd_slots = p->P.fwd_slots;
  p->info.bwd_slots = p->P.bwd_slots;
  p->info.far_terms = p->P.far_terms;
  p->info.lanes_per_lattice = p->P.G;
  p->info.ring_slots = p->P.W;
  p->info.num_jobs = (int32_t)p->P.jobs.size();
  p->info.max_rounds = p->P.max_rounds;
  p->info.num_layouts = (int32_t)p->P.classes.size();
  *out = p;
  return KLAT_OK;
}

void klat_packed_destroy(klat_packed* p) { delete p; }

int klat_packed_info(const klat_packed* p, klat_batch_info* info, int64_t* num_slots,
                     int64_t* num_lane_infos, int64_t* num_items) {
  if (!p) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (info) *info = p->info;
  if (num_slots) *num_slots = (int64_t)p->P.sw.size();
  if (num_lane_infos) *num_lane_infos = (int64_t)p->P.lanes.size();
  if (num_items) *num_items = (int64_t)p->P.items.size();
  return KLAT_OK;
}

int klat_packed_copy(const klat_packed* p, int32_t* sw, float* sg, float* sa, void* jobs,
                     void* lanes, void* items) {
  if (!p) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (sw) std::memcpy(sw, p->P.sw.data(), p->P.sw.size() * sizeof(int32_t));
  if (sg) std::memcpy(sg, p->P.sg.data(), p->P.sg.size() * sizeof(float));
  if (sa) std::memcpy(sa, p->P.sa.data(), p->P.sa.size() * sizeof(float));
  if (jobs) std::memcpy(jobs, p->P.jobs.data(), p->P.jobs.size() * sizeof(klat::SweepJob));
  if (lanes) std::memcpy(lanes, p->P.lanes.data(), p->P.lanes.size() * sizeof(klat::LaneInfo));
  if (items) std::memcpy(items, p->P.items.data(), p->P.items.size() * sizeof(klat::PostItem));
  return KLAT_OK;
}

}  // extern "C"
