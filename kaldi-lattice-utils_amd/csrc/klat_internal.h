// klat_internal.h -- definitions behind the opaque handles of include/klat.h.
#ifndef KLAT_INTERNAL_H_
#define KLAT_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/klat.h"
#include "klat_device.h"
#include "klat_accum.h"
#include "klat_chain.h"
#include "klat_kernels.h"
#include "klat_level.h"
#include "klat_pack.h"

namespace klat {

int set_error(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define KLAT_HIP(expr)                                        \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return ::klat::hip_fail(_e, #expr); \
  } while (0)

// Device allocations of a context are recycled: hipFree synchronises the whole device (a second
// context's sweep stalls behind it) and hipMalloc + hipFree of a batch's ~20 arrays cost tens of
// milliseconds, several times the sweep they serve.  A freed block goes to its context's pool and
// is handed out again for a request of at most the same size and at least half of it (batches of
// one shape recur).  What the pool keeps is bounded -- a quarter of the device's memory at most
// (max_bytes, set when the context is created) -- and can be given back: klat_ctx_trim, the
// destruction of the context, and an out-of-memory hipMalloc anywhere in the library, which empties
// the pools of EVERY context of the process before it tries again (the registry below).  While a
// call is failing (`failing`, set by hip_fail, cleared when the call returns) its temporaries go
// back through hipFree, which waits for the device: kernels of the failed call may still be running
// on them, and a recycled block would be handed to the next call at once.
struct DevPool {
  static constexpr size_t kMaxBytes = (size_t)96 << 30;
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  size_t cached = 0;
  size_t max_bytes = kMaxBytes;
  bool failing = false;
  DevPool() { registry(this, true); }
  ~DevPool() { registry(this, false); }
  void* take(size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = free_blocks.lower_bound(bytes);
    if (it == free_blocks.end() || it->first > 2 * bytes + 4096) return nullptr;
    void* p = it->second;
    *got = it->first;
    cached -= it->first;
    free_blocks.erase(it);
    return p;
  }
  bool give(void* p, size_t bytes) {
    std::lock_guard<std::mutex> lock(mu);
    if (failing || cached + bytes > max_bytes) return false;
    free_blocks.emplace(bytes, p);
    cached += bytes;
    return true;
  }
  void clear() {
    std::lock_guard<std::mutex> lock(mu);
    for (auto& kv : free_blocks) (void)hipFree(kv.second);
    free_blocks.clear();
    cached = 0;
  }
  // every pool of the process (contexts of all devices: a block cached for one GPU is of no use to an
  // allocation on another, but telling them apart is not worth a device query per block)
  static void registry(DevPool* p, bool add) {
    static std::mutex rmu;
    static std::vector<DevPool*>* all = new std::vector<DevPool*>();  // never destroyed: contexts may outlive static destruction
    std::lock_guard<std::mutex> lock(rmu);
    if (p) {
      if (add) {
        all->push_back(p);
      } else {
        for (size_t i = 0; i < all->size(); ++i)
          if ((*all)[i] == p) {
            all->erase(all->begin() + (long)i);
            break;
          }
      }
      return;
    }
    for (DevPool* q : *all) q->clear();  // p == nullptr: empty them all
  }
  static void clear_all() { registry(nullptr, false); }
};
// the pool the calling thread's current API call allocates from (set by PoolScope)
DevPool*& current_pool();
struct PoolScope {
  DevPool* saved;
  explicit PoolScope(DevPool* p) : saved(current_pool()) { current_pool() = p; }
  // (the call's locals are gone by now: a failing call has returned its temporaries through hipFree).  Only the
  // OUTERMOST scope on a pool ends the failing state: an API call that wraps another one on the same context
  // (klat_batch_create_compact -> klat_batch_create) still holds temporaries the failed call's kernels may be reading.
  ~PoolScope() {
    if (current_pool() && current_pool() != saved) current_pool()->failing = false;
    current_pool() = saved;
  }
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;    // elements in use
  size_t cap = 0;  // elements allocated
  DevPool* pool = nullptr;  // where the block came from and goes back to
  bool borrowed = false;    // p is the caller's memory (klat_pack_options.borrow_device_arrays): never freed here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && !borrowed && !(pool && pool->give(p, cap * sizeof(T)))) (void)hipFree(p);
    p = nullptr;
    n = cap = 0;
    borrowed = false;
  }
  void borrow(const T* device_ptr, size_t count) {  // the caller keeps the array alive and unchanged while this object uses it
    release();
    p = const_cast<T*>(device_ptr), n = cap = count, pool = nullptr, borrowed = true;
  }
  // contents unspecified; an allocation that is large enough is kept (hipFree + hipMalloc of the
  // gigabyte-sized scratch of the accumulate stage costs tens of milliseconds)
  int alloc(size_t count) {
    if (p && count <= cap) {
      n = count;
      return KLAT_OK;
    }
    release();
    const size_t want = count == 0 ? 1 : count;  // keep pointers non-null for empty batches
    pool = current_pool();
    size_t got = 0;
    if (pool && (p = static_cast<T*>(pool->take(want * sizeof(T), &got)))) {
      n = count;
      cap = got / sizeof(T);
      return KLAT_OK;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e == hipErrorOutOfMemory) {  // give the cached blocks of every context back and try once more
      (void)hipGetLastError();
      DevPool::clear_all();
      e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    }
    if (e != hipSuccess) {
      p = nullptr;
      return hip_fail(e, "hipMalloc");
    }
    n = count;
    cap = want;
    return KLAT_OK;
  }
  void take(DevBuf& o) {  // steals o's block
    release();
    p = o.p, n = o.n, cap = o.cap, pool = o.pool, borrowed = o.borrowed;
    o.p = nullptr;
    o.n = o.cap = 0;
    o.borrowed = false;
  }
  // pad: elements allocated (not copied) behind the array: the stream packer reads the arc arrays four elements a load,
  // and the last load of the batch may reach past the last arc
  int upload(const T* host, size_t count, hipStream_t s, size_t pad = 0) {
    int rc = alloc(count + pad);
    if (rc != KLAT_OK) return rc;
    n = count;
    if (count == 0) return KLAT_OK;
    // hipMemcpyDefault: the source may be host or device memory (klat_batch_desc's arc and state
    // arrays may already live on the GPU)
    hipError_t e = hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyDefault, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync");
    return KLAT_OK;
  }
};

// Launches side by side (klat_capi.cc): the first on the context's stream, the others on side streams that fork from it
// and join it again.  A piece names its stream (0: the context's; 1 ..: a side stream) and is launched in list order.
struct SweepPiece {
  int slot;
  std::function<hipError_t(hipStream_t)> launch;
};
int run_side_by_side(klat_ctx* c, const std::vector<SweepPiece>& work);

}  // namespace klat

struct klat_ctx {
  double frame_pool_ratio = 4.0;  // records per arc the frame posteriors' pool is sized for: raised to what a batch asked for
  // A context lives until its handle AND every batch created on it are gone: a caller that destroys
  // the context first (a test fixture torn down while a failed test still holds its batch, an exit
  // handler closing things in the wrong order) must not leave the batches with a dangling stream --
  // that ended in `std::bad_variant_access` out of the HIP runtime, or in a hang, at process exit.
  std::atomic<int> refs{1};
  int device = 0;
  hipStream_t stream = nullptr;
  // timing events of the last kTimingSlots calls of klat_forward_backward on this context (a ring:
  // a caller that enqueues many steps without waiting can read all their kernel times afterwards)
  static const int kTimingSlots = 512;
  hipEvent_t evring[kTimingSlots][4] = {};
  int64_t steps = 0;  // calls of klat_forward_backward so far
  // layout classes of one batch sweep side by side: class 0 on `stream`, the others on these,
  // forked from and joined to `stream` by events (created on first use)
  static const int kSideStreams = 3;
  hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[kSideStreams] = {nullptr, nullptr, nullptr};
  klat::DevBuf<klat::LnTabEntry> ln_table;  // table behind the sweep's natural log
  klat::DevPool pool;                       // recycled device blocks of this context's batches
  // klat_forward_backward_host pipelines the pieces of a batch over three helper contexts (streams and pools of their
  // own), created when the call first runs and gone with this context
  klat_ctx* pipe[3] = {nullptr, nullptr, nullptr};
  std::mutex pipe_mu;  // one klat_forward_backward_host at a time per context: the helpers' streams and pools are the call's
  // page-locked host scratch of klat_batch_create (the count pass's verdicts come back through it: into pageable memory the
  // copy goes through the runtime's staging buffers, ~0.1 ms for 10 000 lattices); one call at a time uses it, a second
  // one running beside it on the same context takes pageable memory
  std::mutex pinned_mu;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
};

// device scratch + cached result of the accumulate stage (klat_accum_api.cc)
struct klat_accum_state {
  bool scratch_ready = false;
  int64_t scratch_slots = 0;
  int cached_mode = 0;  // 0 none, 1 segment, 2 position
  int dense_mode = 0;   // the mode `dense` was gathered for (0: stale)
  int pack_mode = 0;    // the mode `pack` was computed for (label / time ranges of the batch: they do not change)
  klat::RadixPack pack = {0, 0, 0, 0};
  klat::DevBuf<klat::OutRecord> dense;  // the cached records, gathered utterance-major
  klat::DevBuf<int64_t> d_record_off;
  hipEvent_t ev[2] = {nullptr, nullptr};  // around the index kernels of the last sizing call
  std::vector<int64_t> record_off;
  klat::DevBuf<int64_t> scratch_off;
  klat::DevBuf<unsigned char> key_scratch, rec_scratch;
  klat::DevBuf<double> val_scratch;
  klat::DevBuf<int32_t> num_records, include, exclude, state_len;
  // the hash kernel's size classes (klat_accum.h): class c's lattices are class_list[class_begin[c] .. class_begin[c + 1])
  bool classes_ready = false;
  klat::PieceShape pieces = klat::kPiecesBig;  // how this batch's long lattices are cut (chosen with the classes)
  klat::DevBuf<klat::IndexItem> class_items;
  int64_t class_begin[klat::kHashClasses + 1] = {0};
  // lattices of more than kHashClassArcs[last] arcs: cut into pieces on the device (index_slice_kernel), a piece a work
  // item of the largest classes; what could not be cut takes the general kernel
  int64_t num_long = 0, num_piece_slots = 0;
  bool any_general = false;  // some lattice is longer than the slicing takes (1 024 pieces): the general kernel runs for it
  bool pieces_live = false;  // the last sizing call went through the hash kernel (piece_count is current)
  klat::DevBuf<int32_t> long_list, piece_first, piece_count, item_records;
  klat::DevBuf<klat::IndexItem> piece_items;
};

struct klat_frame_state {
  bool ready = false;
  hipEvent_t ev[2] = {nullptr, nullptr};  // around the frame kernels of the last sizing call
  std::vector<int64_t> frame_base, entry_off;
  klat::DevBuf<int64_t> scratch_off, num_expansions;
  klat::DevBuf<int32_t> num_frames, num_records, num_stored, redo;
  klat::DevBuf<int2> frame_ref;
  int64_t stored_records = 0;  // records the last run stored (the (frame, word) entries share them)
  klat::DevBuf<int64_t> d_frame_base;
  klat::DevBuf<unsigned char> key_scratch, rec_scratch, arc_scratch;
  klat::DevBuf<uint64_t> pool;             // the lean kernel's records (csrc/klat_frame.hip)
  klat::DevBuf<unsigned long long> pool_used;
  klat::DevBuf<const uint64_t*> rec_ptr;   // [U] where every lattice's records begin
  klat::DevBuf<int64_t> rec_fmt;           // [U] their format (klat_accum.h)
  // the lean kernel's size classes: class c's lattices are class_list[class_begin[c] .. class_begin[c + 1])
  bool classes_ready = false;
  klat::DevBuf<int32_t> class_list, listed, dense, piece_dense, big_n;  // big_n: one word, the last class's size
  // lattices too long for the lean kernel, sliced on the device (frame_slice_kernel)
  int64_t num_long = 0, num_piece_slots = 0, piece_scratch_cap = 0;
  klat::DevBuf<int32_t> long_list, piece_first;
  klat::DevBuf<klat::FrameItem> piece_items;
  klat::DevBuf<klat::FrameArc> piece_scratch;
  int64_t class_begin[klat::kFrameClasses + 1] = {0};
  int32_t class_max_arcs[klat::kFrameClasses] = {0};
};

struct klat_batch {
  klat_accum_state accum;
  klat_frame_state frame;
  klat_ctx* ctx = nullptr;
  int64_t U = 0, S = 0, A = 0;
  int G = 1, W = 8, num_jobs = 0, num_items = 0;  // G, W: of the largest layout class (info only)
  std::vector<klat::LayoutClass> classes;
  int32_t deep_prefetch = 0;  // SweepArgs.deep_prefetch of this batch
  bool swept = false;
  bool have_logpost = false;   // the last sweep (or a later klat_get_arc_logpost) ran the arc-posterior pass
  bool post_prepared = false;  // the posterior pass's one-byte tables were built (first pass over the batch)
  int64_t timing_step = -1;  // the context's step counter at this batch's last klat_forward_backward
  klat_scales scales = {1.f, 1.f, 0.f};
  klat_batch_info info = {};
  std::vector<int64_t> h_state_off, h_arc_off;
  // caller-order CSR/SoA batch
  klat::DevBuf<int64_t> state_off, arc_off;
  klat::DevBuf<int32_t> src, dst, ilabel, olabel, state_time;
  klat::DevBuf<float> graph, acoustic;
  // lockstep streams
  klat::DevBuf<int32_t> sw;
  klat::DevBuf<float> sg, sa;
  // fused costs (klat_pack_options.fuse_scales): sg = (float)cost, sl = the residual in units of lo_unit, no sa
  bool fused = false;
  klat_scales fuse_scales = {1.f, 1.f, 0.f};
  double lo_unit = 0.0;
  klat::DevBuf<int16_t> sl;
  klat::DevBuf<klat::SweepJob> jobs;
  klat::DevBuf<klat::LaneInfo> lanes;
  klat::DevBuf<klat::PostItem> items;
  klat::DevBuf<uint8_t> src_delta;   // src as one-byte steps (posterior pass); empty if a step overflowed
  klat::DevBuf<uint8_t> arc_span;    // dst - src in one byte (the same)
  klat::DevBuf<int32_t> item_src;    // source of the first arc of every posterior item
  // results
  klat::DevBuf<int32_t> job_flags;      // fast sweep -> exact sweep hand-over
  klat::DevBuf<double> total, tot_fwd, tot_bwd, logpost;
  klat::DevBuf<double> alpha, beta;     // padded: S + U entries (klat_kernels.h)
  klat::DevBuf<double> dense;           // scratch of klat_get_alpha_beta
  klat::DevBuf<double> sink;            // SweepArgs::sink
  klat::DevBuf<double> fwd_cost, bwd_cost;  // klat_viterbi (padded), allocated on first use
  klat::DevBuf<float> final_graph, final_acoustic;
  bool viterbi_done = false;
  klat_scales viterbi_scales = {1.f, 1.f, 0.f};
  // lattices on the level-synchronous path (klat_level.hip): not in any stream job
  int32_t num_level = 0, level_max_states = 0;
  klat::DevBuf<int32_t> lv_lattices, lv_records, lv_stream;
  // klat_batch_create's device passes: events around count | level plan | rounds | fill | posterior prep
  static const int kPackEvents = 11;  // (9, 10: around the segment path's record pass)
  hipEvent_t pack_ev[kPackEvents] = {};
  bool pack_level_timed = false, pack_prep_timed = false, pack_chain_timed = false;
  // long lattices swept in segments (klat_chain.hip).  Their stream jobs stay, as layout classes of their own
  // (LayoutClass::chain): klat_viterbi runs them, the log-semiring sweep leaves them out.
  int32_t num_chain = 0, chain_segs = 0, chain_c = 0, chain_seg_len = 0;
  klat::DevBuf<klat::ChainLattice> ch_lats;
  klat::DevBuf<klat::ChainSegment> ch_segs;
  klat::DevBuf<uint64_t> ch_recs;
  klat::DevBuf<double> ch_qm, ch_qfm, ch_bfm, ch_bbm;
  klat::DevBuf<int32_t> ch_qe, ch_qfe, ch_bfe, ch_bbe;
  int32_t chain_grps = 0;  // groups of kChainGroup segments (the two-level combine)
  klat::DevBuf<klat::ChainGroup> ch_grps;
  klat::DevBuf<double> ch_gm, ch_gfm, ch_gbfm, ch_gbbm;
  klat::DevBuf<int32_t> ch_ge, ch_gfe, ch_gbfe, ch_gbbe;
  // for klat_viterbi on those lattices: the count pass's tables and, once it has run, the COST records of the (min,+) twin
  klat::DevBuf<int32_t> ch_cnt, ch_row, ch_in_arc, ch_out_arc;
  klat::DevBuf<uint64_t> ch_recs_t;
  size_t chain_records = 0;
  bool chain_trop_ready = false, chain_log_ready = false, chain_log_ok = false;
  klat_scales chain_trop_scales = {1.f, 1.f, 0.f}, chain_log_scales = {1.f, 1.f, 0.f};  // what the records were built for
  klat::DevBuf<int32_t> ch_status;  // [0] the record pass met a cost that fits no record, [1] and [4] the sweep's fall-back flags (they take turns), [2..3] the largest |cost|
  int chain_flag_turn = 0;          // which of the two the last (or the running) sweep uses
  int32_t chain_rescale = 1;        // ChainArgs::rescale_iters
  float pack_host_ms = 0.0f;  // wall time of klat_batch_create (uploads, host planning, waits included)
  klat::DevBuf<int64_t> lv_stream_off;
};

#endif
