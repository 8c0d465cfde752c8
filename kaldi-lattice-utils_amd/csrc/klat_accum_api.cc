// klat_accum_api.cc -- C ABI of the accumulate stage (segment / position index, frame
// posteriors) declared in include/klat.h.  Device work is in klat_accum.hip; this file sizes
// scratch, moves the small per-call inputs (label filter, state lengths) and the results.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/klat.h"
#include "klat_accum.h"
#include "klat_internal.h"
#include "klat_nbest.h"

using namespace klat;

static_assert(sizeof(klat_segment_record) == 24 && sizeof(klat_position_record) == 24 &&
                  sizeof(OutRecord) == 24,
              "record layouts must agree");

namespace {

int64_t next_pow2(int64_t n) {
  int64_t p = 1;
  while (p < n) p <<= 1;
  return n <= 0 ? 0 : p;
}

int ensure_index_scratch(klat_batch* b) {
  klat_accum_state& st = b->accum;
  if (st.scratch_ready) return KLAT_OK;
  std::vector<int64_t> off(b->U + 1, 0);
  // (two slots at least: every row then starts on a 16-byte boundary in each of the three scratch arrays)
  for (int64_t u = 0; u < b->U; ++u) off[u + 1] = off[u] + std::max<int64_t>(2, next_pow2(b->h_arc_off[u + 1] - b->h_arc_off[u]));
  st.scratch_slots = off[b->U];
  int rc;
  if ((rc = st.scratch_off.upload(off.data(), off.size(), b->ctx->stream))) return rc;
  if ((rc = st.key_scratch.alloc((size_t)st.scratch_slots * 16))) return rc;
  if ((rc = st.rec_scratch.alloc((size_t)st.scratch_slots * 24))) return rc;
  if ((rc = st.val_scratch.alloc((size_t)st.scratch_slots))) return rc;
  if ((rc = st.num_records.alloc((size_t)b->U))) return rc;
  KLAT_HIP(hipStreamSynchronize(b->ctx->stream));
  st.scratch_ready = true;
  return KLAT_OK;
}

// The cached per-lattice records gathered into one dense device array (utterance-major): what the
// fetch call copies to the host and what the multi-GPU gather sends from (klat_dist.h).
int gather_dense(klat_batch* b, const AccumArgs& a_in, bool position) {
  klat_accum_state& st = b->accum;
  if (st.dense_mode == st.cached_mode) return KLAT_OK;
  // (whoever calls: the pieces of the last sizing call are this state's, not the caller's -- klat_index_device_records,
  // what the multi-GPU gather sends from, came without them and handed long lattices' runs over unmerged)
  AccumArgs a = a_in;
  if (st.pieces_live)
    a.piece_first = st.piece_first.p, a.piece_count = st.piece_count.p, a.item_records = st.item_records.p, a.piece_items = st.piece_items.p;
  else
    a.piece_first = nullptr, a.piece_count = nullptr, a.item_records = nullptr, a.piece_items = nullptr;
  klat_ctx* c = b->ctx;
  const int64_t total = st.record_off[b->U];
  int rc;
  if ((rc = st.d_record_off.upload(st.record_off.data(), st.record_off.size(), c->stream))) return rc;
  if ((rc = st.dense.alloc((size_t)total))) return rc;
  DevBuf<int64_t> chunk_buf;  // (8-byte entries)
  if ((rc = chunk_buf.alloc((size_t)index_gather_chunks(st.record_off.data(), (int)b->U)))) return rc;
  KLAT_HIP(launch_index_gather(a, (int)b->U, position, st.d_record_off.p, st.record_off.data(), chunk_buf.p, st.dense.p, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));  // record_off was read from pageable host memory
  st.dense_mode = st.cached_mode;
  return KLAT_OK;
}

int run_index(klat_batch* b, bool position, const int32_t* state_len, const klat_label_filter* f,
              int64_t* record_off, void* records, int64_t capacity, int64_t* num_records) {
  if (!b || !num_records || !record_off) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  if (b->S > 0 && b->state_time.n == 0)
    return set_error(KLAT_ERR_INVALID_ARG, "the batch was created without state_time: the index accumulators need it");
  if (b->A > 0 && b->olabel.n == 0)
    return set_error(KLAT_ERR_INVALID_ARG, "the batch was created without olabel");
  if (position && b->S > 0 && !state_len) return set_error(KLAT_ERR_INVALID_ARG, "state_len is null");
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  klat_accum_state& st = b->accum;
  int rc = ensure_index_scratch(b);
  if (rc != KLAT_OK) return rc;

  AccumArgs a;
  std::memset(&a, 0, sizeof(a));
  a.state_off = b->state_off.p;
  a.arc_off = b->arc_off.p;
  a.scratch_off = st.scratch_off.p;
  a.src = b->src.p;
  a.dst = b->dst.p;
  a.olabel = b->olabel.p;
  a.state_time = b->state_time.p;
  a.alpha = b->alpha.p;
  a.beta = b->beta.p;
  a.total = b->total.p;
  a.ilabel = b->ilabel.p;
  a.graph = b->graph.p;
  a.acoustic = b->acoustic.p;
  a.sp.gs_f = b->scales.graph_scale;
  a.sp.as_f = b->scales.acoustic_scale;
  a.sp.pen = b->scales.insertion_penalty;
  a.sp.scaled = (b->scales.acoustic_scale != 1.0f || b->scales.graph_scale != 1.0f) ? 1 : 0;
  a.sp.penalized = (b->scales.insertion_penalty != 0.0f) ? 1 : 0;
  a.key_scratch = st.key_scratch.p;
  a.rec_scratch = st.rec_scratch.p;
  a.val_scratch = st.val_scratch.p;
  a.min_arcs = -1;
  a.max_arcs = 0x7FFFFFFF;
  a.num_records = st.num_records.p;
  if (st.pieces_live) a.piece_first = st.piece_first.p, a.piece_count = st.piece_count.p, a.item_records = st.item_records.p, a.piece_items = st.piece_items.p;
  {
    // sort tile: big enough for the largest lattice's keys if that fits 128 KiB of LDS
    int64_t max_arcs = 0;
    for (int64_t u = 0; u < b->U; ++u) max_arcs = std::max(max_arcs, b->h_arc_off[u + 1] - b->h_arc_off[u]);
    const int64_t want = next_pow2(max_arcs) * 24;
    a.tile_bytes = (int32_t)std::min<int64_t>(128 * 1024, std::max<int64_t>(4096, want));
  }

  if (records == nullptr) {  // sizing call: compute and cache
    klat_label_filter flt = {nullptr, 0, nullptr, 0};
    if (f) flt = *f;
    if (flt.num_include < 0 || flt.num_exclude < 0 || (flt.num_include > 0 && !flt.include) ||
        (flt.num_exclude > 0 && !flt.exclude))
      return set_error(KLAT_ERR_INVALID_ARG, "bad label filter");
    std::vector<int32_t> inc(flt.include, flt.include + flt.num_include), exc(flt.exclude, flt.exclude + flt.num_exclude);
    std::sort(inc.begin(), inc.end());
    std::sort(exc.begin(), exc.end());
    if ((rc = st.include.upload(inc.data(), inc.size(), c->stream))) return rc;
    if ((rc = st.exclude.upload(exc.data(), exc.size(), c->stream))) return rc;
    if (position && (rc = st.state_len.upload(state_len, (size_t)b->S, c->stream))) return rc;
    a.include = st.include.p;
    a.num_include = (int32_t)inc.size();
    a.exclude = st.exclude.p;
    a.num_exclude = (int32_t)exc.size();
    a.state_len = st.state_len.p;
    // lattices of up to 16384 arcs take the hash kernel if the batch's keys pack into 62 bits (a
    // table of 2^14 slots up to 8192 arcs, 2^15 above), the others the general kernel
    bool radix = std::getenv("KLAT_INDEX_BITONIC") == nullptr;  // (test hook: force the general kernel)
    RadixPack pk = {0, 0, 0, 0};
    radix = radix && b->A > 0;  // (lattices of up to 16 384 arcs whole, longer ones in pieces where they can be cut)
    st.cached_mode = st.dense_mode = 0;
    if (!st.ev[0]) {
      KLAT_HIP(hipEventCreate(&st.ev[0]));
      KLAT_HIP(hipEventCreate(&st.ev[1]));
    }
    KLAT_HIP(hipEventRecord(st.ev[0], c->stream));
    if (radix && !position && st.pack_mode == 1) {
      pk = st.pack;  // the segment keys' ranges are the batch's own: computed once
      radix = pk.total_bits <= 62;
    } else if (radix) {
      const int32_t init[4] = {0x7FFFFFFF, (int32_t)0x80000000, 0, 0};
      int32_t range[4];
      DevBuf<int32_t> d_range;
      if ((rc = d_range.upload(init, 4, c->stream))) return rc;
      RangeArgs ra = {b->A, b->S, b->olabel.p, b->state_time.p, position ? st.state_len.p : nullptr, d_range.p};
      KLAT_HIP(launch_range(ra, c->stream));
      KLAT_HIP(hipMemcpyAsync(range, d_range.p, sizeof(range), hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipStreamSynchronize(c->stream));
      auto bits = [](uint64_t v) {
        int n = 1;
        while (n < 64 && (v >> n) != 0) ++n;
        return n;
      };
      pk.label_min = range[0];
      pk.label_bits = bits((uint64_t)((int64_t)range[1] - (int64_t)range[0]));
      pk.time_bits = bits((uint64_t)(uint32_t)(position ? range[3] : range[2]));
      pk.total_bits = pk.label_bits + (position ? 1 : 2) * pk.time_bits;
      if (!position) st.pack = pk, st.pack_mode = 1;  // (the position keys depend on the caller's state_len)
      radix = pk.total_bits <= 62;
    }
    DevBuf<long long> d_clock;
    if (radix && std::getenv("KLAT_ACCUM_PROFILE")) {  // diagnostic: where a workgroup of the hash kernel spends its cycles
      if ((rc = d_clock.alloc((size_t)b->U * 8))) return rc;
      KLAT_HIP(hipMemsetAsync(d_clock.p, 0, sizeof(long long) * (size_t)b->U * 8, c->stream));
      a.phase_clock = d_clock.p;
    }
    if (radix) {
      // Size classes: a lattice runs in the smallest workgroup that holds its arcs (128 threads and 17 KiB of LDS
      // for 1 536 arcs ... 1 024 threads and 150 KiB for 16 384), every class one launch over its own list of
      // lattices; a class of fewer than 64 lattices rides with the next larger one that has any.  Above 16 384
      // arcs: the general kernel.
      if (!st.classes_ready) {
        std::vector<int32_t> cls((size_t)b->U);
        int64_t cnt[kHashClasses + 1] = {0};
        // Lattices of 6 145 .. 16 384 arcs run whole in the 1 024-thread classes (one workgroup a CU).  Cutting them in
        // pieces for the 512 x 12 class instead (two workgroups a CU, KLAT_INDEX_PIECES=small) was measured in round 5
        // on the configs[3] shard (12 500 x 14 997 arcs): index 4.21 -> 4.39 ms (the pieces fill their workgroups to
        // 61 %, a slicing launch and an atomic per piece on top), and the gather that merges the pieces' runs
        // 1.9 -> 5.0 ms -- so they stay whole; only lattices above 16 384 arcs are cut (13 312 + 2 048 arcs a piece).
        {
          const char* env = std::getenv("KLAT_INDEX_PIECES");
          st.pieces = env && std::string(env) == "small" ? kPiecesSmall : kPiecesBig;
        }
        for (int64_t u = 0; u < b->U; ++u) {
          const int64_t arcs = b->h_arc_off[u + 1] - b->h_arc_off[u];
          int k = 0;
          while (k < kHashClasses && arcs > kHashClassArcs[k]) ++k;
          if (arcs > st.pieces.long_arcs) k = kHashClasses;  // (long: cut in pieces)
          cls[(size_t)u] = k;
          ++cnt[k];
        }
        int to[kHashClasses + 1];
        for (int k = kHashClasses; k >= 0; --k) {
          to[k] = k;
          if (k < kHashClasses - 1 && cnt[k] < 64) {
            int up = k + 1;
            while (up < kHashClasses && cnt[up] == 0) ++up;
            if (up < kHashClasses) to[k] = to[up], cnt[to[up]] += cnt[k], cnt[k] = 0;
          }
        }
        st.class_begin[0] = 0;
        for (int k = 0; k < kHashClasses; ++k) st.class_begin[k + 1] = st.class_begin[k] + cnt[k];
        std::vector<IndexItem> items((size_t)st.class_begin[kHashClasses]);
        std::vector<int32_t> longs, first((size_t)b->U, -1);
        int64_t at[kHashClasses], slots = 0;
        for (int k = 0; k < kHashClasses; ++k) at[k] = st.class_begin[k];
        for (int64_t u = 0; u < b->U; ++u) {
          const int64_t arcs = b->h_arc_off[u + 1] - b->h_arc_off[u];
          if (cls[(size_t)u] == kHashClasses) {  // a long lattice: piece slots for the slicing to fill
            if (arcs <= (int64_t)1024 * st.pieces.stride) {  // (the slicing kernel: a thread per piece, 1 024 pieces a lattice; beyond: the general kernel)
              longs.push_back((int32_t)u);
              first[(size_t)u] = (int32_t)slots;
              slots += (arcs + st.pieces.stride - 1) / st.pieces.stride;
            } else {
              st.any_general = true;  // (no piece slots: the general kernel must run for it even if nothing else is long)
            }
            continue;
          }
          items[(size_t)at[to[cls[(size_t)u]]]++] = IndexItem{(int32_t)u, 0, (int32_t)arcs, -1};
        }
        st.num_long = (int64_t)longs.size(), st.num_piece_slots = slots;
        if ((rc = st.class_items.upload(items.data(), items.size(), c->stream))) return rc;
        if ((rc = st.long_list.upload(longs.data(), longs.size(), c->stream))) return rc;
        if ((rc = st.piece_first.upload(first.data(), first.size(), c->stream))) return rc;
        if ((rc = st.piece_count.alloc((size_t)b->U)) || (rc = st.item_records.alloc((size_t)slots)) || (rc = st.piece_items.alloc((size_t)slots))) return rc;
        KLAT_HIP(hipStreamSynchronize(c->stream));  // (the lists are locals)
        st.classes_ready = true;
      }
      a.piece_first = st.piece_first.p, a.piece_count = st.piece_count.p, a.item_records = st.item_records.p, a.piece_items = st.piece_items.p;
      st.pieces_live = true;
      KLAT_HIP(hipMemsetAsync(st.piece_count.p, 0, sizeof(int32_t) * (size_t)b->U, c->stream));
      // A batch of one size has one class, one launch.  A mixed one has up to eight, most of them with fewer workgroups
      // than the device holds (a class of 142 lattices takes as long as one of its lattices): the long lattices' chain
      // (slicing, pieces, general kernel) on the context's stream, the classes on side streams, largest first.
      std::vector<SweepPiece> work;
      AccumArgs long_args = a;
      if (st.num_long > 0 || st.any_general) {
        // long lattices: cut where the source states' time changes, the pieces (at most stride + window arcs) through
        // their class's kernel, what could not be cut through the general kernel
        long_args.min_arcs = st.pieces.long_arcs, long_args.max_arcs = 0x7FFFFFFF;
        work.push_back(SweepPiece{0, [&](hipStream_t s) {
                                    hipError_t e = hipSuccess;
                                    if (st.num_long > 0) e = launch_index_slice(a, position, st.pieces, st.long_list.p, (int)st.num_long, s);
                                    if (e == hipSuccess && st.num_long > 0)
                                      e = launch_index_hash2(a, pk, position, st.pieces.cls, st.piece_items.p, (int)st.num_piece_slots, s);
                                    if (e == hipSuccess) e = launch_index(long_args, (int)b->U, position, s);
                                    return e;
                                  }});
      }
      for (int k = kHashClasses - 1, slot = work.empty() ? 0 : 1; k >= 0; --k) {
        const int count = (int)(st.class_begin[k + 1] - st.class_begin[k]);
        if (count <= 0) continue;
        work.push_back(SweepPiece{slot++, [&, k, count](hipStream_t s) { return launch_index_hash2(a, pk, position, k, st.class_items.p + st.class_begin[k], count, s); }});
      }
      if ((rc = run_side_by_side(c, work))) return rc;
    } else {
      st.pieces_live = false;  // (every lattice through the general kernel: whole)
      a.piece_first = nullptr, a.piece_count = nullptr, a.item_records = nullptr, a.piece_items = nullptr;
      KLAT_HIP(launch_index(a, (int)b->U, position, c->stream));
    }
    KLAT_HIP(hipEventRecord(st.ev[1], c->stream));
    if (a.phase_clock) {
      std::vector<long long> clk((size_t)b->U * 8);
      KLAT_HIP(hipMemcpyAsync(clk.data(), d_clock.p, sizeof(long long) * clk.size(), hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipStreamSynchronize(c->stream));
      double sum[8] = {0};
      int64_t cnt = 0;
      for (int64_t u = 0; u < b->U; ++u)
        if (clk[(size_t)u * 8 + 7] > clk[(size_t)u * 8]) {
          ++cnt;
          for (int k = 1; k < 8; ++k) sum[k] += (double)(clk[(size_t)u * 8 + k] - clk[(size_t)u * 8 + k - 1]);
        }
      std::fprintf(stderr,
                   "index_hash_kernel: mean shader cycles per workgroup over %lld lattices: keys %.0f  hash %.0f  classify %.0f  fold %.0f  "
                   "bucket+scatter %.0f  order %.0f  write %.0f\n",
                   (long long)cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt);
      a.phase_clock = nullptr;
    }
    std::vector<int32_t> counts((size_t)b->U);
    if (b->U > 0)
      KLAT_HIP(hipMemcpyAsync(counts.data(), st.num_records.p, sizeof(int32_t) * (size_t)b->U,
                              hipMemcpyDeviceToHost, c->stream));
    KLAT_HIP(hipStreamSynchronize(c->stream));
    st.record_off.assign((size_t)b->U + 1, 0);
    for (int64_t u = 0; u < b->U; ++u) {
      if (counts[u] < 0) return set_error(KLAT_ERR_HIP, "index kernel: scratch overflow");
      st.record_off[u + 1] = st.record_off[u] + counts[u];
    }
    st.cached_mode = position ? 2 : 1;
    std::memcpy(record_off, st.record_off.data(), sizeof(int64_t) * ((size_t)b->U + 1));
    *num_records = st.record_off[b->U];
    return KLAT_OK;
  }

  // fetch call: gather the cached per-lattice records into the caller's dense array
  if (st.cached_mode != (position ? 2 : 1))
    return set_error(KLAT_ERR_STATE, "call with records == NULL first: it computes and sizes the index");
  const int64_t total = st.record_off[b->U];
  if (capacity < total) return set_error(KLAT_ERR_INVALID_ARG, "records capacity too small");
  if ((rc = gather_dense(b, a, position))) return rc;
  if (total > 0)
    KLAT_HIP(hipMemcpyAsync(records, st.dense.p, sizeof(OutRecord) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  std::memcpy(record_off, st.record_off.data(), sizeof(int64_t) * ((size_t)b->U + 1));
  *num_records = total;
  return KLAT_OK;
}

}  // namespace

extern "C" {

int klat_utterance_scores(klat_batch* b, const int64_t* word_off, const int32_t* words, double* logp) {
  if (!b || !word_off) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  const int64_t nw = word_off[b->U];
  if (nw < 0 || (nw > 0 && (!words || !logp))) return set_error(KLAT_ERR_INVALID_ARG, "bad word list");
  if (nw == 0) return KLAT_OK;
  if (b->olabel.n == 0) return set_error(KLAT_ERR_INVALID_ARG, "the batch was created without olabel");
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  std::vector<UttItem> items;
  int64_t max_states = 1;
  for (int64_t u = 0; u < b->U; ++u) {
    if (word_off[u + 1] < word_off[u]) return set_error(KLAT_ERR_INVALID_ARG, "word_off must not decrease");
    max_states = std::max(max_states, b->h_state_off[u + 1] - b->h_state_off[u]);
    for (int64_t k = word_off[u]; k < word_off[u + 1]; k += 64)
      items.push_back(UttItem{(int32_t)u, (int32_t)k, (int32_t)std::min<int64_t>(64, word_off[u + 1] - k), 0});
  }
  // persistent workgroups of one wave, each with its own [states][64] layer; at most 8 GiB of it
  int64_t wgs = std::min<int64_t>((int64_t)items.size(), 256 * 8);
  wgs = std::max<int64_t>(1, std::min<int64_t>(wgs, (8ll << 30) / (max_states * 64 * 8)));
  DevBuf<UttItem> d_items;
  DevBuf<int32_t> d_words, d_next;
  DevBuf<double> d_logp, d_scratch;
  int rc;
  const int32_t zero = 0;
  if ((rc = d_items.upload(items.data(), items.size(), c->stream))) return rc;
  if ((rc = d_words.upload(words, (size_t)nw, c->stream))) return rc;
  if ((rc = d_next.upload(&zero, 1, c->stream))) return rc;
  if ((rc = d_logp.alloc((size_t)nw))) return rc;
  if ((rc = d_scratch.alloc((size_t)(wgs * max_states * 64)))) return rc;
  UttArgs a;
  a.items = d_items.p;
  a.num_items = (int32_t)items.size();
  a.next_item = d_next.p;
  a.state_off = b->state_off.p;
  a.arc_off = b->arc_off.p;
  a.src = b->src.p;
  a.dst = b->dst.p;
  a.ilabel = b->ilabel.p;
  a.olabel = b->olabel.p;
  a.graph = b->graph.p;
  a.acoustic = b->acoustic.p;
  a.beta = b->beta.p;
  a.words = d_words.p;
  a.logp = d_logp.p;
  a.scratch = d_scratch.p;
  a.scratch_states = max_states;
  a.sp.gs_f = b->scales.graph_scale;
  a.sp.as_f = b->scales.acoustic_scale;
  a.sp.pen = b->scales.insertion_penalty;
  a.sp.scaled = (b->scales.acoustic_scale != 1.0f || b->scales.graph_scale != 1.0f) ? 1 : 0;
  a.sp.penalized = (b->scales.insertion_penalty != 0.0f) ? 1 : 0;
  KLAT_HIP(launch_utterance(a, (int)wgs, c->stream));
  KLAT_HIP(hipMemcpyAsync(logp, d_logp.p, sizeof(double) * (size_t)nw, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  return KLAT_OK;
}

int klat_segment_index(klat_batch* b, const klat_label_filter* f, int64_t* record_off,
                       klat_segment_record* records, int64_t capacity, int64_t* num_records) {
  return run_index(b, false, nullptr, f, record_off, records, capacity, num_records);
}

int klat_position_index(klat_batch* b, const int32_t* state_len, const klat_label_filter* f,
                        int64_t* record_off, klat_position_record* records, int64_t capacity,
                        int64_t* num_records) {
  return run_index(b, true, state_len, f, record_off, records, capacity, num_records);
}

int klat_frame_post(klat_batch* b, const int32_t* num_frames, int64_t* frame_off, int32_t* labels,
                    float* logps, int64_t capacity, int64_t* num_entries) {
  if (!b || !num_entries || !frame_off || (b->U > 0 && !num_frames))
    return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->swept) return set_error(KLAT_ERR_STATE, "klat_forward_backward has not run on this batch");
  if (b->S > 0 && b->state_time.n == 0)
    return set_error(KLAT_ERR_INVALID_ARG, "the batch was created without state_time: frame posteriors need it");
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  klat_frame_state& st = b->frame;
  int rc;

  FrameArgs a;
  std::memset(&a, 0, sizeof(a));
  a.state_off = b->state_off.p;
  a.arc_off = b->arc_off.p;
  a.src = b->src.p;
  a.dst = b->dst.p;
  a.ilabel = b->ilabel.p;
  a.graph = b->graph.p;
  a.acoustic = b->acoustic.p;
  a.state_time = b->state_time.p;
  a.alpha = b->alpha.p;
  a.beta = b->beta.p;
  a.total = b->total.p;
  a.sp.gs_f = b->scales.graph_scale;
  a.sp.as_f = b->scales.acoustic_scale;
  a.sp.pen = b->scales.insertion_penalty;
  a.sp.scaled = (b->scales.acoustic_scale != 1.0f || b->scales.graph_scale != 1.0f) ? 1 : 0;
  a.sp.penalized = (b->scales.insertion_penalty != 0.0f) ? 1 : 0;

  if (labels == nullptr || logps == nullptr) {  // sizing call: compute and cache
    st.ready = false;
    st.frame_base.assign((size_t)b->U + 1, 0);
    for (int64_t u = 0; u < b->U; ++u) {
      if (num_frames[u] < 0) return set_error(KLAT_ERR_INVALID_ARG, "negative num_frames");
      st.frame_base[u + 1] = st.frame_base[u] + num_frames[u];
    }
    if ((rc = st.num_frames.upload(num_frames, (size_t)b->U, c->stream))) return rc;
    if ((rc = st.d_frame_base.upload(st.frame_base.data(), st.frame_base.size(), c->stream))) return rc;
    if ((rc = st.frame_ref.alloc((size_t)st.frame_base[b->U]))) return rc;
    a.frame_base = st.d_frame_base.p;
    a.frame_ref = st.frame_ref.p;
    if ((rc = st.arc_scratch.alloc((size_t)b->A * sizeof(FrameArc)))) return rc;
    if ((rc = st.num_expansions.alloc((size_t)b->U))) return rc;
    if ((rc = st.num_records.alloc((size_t)b->U))) return rc;
    if ((rc = st.num_stored.alloc((size_t)b->U))) return rc;
    if ((rc = st.rec_ptr.alloc((size_t)b->U))) return rc;
    if ((rc = st.rec_fmt.alloc((size_t)b->U))) return rc;
    a.num_stored = st.num_stored.p;
    a.num_frames = st.num_frames.p;
    a.num_expansions = st.num_expansions.p;
    a.num_records = st.num_records.p;
    a.rec_ptr = st.rec_ptr.p;
    a.rec_fmt = st.rec_fmt.p;
    a.arc_info = reinterpret_cast<FrameArc*>(st.arc_scratch.p);
    if (!st.ev[0]) {
      KLAT_HIP(hipEventCreate(&st.ev[0]));
      KLAT_HIP(hipEventCreate(&st.ev[1]));
    }
    DevBuf<long long> d_clock;
    if (std::getenv("KLAT_ACCUM_PROFILE")) {
      if ((rc = d_clock.alloc((size_t)b->U * 8))) return rc;
      KLAT_HIP(hipMemsetAsync(d_clock.p, 0, sizeof(long long) * (size_t)b->U * 8, c->stream));
      a.phase_clock = d_clock.p;
    }
    auto print_clock = [&](const char* which) -> int {
      if (!a.phase_clock) return KLAT_OK;
      std::vector<long long> clk((size_t)b->U * 8);
      KLAT_HIP(hipMemcpyAsync(clk.data(), d_clock.p, sizeof(long long) * clk.size(), hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipStreamSynchronize(c->stream));
      double sum[5] = {0};
      for (int64_t u = 0; u < b->U; ++u)
        for (int k = 0; k < 5; ++k) sum[k] += (double)clk[(size_t)u * 8 + k];
      std::fprintf(stderr, "%s: mean shader cycles per workgroup over %lld lattices: spans %.0f  counts+scan %.0f  scatter %.0f  crowded %.0f  frames %.0f\n",
                   which, (long long)b->U, sum[0] / b->U, sum[1] / b->U, sum[2] / b->U, sum[3] / b->U, sum[4] / b->U);
      return KLAT_OK;
    };
    int64_t fr_max_arcs = 0;
    for (int64_t u = 0; u < b->U; ++u) fr_max_arcs = std::max(fr_max_arcs, b->h_arc_off[u + 1] - b->h_arc_off[u]);
    const int max_arcs = (int)std::min<int64_t>(fr_max_arcs, 1 << 30);
    KLAT_HIP(hipEventRecord(st.ev[0], c->stream));
    // The lean kernel first (csrc/klat_frame.hip): it takes its records from a pool sized by what the context's
    // batches needed so far, so nothing counts the (frame, word) entries beforehand.  The lattices it lists --
    // they do not fit it, or the pool was short -- go through the count pass and the general kernel.
    const bool general_only = std::getenv("KLAT_FRAME_GENERAL") != nullptr;  // (tests, comparisons)
    int32_t redo_n = (int32_t)b->U;
    std::vector<int32_t> redo_list;
    bool lean = !general_only && b->U > 0;
    int64_t cap = 0;
    if (lean) {
      const double ratio = std::getenv("KLAT_FRAME_POOL_RATIO") ? std::atof(std::getenv("KLAT_FRAME_POOL_RATIO")) : c->frame_pool_ratio;  // (the variable: tests)
      cap = std::min<int64_t>(((int64_t)1 << 31) - 1, (int64_t)(ratio * (double)b->A) + 4096);
      // No room for the pool: the general kernel needs none (it sizes its scratch behind a count pass).  Only the
      // allocation failed, nothing is running on this call's temporaries: the call is not a failing one.
      if ((rc = st.pool.alloc((size_t)cap))) {
        if (rc != KLAT_ERR_OOM) return rc;
        lean = false;
        c->pool.failing = false;
      }
    }
    if (lean) {
      if ((rc = st.redo.alloc(2 * (size_t)b->U))) return rc;
      // the call's counters and flags in one block (one clearing, one copy back): records asked of the pool / asked for in
      // vain / the pieces' arc rows taken; lattices listed for the second instantiation / for the general kernel / for the
      // dense pass; a flag per lattice (on a list already)
      if ((rc = st.pool_used.alloc(5 + ((size_t)b->U + 1) / 2))) return rc;
      if ((rc = st.dense.alloc((size_t)b->U))) return rc;
      int32_t* counters = reinterpret_cast<int32_t*>(st.pool_used.p + 3);
      a.pool_used = st.pool_used.p;
      a.redo_n = counters, a.redo2_n = counters + 1, a.dense_n = counters + 2;
      a.listed = reinterpret_cast<int32_t*>(st.pool_used.p + 5);
      a.dense = st.dense.p;
      a.pool = st.pool.p;
      a.pool_cap = cap;
      a.redo_list = st.redo.p;
      a.redo2_list = st.redo.p + b->U;
      a.ln_table = c->ln_table.p;
      KLAT_HIP(hipMemsetAsync(st.pool_used.p, 0, sizeof(unsigned long long) * (5 + ((size_t)b->U + 1) / 2), c->stream));
      // size classes: a lattice runs in the smallest workgroup that holds it, every class one launch over its own list
      // (a class of fewer than 64 lattices rides with the next larger one)
      if (!st.classes_ready) {
        std::vector<int32_t> cls((size_t)b->U), longs, pfirst((size_t)b->U, -1);
        int64_t cnt[kFrameClasses] = {0}, slots = 0, long_arcs = 0;
        for (int64_t u = 0; u < b->U; ++u) {
          const int64_t arcs = b->h_arc_off[u + 1] - b->h_arc_off[u], states = b->h_state_off[u + 1] - b->h_state_off[u];
          // too long for the lean kernel (8 192 states or 65 535 arcs): in pieces (frame_slice_kernel)
          if ((states > 8192 || arcs > 65535) && arcs < ((int64_t)1 << 30) && states < ((int64_t)1 << 30)) {
            cls[(size_t)u] = -1;
            longs.push_back((int32_t)u);
            pfirst[(size_t)u] = (int32_t)slots;
            slots += (states + kFramePieceStates - 1) / kFramePieceStates;
            long_arcs += arcs;
            continue;
          }
          int k = 0;
          while (k < kFrameClasses - 1 && (arcs > kFrameClassArcs[k] || states > kFrameClassStates[k])) ++k;
          cls[(size_t)u] = k;
          ++cnt[k];
        }
        st.num_long = (int64_t)longs.size(), st.num_piece_slots = slots;
        st.piece_scratch_cap = long_arcs + long_arcs / 2 + 65536;
        if ((rc = st.long_list.upload(longs.data(), longs.size(), c->stream))) return rc;
        if ((rc = st.piece_first.upload(pfirst.data(), pfirst.size(), c->stream))) return rc;
        if ((rc = st.piece_items.alloc((size_t)slots))) return rc;
        if (slots > 0 && (rc = st.piece_scratch.alloc((size_t)st.piece_scratch_cap))) return rc;
        int to[kFrameClasses];
        for (int k = kFrameClasses - 1; k >= 0; --k) {
          to[k] = k;
          if (k < kFrameClasses - 1 && cnt[k] < 64) to[k] = to[k + 1], cnt[to[k + 1]] += cnt[k], cnt[k] = 0;
        }
        st.class_begin[0] = 0;
        for (int k = 0; k < kFrameClasses; ++k) st.class_begin[k + 1] = st.class_begin[k] + cnt[k], st.class_max_arcs[k] = 0;
        std::vector<int32_t> list((size_t)b->U);
        int64_t at[kFrameClasses];
        for (int k = 0; k < kFrameClasses; ++k) at[k] = st.class_begin[k];
        for (int64_t u = 0; u < b->U; ++u) {
          if (cls[(size_t)u] < 0) continue;
          const int k = to[cls[(size_t)u]];
          list[(size_t)at[k]++] = (int32_t)u;
          st.class_max_arcs[k] = (int32_t)std::max<int64_t>(st.class_max_arcs[k], std::min<int64_t>(b->h_arc_off[u + 1] - b->h_arc_off[u], 1 << 30));
        }
        if ((rc = st.class_list.upload(list.data(), list.size(), c->stream))) return rc;
        const int32_t big_n = (int32_t)(st.class_begin[kFrameClasses] - st.class_begin[kFrameClasses - 1]);
        if ((rc = st.big_n.upload(&big_n, 1, c->stream))) return rc;
        KLAT_HIP(hipStreamSynchronize(c->stream));  // (the list is a local)
        st.classes_ready = true;
      }
      if (st.num_long > 0) {  // long lattices: sliced on the device, the pieces through the second instantiation
        a.piece_items = st.piece_items.p, a.piece_first = st.piece_first.p, a.piece_scratch = st.piece_scratch.p;
        a.piece_scratch_used = st.pool_used.p + 2, a.piece_scratch_cap = st.piece_scratch_cap;
        if ((rc = st.piece_dense.alloc((size_t)st.num_piece_slots))) return rc;
        a.piece_dense = st.piece_dense.p;
        KLAT_HIP(hipMemsetAsync(a.piece_dense, 0, sizeof(int32_t) * (size_t)st.num_piece_slots, c->stream));
      }
      {
        // the long lattices' chain on the context's stream, the size classes beside it on side streams, largest first (a
        // mixed batch's classes have fewer workgroups than the device holds; a batch of one size has one launch)
        std::vector<SweepPiece> work;
        if (st.num_long > 0)
          work.push_back(SweepPiece{0, [&](hipStream_t s) {
                                      hipError_t e = launch_frame_slice(a, st.long_list.p, (int)st.num_long, s);
                                      if (e == hipSuccess) e = launch_frame_lean_pieces(a, (int)st.num_piece_slots, s);
                                      return e;
                                    }});
        for (int k = kFrameClasses - 1, slot = work.empty() ? 0 : 1; k >= 0; --k) {
          const int count = (int)(st.class_begin[k + 1] - st.class_begin[k]);
          if (count <= 0) continue;
          work.push_back(SweepPiece{slot++, [&, k, count](hipStream_t s) {
                                      FrameArgs ak = a;
                                      if (k == kFrameClasses - 1) {  // (spans in L2: the second instantiation, over this list)
                                        ak.redo_list = st.class_list.p + st.class_begin[k], ak.redo_n = st.big_n.p;
                                        return launch_frame_lean(ak, count, st.class_max_arcs[k], true, 0, s);
                                      }
                                      ak.lattice_list = st.class_list.p + st.class_begin[k];
                                      return launch_frame_lean(ak, count, st.class_max_arcs[k], false, k, s);
                                    }});
        }
        if ((rc = run_side_by_side(c, work))) return rc;
      }
      if ((rc = print_clock("frame_lean_kernel"))) return rc;
      struct {
        unsigned long long used[3];  // [0] records asked of the pool, [1] those of them the pool had no room for
        int32_t first_n, general_n, dense_n, pad;
      } back = {{0, 0, 0}, 0, 0, 0, 0};
      KLAT_HIP(hipMemcpyAsync(&back, st.pool_used.p, sizeof(back), hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipStreamSynchronize(c->stream));
      const int32_t first_n = back.first_n, dense_n = back.dense_n;
      unsigned long long used[2] = {back.used[0], back.used[1]};
      redo_n = back.general_n;  // (so far the slicing and the last class list for the general kernel)
      unsigned long long need = used[0];
      const unsigned long long short_first = used[1];
      if (first_n > 0) {  // larger lattices (spans in L2), and one more try for what else did not fit
        KLAT_HIP(launch_frame_lean(a, (int)first_n, max_arcs, true, 0, c->stream));
        if ((rc = print_clock("frame_lean_kernel (second instantiation)"))) return rc;
      }
      // the lattices and pieces marked dense (more than twelve entries a rank): thirty-two registers a rank
      // (the second instantiation, still running, may list up to first_n more)
      if (first_n > 0 || dense_n > 0 || st.num_long > 0) {  // (a batch the first instantiation finished has nothing more to wait for)
        {  // (the dense lattices and the dense pieces side by side: either has a workgroup a CU, and a few rounds of them)
          const int dense_lattices = (int)std::min<int64_t>(b->U, (int64_t)dense_n + first_n), dense_slots = st.num_long > 0 ? (int)st.num_piece_slots : 0;
          std::vector<SweepPiece> work;
          if (dense_slots > 0) work.push_back(SweepPiece{0, [&, dense_slots](hipStream_t s) { return launch_frame_lean_dense(a, 0, dense_slots, s); }});
          if (dense_lattices > 0) work.push_back(SweepPiece{(int)work.size(), [&, dense_lattices](hipStream_t s) { return launch_frame_lean_dense(a, dense_lattices, 0, s); }});
          if ((rc = run_side_by_side(c, work))) return rc;
        }
        if ((rc = print_clock("frame_lean_kernel (dense pass)"))) return rc;
        KLAT_HIP(hipMemcpyAsync(&back, st.pool_used.p, sizeof(back), hipMemcpyDeviceToHost, c->stream));
        KLAT_HIP(hipStreamSynchronize(c->stream));
        redo_n = back.general_n, used[0] = back.used[0], used[1] = back.used[1];
        // the lattices the first pass turned away for want of room asked again: once is what they need
        need = used[0] - std::min(used[0], short_first);
      }
      // The next call's pool: what this batch needed and a tenth, when that is more than it had; otherwise half-way
      // back towards it, so that one batch of long arcs does not size every later pool of the context -- never
      // below the starting ratio, never above what an eighth of the device holds (a batch beyond that is listed
      // and takes the general kernel).
      if (b->A > 0 && std::getenv("KLAT_FRAME_POOL_RATIO") == nullptr) {
        const double asked = 1.1 * (double)need / (double)b->A;
        double next = (int64_t)need > cap ? asked : std::max(asked, 0.5 * (c->frame_pool_ratio + asked));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError(), total_b = (size_t)256 << 30;
        next = std::min(next, (double)(total_b / 8 / sizeof(uint64_t)) / (double)b->A);
        c->frame_pool_ratio = std::max(4.0, next);
      }
      const unsigned long long used_total = need;
      if (a.phase_clock)
        std::fprintf(stderr, "frame_lean_kernel: %d of %lld lattices left by the first instantiation, %d to the general kernel; %llu records asked of a pool of %lld\n",
                     (int)first_n, (long long)b->U, (int)redo_n, used_total, (long long)cap);
      if (redo_n > 0) {
        redo_list.resize((size_t)redo_n);
        KLAT_HIP(hipMemcpyAsync(redo_list.data(), a.redo2_list, sizeof(int32_t) * (size_t)redo_n, hipMemcpyDeviceToHost, c->stream));
        // the general kernel reads its lattices from redo_list / redo_n
        a.redo_list = a.redo2_list;
        a.redo_n = a.redo2_n;
      }
    }
    if (redo_n > 0) {
      KLAT_HIP(launch_frame_count(a, (int)b->U, c->stream));
      std::vector<int64_t> nexp((size_t)b->U);
      KLAT_HIP(hipMemcpyAsync(nexp.data(), st.num_expansions.p, sizeof(int64_t) * (size_t)b->U, hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipStreamSynchronize(c->stream));
      std::vector<uint8_t> listed((size_t)b->U, lean ? 0 : 1);
      for (int32_t u : redo_list) listed[(size_t)u] = 1;
      std::vector<int64_t> off((size_t)b->U + 1, 0);
      for (int64_t u = 0; u < b->U; ++u) {
        if (nexp[u] > (int64_t)1 << 30) return set_error(KLAT_ERR_OOM, "lattice has more than 2^30 (frame,word) expansions");
        off[u + 1] = off[u] + (listed[(size_t)u] ? next_pow2(nexp[u]) + frame_sort_slots(b->h_arc_off[(size_t)u + 1] - b->h_arc_off[(size_t)u]) : 0);
      }
      if ((rc = st.scratch_off.upload(off.data(), off.size(), c->stream))) return rc;
      if ((rc = st.key_scratch.alloc((size_t)off[b->U] * 16))) return rc;
      if ((rc = st.rec_scratch.alloc((size_t)off[b->U] * 16))) return rc;
      a.scratch_off = st.scratch_off.p;
      a.key_scratch = st.key_scratch.p;
      a.rec_scratch = st.rec_scratch.p;
      KLAT_HIP(launch_frame(a, (int)redo_n, max_arcs, c->stream));
      if ((rc = print_clock("frame_kernel"))) return rc;
    }
    a.phase_clock = nullptr;
    KLAT_HIP(hipEventRecord(st.ev[1], c->stream));
    std::vector<int32_t> counts((size_t)b->U), stored((size_t)b->U);
    if (b->U > 0) {
      KLAT_HIP(hipMemcpyAsync(counts.data(), st.num_records.p, sizeof(int32_t) * (size_t)b->U,
                              hipMemcpyDeviceToHost, c->stream));
      KLAT_HIP(hipMemcpyAsync(stored.data(), st.num_stored.p, sizeof(int32_t) * (size_t)b->U,
                              hipMemcpyDeviceToHost, c->stream));
    }
    KLAT_HIP(hipStreamSynchronize(c->stream));
    st.entry_off.assign((size_t)b->U + 1, 0);
    st.stored_records = 0;
    for (int64_t u = 0; u < b->U; ++u) {
      st.entry_off[u + 1] = st.entry_off[u] + counts[u];
      st.stored_records += stored[u];
    }
    st.ready = true;
    *num_entries = st.entry_off[b->U];
    return KLAT_OK;
  }

  if (!st.ready) return set_error(KLAT_ERR_STATE, "call with labels == NULL first: it computes and sizes the posteriors");
  const int64_t total = st.entry_off[b->U], T = st.frame_base[b->U];
  if (capacity < total) return set_error(KLAT_ERR_INVALID_ARG, "capacity too small");
  a.num_frames = st.num_frames.p;
  a.rec_ptr = st.rec_ptr.p;
  a.rec_fmt = st.rec_fmt.p;
  a.frame_base = st.d_frame_base.p;
  a.frame_ref = st.frame_ref.p;
  DevBuf<int64_t> d_eoff, d_fbase, d_foff;
  DevBuf<int32_t> d_lab;
  DevBuf<float> d_lp;
  if ((rc = d_eoff.upload(st.entry_off.data(), st.entry_off.size(), c->stream))) return rc;
  if ((rc = d_fbase.upload(st.frame_base.data(), st.frame_base.size(), c->stream))) return rc;
  if ((rc = d_foff.alloc((size_t)T + 1))) return rc;
  if ((rc = d_lab.alloc((size_t)total))) return rc;
  if ((rc = d_lp.alloc((size_t)total))) return rc;
  KLAT_HIP(launch_frame_gather(a, (int)b->U, d_eoff.p, d_fbase.p, d_foff.p, d_lab.p, d_lp.p, c->stream));
  if (T > 0) KLAT_HIP(hipMemcpyAsync(frame_off, d_foff.p, sizeof(int64_t) * (size_t)T, hipMemcpyDeviceToHost, c->stream));
  if (total > 0) {
    KLAT_HIP(hipMemcpyAsync(labels, d_lab.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    KLAT_HIP(hipMemcpyAsync(logps, d_lp.p, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
  }
  KLAT_HIP(hipStreamSynchronize(c->stream));
  frame_off[T] = total;
  *num_entries = total;
  return KLAT_OK;
}

int klat_index_device_records(klat_batch* b, const void** dev_records, const int64_t** record_off,
                              int64_t* num_lattices, int64_t* num_records) {
  if (!b || !dev_records || !num_records) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  klat_accum_state& st = b->accum;
  if (st.cached_mode == 0)
    return set_error(KLAT_ERR_STATE, "no index on this batch: call klat_segment_index / klat_position_index with records == NULL first");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  AccumArgs a;
  std::memset(&a, 0, sizeof(a));
  a.scratch_off = st.scratch_off.p;
  a.rec_scratch = st.rec_scratch.p;
  a.num_records = st.num_records.p;
  int rc = gather_dense(b, a, st.cached_mode == 2);
  if (rc != KLAT_OK) return rc;
  *dev_records = st.dense.p;
  if (record_off) *record_off = st.record_off.data();
  if (num_lattices) *num_lattices = b->U;
  *num_records = st.record_off[b->U];
  return KLAT_OK;
}

int klat_frame_stored_records(klat_batch* b, int64_t* num_records) {
  if (!b || !num_records) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (!b->frame.ready) return set_error(KLAT_ERR_STATE, "no frame posteriors on this batch: call klat_frame_post with labels == NULL first");
  *num_records = b->frame.stored_records;
  return KLAT_OK;
}

int klat_get_accum_timings(klat_batch* b, float* index_ms, float* frame_ms) {
  if (!b) return set_error(KLAT_ERR_INVALID_ARG, "batch is null");
  KLAT_HIP(hipSetDevice(b->ctx->device));
  PoolScope pool_scope(&b->ctx->pool);
  if (index_ms) {
    *index_ms = 0.0f;
    if (b->accum.ev[1]) {
      KLAT_HIP(hipEventSynchronize(b->accum.ev[1]));
      KLAT_HIP(hipEventElapsedTime(index_ms, b->accum.ev[0], b->accum.ev[1]));
    }
  }
  if (frame_ms) {
    *frame_ms = 0.0f;
    if (b->frame.ev[1]) {
      KLAT_HIP(hipEventSynchronize(b->frame.ev[1]));
      KLAT_HIP(hipEventElapsedTime(frame_ms, b->frame.ev[0], b->frame.ev[1]));
    }
  }
  return KLAT_OK;
}

int klat_nbest(klat_batch* b, const float* arc_cost, int32_t k, int32_t max_labels, int32_t* num_paths,
               float* path_cost, int32_t* path_len, int32_t* path_labels) {
  if (!b || !num_paths || !path_cost || !path_len || !path_labels) return set_error(KLAT_ERR_INVALID_ARG, "null argument");
  if (k < 1 || k > KLAT_NBEST_MAX) return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: k must be in [1, KLAT_NBEST_MAX]");
  if (max_labels < 1) return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: max_labels must be positive");
  if (b->A > 0 && !arc_cost) return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: arc_cost is null");
  if (b->A > 0 && b->olabel.n == 0) return set_error(KLAT_ERR_INVALID_ARG, "the batch was created without olabel");
  if (b->U == 0) return KLAT_OK;
  klat_ctx* c = b->ctx;
  KLAT_HIP(hipSetDevice(c->device));
  PoolScope pool_scope(&c->pool);
  int64_t max_states = 1;
  for (int64_t u = 0; u < b->U; ++u) {
    max_states = std::max(max_states, b->h_state_off[u + 1] - b->h_state_off[u]);
    // (arc 2^25 - 1 with rank 127 would be the back pointer's "none")
    if (b->h_arc_off[u + 1] - b->h_arc_off[u] >= ((int64_t)1 << 25) - 1 || b->h_state_off[u + 1] - b->h_state_off[u] >= (int64_t)1 << 25)
      return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: lattice too large for the 25-bit back pointers");
  }
  // persistent one-wave workgroups, each with [max_states][128] keys of scratch; at most 8 GiB of it
  int64_t wgs = std::min<int64_t>(b->U, 256 * 8);
  wgs = std::max<int64_t>(1, std::min<int64_t>(wgs, (8ll << 30) / (max_states * kNbestWidth * 8)));
  const size_t paths = (size_t)b->U * (size_t)k;
  DevBuf<float> d_cost, d_pcost;
  DevBuf<int32_t> d_next, d_np, d_plen, d_plab, d_status;
  DevBuf<uint64_t> d_scratch;
  int rc;
  const int32_t zero = 0;
  if ((rc = d_cost.upload(arc_cost, (size_t)b->A, c->stream))) return rc;
  if ((rc = d_next.upload(&zero, 1, c->stream))) return rc;
  if ((rc = d_status.upload(&zero, 1, c->stream))) return rc;
  if ((rc = d_np.alloc((size_t)b->U)) || (rc = d_pcost.alloc(paths)) || (rc = d_plen.alloc(paths)) ||
      (rc = d_plab.alloc(paths * (size_t)max_labels)) || (rc = d_scratch.alloc((size_t)(wgs * max_states * kNbestWidth))))
    return rc;
  NbestArgs a;
  std::memset(&a, 0, sizeof(a));
  a.num_lattices = (int32_t)b->U;
  a.next_lattice = d_next.p;
  a.state_off = b->state_off.p;
  a.arc_off = b->arc_off.p;
  a.src = b->src.p;
  a.dst = b->dst.p;
  a.olabel = b->olabel.p;
  a.final_graph = b->final_graph.p;
  a.final_acoustic = b->final_acoustic.p;
  a.arc_cost = d_cost.p;
  a.k = k;
  a.max_labels = max_labels;
  a.scratch = d_scratch.p;
  a.scratch_states = max_states;
  a.num_paths = d_np.p;
  a.path_cost = d_pcost.p;
  a.path_len = d_plen.p;
  a.path_labels = d_plab.p;
  a.status = d_status.p;
  KLAT_HIP(launch_nbest(a, (int)wgs, c->stream));
  int32_t status = 0;
  KLAT_HIP(hipMemcpyAsync(&status, d_status.p, sizeof(status), hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipMemcpyAsync(num_paths, d_np.p, sizeof(int32_t) * (size_t)b->U, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipMemcpyAsync(path_cost, d_pcost.p, sizeof(float) * paths, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipMemcpyAsync(path_len, d_plen.p, sizeof(int32_t) * paths, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipMemcpyAsync(path_labels, d_plab.p, sizeof(int32_t) * paths * (size_t)max_labels, hipMemcpyDeviceToHost, c->stream));
  KLAT_HIP(hipStreamSynchronize(c->stream));
  if (status > 0) return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: lattice " + std::to_string(status - 1) + " has a negative or NaN arc cost");
  if (status < 0) return set_error(KLAT_ERR_INVALID_ARG, "klat_nbest: a path of lattice " + std::to_string(-status - 1) + " has more labels than max_labels");
  return KLAT_OK;
}

}  // extern "C"
