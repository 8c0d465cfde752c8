// klat_accum.h -- argument blocks and launchers of the accumulate stage
// (klat_accum.hip: segment / position index; klat_frame.hip: frame posteriors).
#ifndef KLAT_ACCUM_H_
#define KLAT_ACCUM_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_kernels.h"

namespace klat {

// What one workgroup of the hash kernel takes: a lattice whole (begin 0, end = its arcs, slot -1), or one PIECE of a
// lattice of more than 16 384 arcs -- the arcs [begin, end) of lattice u, cut where the time (the length, for the
// position index) of the arcs' source states changes, so that no key repeats across pieces (index_slice_kernel).
struct IndexItem {
  int32_t u, begin, end, slot;
};

struct AccumArgs {
  const int64_t* state_off;
  const int64_t* arc_off;
  const int64_t* scratch_off;  // [U+1], per-lattice scratch capacity = next_pow2(arcs)
  const int32_t* src;
  const int32_t* dst;
  const int32_t* olabel;
  const int32_t* state_time;
  const int32_t* state_len;  // position index only
  // what forms an arc's log-likelihood of all paths through it, fw[src] + arc_lkh + bw[dst]
  // (kwsbin2/lattice-word-index-segment.cc:160-163): the accumulators read alpha / beta themselves --
  // the 8 B per arc they spent on the posterior pass's `logpost` now go to the arc's two costs -- and
  // subtract the lattice's total when a record is final (:98-108), as the reference does
  const double* alpha;   // padded (klat_kernels.h)
  const double* beta;
  const double* total;   // [U]
  const int32_t* ilabel; // read only when an insertion penalty applies
  const float* graph;
  const float* acoustic;
  ScaleParams sp;
  const int32_t* include;
  int32_t num_include;
  const int32_t* exclude;
  int32_t num_exclude;
  void* key_scratch;  // 16 B per slot
  void* rec_scratch;  // 24 B per slot
  int32_t* num_records;  // [U]
  int32_t tile_bytes;    // dynamic LDS per workgroup (sort tile), <= 128 KiB
  double* val_scratch;   // 8 B per slot: the hash kernel's record values, by record id
  int32_t min_arcs, max_arcs;  // a launch takes the lattices with min_arcs < arcs <= max_arcs
  long long* phase_clock;      // [U][8] or null: s_memtime stamps of the hash kernel's phases (KLAT_ACCUM_PROFILE)
  const IndexItem* items;      // the hash kernel's launch: workgroup k takes items[k]
  // long lattices in pieces: lattice u's pieces are piece_items[piece_first[u] .. + piece_count[u]) (piece_count[u] = 0: the
  // lattice was not cut), piece k's records lie at its `begin` in the lattice's record row, item_records[k] of them;
  // the gather merges the pieces' sorted runs
  IndexItem* piece_items;
  const int32_t* piece_first;  // [U] (-1: not a long lattice)
  int32_t* piece_count;        // [U]
  int32_t* item_records;       // [number of piece slots]
};

}  // namespace klat
#if defined(__HIPCC__)
#include "klat_math.h"
namespace klat {
// alpha[src] + beta[dst] - cost of arc i (a0: the lattice's first arc, pb: its padded state base): the
// expression of the posterior pass (klat_kernels.hip), so the values are the same bits
__device__ __forceinline__ double accum_through(const AccumArgs& a, int64_t a0, int64_t pb, int i, int src, int dst) {
  const bool word = a.sp.penalized ? a.ilabel[a0 + i] != 0 : false;
  const double c = arc_cost(a.graph[a0 + i], a.acoustic[a0 + i], word, a.sp);
  return a.alpha[pb + src] + a.beta[pb + dst] - c;
}
__device__ __forceinline__ double accum_through(const AccumArgs& a, int64_t a0, int64_t pb, int i) {
  return accum_through(a, a0, pb, i, a.src[a0 + i], a.dst[a0 + i]);
}
// a record's value once its arcs are folded: log-zero stays log-zero (also when the lattice has no path: total = -inf)
__device__ __forceinline__ double accum_normalise(double through, double total) { return through > -INFINITY ? through - total : -INFINITY; }
}  // namespace klat
#endif
namespace klat {

struct OutRecord {  // == klat_segment_record / klat_position_record (24 B)
  int32_t f0, f1, f2, f3;
  double logp;
};

hipError_t launch_index(const AccumArgs& a, int num_lattices, bool position, hipStream_t stream);

// Hash path (klat_accum_hash.hip): lattices of at most kHashClassArcs[kHashClasses - 1] arcs whose index keys pack
// into total_bits <= 62 bits -- (label - label_min) in label_bits, times / positions in time_bits
// each.  The ranges come from range_kernel over the batch.
struct RadixPack {
  int32_t label_min;
  int32_t label_bits, time_bits, total_bits;
};
struct RangeArgs {
  int64_t num_arcs, num_states;
  const int32_t* olabel;
  const int32_t* state_time;
  const int32_t* state_len;  // may be null
  int32_t* out;              // [4]: label min, label max, time max, length max
};
hipError_t launch_range(const RangeArgs& a, hipStream_t stream);
// The hash kernel's lattices by size class -- up to kHashClassArcs[c] arcs each, class c's work items in `items`
// (device memory, `count` of them).
static const int kHashClasses = 6;
static const int kHashClassArcs[kHashClasses] = {1536, 3072, 6144, 8192, 15360, 16384};
hipError_t launch_index_hash2(const AccumArgs& a, const RadixPack& pk, bool position, int cls, const IndexItem* items, int count,
                              hipStream_t stream);
// Lattices of more than kHashClassArcs[last] arcs: pieces of at most kPieceStride + kPieceWindow arcs, cut at a change of the
// source states' time within kPieceWindow arcs before every multiple of kPieceStride.  `long_list`: the lattices (device),
// whose piece slots (a.piece_first, ceil(arcs / kPieceStride) each) the kernel fills; a lattice whose times descend
// somewhere, or that has no change of time inside a window, keeps piece_count 0 and takes the general kernel.
// Two shapes: pieces of up to 13 312 + 2 048 arcs for the 1 024 x 15 class (lattices above 16 384 arcs only: the default),
// or of up to 4 096 + 2 048 arcs for the 512 x 12 class, which every lattice above 6 144 arcs then takes (two workgroups
// a CU where the 1 024-thread classes run one; KLAT_INDEX_PIECES=small -- measured in round 5 and not chosen:
// klat_accum_api.cc has the figures).
struct PieceShape {
  int32_t stride, window;  // a cut within `window` arcs before every multiple of `stride`
  int32_t cls;             // the hash kernel's size class of the pieces
  int32_t long_arcs;       // lattices of more arcs are cut
};
static const PieceShape kPiecesBig = {13312, 2048, 4, 16384}, kPiecesSmall = {4096, 2048, 2, 6144};
hipError_t launch_index_slice(const AccumArgs& a, bool position, const PieceShape& shape, const int32_t* long_list, int count, hipStream_t stream);
int64_t index_gather_chunks(const int64_t* record_off_host, int num_lattices);
hipError_t launch_index_gather(const AccumArgs& a, int num_lattices, bool position, const int64_t* record_off, const int64_t* record_off_host,
                               void* chunk_buf, OutRecord* out, hipStream_t stream);

// Utterance index (klat_utt.hip): one item = up to 64 query words of one lattice.
struct UttItem {
  int32_t lattice, word_begin, count, pad;
};
struct UttArgs {
  const UttItem* items;
  int32_t num_items;
  int32_t* next_item;  // work queue head, zeroed before the launch
  const int64_t* state_off;
  const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel; const int32_t* olabel;
  const float* graph; const float* acoustic;
  const double* beta;    // padded (klat_kernels.h)
  const int32_t* words;  // query words, lattice by lattice
  double* logp;          // [number of words] out: log P(word occurs)
  double* scratch;       // [workgroups][scratch_states][64]
  int64_t scratch_states;
  ScaleParams sp;
};
hipError_t launch_utterance(const UttArgs& a, int num_workgroups, hipStream_t stream);

struct FrameArc {  // 16 B
  int32_t k0, k1;  // frames [k0, k1) the arc covers, clipped to the utterance; empty for epsilon arcs
  double x;        // its log-posterior with the arc cost formed the reference's way
};

// Frame buckets above kFrSerialMax entries are sorted by the whole workgroup in scratch rows behind
// the lattice's records; a bucket holds at most one entry per arc.
static constexpr int kFrSerialMax = 256;
__host__ __device__ inline int64_t frame_sort_slots(int64_t num_arcs) {
  if (num_arcs <= kFrSerialMax) return 0;
  int64_t p = 1;
  while (p < num_arcs) p <<= 1;
  return p;
}

// A piece of a lattice too long for the lean frame kernel (more than 8 192 states, 65 535 arcs or 65 535 frames): the frames
// [f0, f1) and everything that can cross them -- the states [s_lo, s_hi) (the piece's own and as many before and behind
// them as the lattice's longest arc spans), the arcs [a_lo, a_hi) leaving the states before the own ones' end.  The times
// of a piece are the lattice's, clipped to the window: an arc that crosses a cut is counted in both pieces, in each for
// the frames of that piece (frame_slice_kernel, klat_frame.hip).
struct FrameItem {
  int32_t u, s_lo, s_hi, a_lo, a_hi, f0, f1, lab_base;
  int64_t scratch;  // the piece's rows in piece_scratch (16 B per arc)
};

struct FrameArgs {
  const int64_t* state_off;
  const int64_t* arc_off;
  const int64_t* scratch_off;  // [U+1], per-lattice capacity = next_pow2(expansions) + frame_sort_slots(arcs)
  const int32_t* src;
  const int32_t* dst;
  const int32_t* ilabel;
  const float* graph;
  const float* acoustic;
  const int32_t* state_time;
  const int32_t* num_frames;  // [U]
  FrameArc* arc_info;         // [A] scratch: clipped span and x of every arc
  const double* alpha;        // padded
  const double* beta;
  const double* total;
  void* key_scratch;  // 16 B per slot
  void* rec_scratch;  // 16 B per slot
  int64_t* num_expansions;  // [U] (count kernel)
  int32_t* num_records;     // [U] (frame, word) entries of the lattice (every frame counts its records)
  int32_t* num_stored;      // [U] records stored (frames without an arc boundary share their predecessor's)
  ScaleParams sp;
  long long* phase_clock;   // [U][8] or null: shader cycles of the frame kernel's phases (KLAT_ACCUM_PROFILE)
  const int64_t* frame_base;  // [U+1] first frame of every lattice in the batch-wide frame numbering
  int2* frame_ref;            // [sum of num_frames] x: records of the frame, y: the first of them in the lattice's record rows
                              // (one word pair: a frame's two numbers are one store, and the stores of a wave one run of lines)
  int32_t* redo_list;         // [U] or null: the lattices the lean kernel leaves to the general one, ...
  int32_t* redo_n;            // [1] ... and their number
  int32_t* redo2_list;        // [U], [1]: what the lean kernel's second instantiation (over redo_list) leaves in its turn
  int32_t* redo2_n;
  const LnTabEntry* ln_table;  // [kLnTabSize] (klat_kernels.h), the lean kernel's LogAdd
  const int32_t* lattice_list; // the lean kernel's first instantiation: workgroup k takes lattice_list[k] (null: lattice k)
  int32_t* listed;             // [U] 1: the lattice is on redo_list / redo2_list already (a lattice's pieces list it once)
  int32_t* dense;              // [U] the lattices with more than twelve entries a rank: the dense pass's (klat_frame.hip), ...
  int32_t* dense_n;            // [1] ... and their number
  int32_t* piece_dense;        // [piece slots] the same for a piece
  // long lattices in pieces (FrameItem): slots [piece_first[u], + ceil(states / kFramePieceStates)), filled by the slicing
  FrameItem* piece_items;
  const int32_t* piece_first;  // [U]
  FrameArc* piece_scratch;     // the pieces' arc rows (x and span), taken with piece_scratch_used
  unsigned long long* piece_scratch_used;
  int64_t piece_scratch_cap;
  uint64_t* pool;             // the lean kernel's records, a lattice's in one piece
  unsigned long long* pool_used;  // [2] records asked for so far (also by the lattices the pool had no room for); those asked for in vain
  int64_t pool_cap;
  const uint64_t** rec_ptr;   // [U] where a lattice's records begin (the pool, or its rows of rec_scratch)
  int64_t* rec_fmt;           // [U] 0: desc(float logp) << 32 | biased label; 1 << 32 | lab_base: the lean kernel's keys
};

hipError_t launch_frame_count(const FrameArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_frame(const FrameArgs& a, int num_lattices, int max_arcs, hipStream_t stream);
// The lean kernel's size classes: class c takes the lattices of at most kFrameClassArcs[c] arcs and kFrameClassStates[c]
// states (workgroups of 128 / 256 / 512 threads with 16 / 32 / 64 KiB of LDS), listed in a.lattice_list; the last class
// whatever is left: the lattices whose spans do not fit LDS, which the host hands to the second instantiation at once
// (listed in a.redo_list, their number in a.redo_n) instead of to the first, which would only list them.  big: the second
// instantiation over a.redo_list.
static const int kFrameClasses = 4;
static const int kFrameClassArcs[kFrameClasses] = {1536, 3072, 6144, 0x7FFFFFFF};
static const int kFrameClassStates[kFrameClasses] = {1024, 2048, 4096, 0x7FFFFFFF};
hipError_t launch_frame_lean(const FrameArgs& a, int num_lattices, int max_arcs, bool big, int size_class, hipStream_t stream);
// Long lattices: pieces of kFramePieceStates own states (the cut moved up to kFramePieceWindow states forward, to where the
// time changes) through the lean kernel's second instantiation; a lattice whose times descend somewhere along its states,
// whose longest arc spans more than kFramePieceWindow states or that has no change of time inside a window is listed for
// the general kernel.
static const int kFramePieceStates = 6144, kFramePieceWindow = 1024;
hipError_t launch_frame_slice(const FrameArgs& a, const int32_t* long_list, int count, hipStream_t stream);
hipError_t launch_frame_lean_pieces(const FrameArgs& a, int num_slots, hipStream_t stream);
hipError_t launch_frame_lean_dense(const FrameArgs& a, int num_lattices, int num_slots, hipStream_t stream);
hipError_t launch_frame_gather(const FrameArgs& a, int num_lattices, const int64_t* entry_off,
                               const int64_t* frame_base, int64_t* frame_off, int32_t* labels, float* logps,
                               hipStream_t stream);

}  // namespace klat
#endif
