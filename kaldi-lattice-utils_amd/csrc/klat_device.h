// klat_device.h -- structures shared by the host packer and the HIP kernels.
//
// Device data layout (DESIGN.md "Data layout in HBM"):
//   * caller-order SoA arc arrays (src,dst,ilabel,olabel,graph,acoustic) -- the CSR-flattened
//     batch exactly as klat_batch_desc hands it over;
//   * two lockstep arc streams (forward: terms grouped by destination state in ascending
//     order; backward: terms grouped by source state in descending order), sliced 64 lanes
//     wide: a wave owns 64/G lattices, G lanes each; one *round* is one term per lane; four
//     rounds of one lane are contiguous (16 B: a *unit*) so that every stream load is one
//     contiguous dwordx4 wave access.  A job whose states never hold more than A < G terms in a
//     round (out-degree 3 with G = 4: the whole backward stream) stores only the units of its first
//     A lanes per lattice: a chunk row is A * (64/G) units, lane (slot, sub) owns unit
//     sub * (64/G) + slot of it, and the lanes sub >= A carry padding the kernel makes up itself;
//   * natural-log alpha/beta per state (written by the sweep itself), per-arc log posterior.
#ifndef KLAT_DEVICE_H_
#define KLAT_DEVICE_H_
#include <stdint.h>

namespace klat {

// ---- stream term word -------------------------------------------------------------------
// Everything about a term that does not depend on state values is resolved by the packer,
// which knows the layout (G, W) the stream is built for:
// bits 0..11  RD    ring cell ((target position - rel) mod W) * (64/G) + lattice slot to read
// bits 12..23 WR    ring cell of the target state (only on the COMMIT lane)
// bit  24     COMMIT  LAST round and this lane (first of the lattice's G) stores the result
// bit  25     NIL   padding term, contributes log-zero; its graph cost in the stream is +inf, so
//                   the kernel needs no test of this bit to give it weight zero (never with FAR)
// bit  26     ONE   constant-one source (alpha[start] = 0, final weights)
// bit  27     FAR   source older than the ring: bits 0..26 then hold rel = target position -
//                   source position and RD/WR/COMMIT are resolved by the kernel
// bit  28     LAST  last round of this target state (uniform over the lattice's G lanes)
// bit  29     FIRST first round of this target state (uniform over the lattice's G lanes)
// bit  30     WORD  ilabel != 0 (AddWordInsPenToCompactLattice applies).  Bit 30 alone is the
//                   float 2.0f: the kernel adds the penalty as fma(bits, pen/2, graph)
// bit  31     BYPASS  source = state finished by the previous round (first round, rel == 1): its
//                     value is taken from registers, not from the ring.  The sign bit: the one flag
//                     the chain wave tests every round costs it a compare and no AND
static const uint32_t kCellMask = 0xFFFu;  // ring cells + 64 dummy cells per workgroup <= 4096
static const int kWrShift = 12;
static const uint32_t kCommit = 1u << 24;
static const uint32_t kNil = 1u << 25;
static const uint32_t kOne = 1u << 26;
static const int kMaxRingCells = 4096 - 64;  // W * (64/G) may not exceed this
static const uint32_t kRelMask = (1u << 27) - 1u;  // FAR terms only
static const uint32_t kFar = 1u << 27;
static const uint32_t kLast = 1u << 28;
static const uint32_t kFirst = 1u << 29;
static const uint32_t kWord = 1u << 30;
static const uint32_t kBypass = 1u << 31;

static const int kChunkRounds = 4;    // rounds per 16-byte lane load
static const int kPrefetchChunks = 4; // register prefetch depth of the sweep kernel (chunks of 4 rounds)
static const int kDeepPrefetchChunks = 8; // the same for the deep instantiation; jobs are padded to whole multiples of it
static const int kExpMin = -(1 << 29); // exponent tag of log-zero (mantissa 0)

// One wave of the sweep kernel.
struct SweepJob {
  int64_t chunk_off;  // first unit (16 B = 4 rounds of one lane) of this job's stream; a chunk row is A * (64/G) units
  int32_t nchunks;    // multiple of kDeepPrefetchChunks
  int32_t dir;        // 0 forward, 1 backward
  int32_t lane_tab;   // first LaneInfo of this job (64/G entries)
  int32_t tag;        // bits 0-1: wave priority (long chains first); 2-7: A - 1 (stored lanes per lattice); 8-15: G; 16-27: W
};
static const int kJobPrioMask = 3;
inline int job_tag(int prio, int G, int W, int A) { return prio | ((A - 1) << 2) | (G << 8) | (W << 16); }
#if defined(__HIPCC__)
#define KLAT_HD __host__ __device__
#else
#define KLAT_HD
#endif
KLAT_HD inline int job_lanes(int tag) { return (tag >> 8) & 0xFF; }
KLAT_HD inline int job_ring(int tag) { return (tag >> 16) & 0xFFF; }
KLAT_HD inline int job_stored_lanes(int tag) { return ((tag >> 2) & 0x3F) + 1; }
// stream slot of (chunk c, lane, round r) of a job; only for lanes with lane % G < A
KLAT_HD inline int64_t stream_slot(int64_t chunk_off, int G, int A, int64_t c, int lane, int r) {
  const int LPW = 64 / G;
  return ((chunk_off + c * (int64_t)(A * LPW) + (lane % G) * LPW + lane / G) * kChunkRounds) + r;
}

// What the count pass learns about one lattice: the rounds its forward and backward stream take
// with G = 1, 2, 4 ... 64 lanes, and its longest arc span.  The layout planner (klat_pack.cc) picks
// every lattice's G from these.
static const int kNumLaneChoices = 7;
struct LatticeRounds {
  int32_t fwd[kNumLaneChoices], bwd[kNumLaneChoices];
  int32_t max_rel;
  int32_t max_terms;  // most terms of one position, forward stream in bits 0-15, backward in 16-31 (capped at 64)
  int32_t max_span;   // longest arc, in states (max_rel without the final states' distance to the end)
};

// One lattice inside a job.
struct LaneInfo {
  int64_t state_base;  // state_off[u]; -1 for an empty slot
  int32_t num_states;  // S
  int32_t lattice;     // u
};

// Work item of the posterior kernel: <=kPostItemArcs consecutive arcs of one lattice.
struct PostItem {
  int64_t arc_begin;
  int32_t count;    // | kPostItemFirst on the first item of its lattice (which also writes the lattice's totals;
                    //   a lattice without arcs has one item of count 0 for that)
  int32_t lattice;
};
static const int32_t kPostItemFirst = 1 << 30;

static const int kPostItemArcs = 1024;

}  // namespace klat
#endif
