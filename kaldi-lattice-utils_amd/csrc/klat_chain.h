// klat_chain.h -- the forward-backward of ONE long lattice cut into segments (klat_chain.hip).
//
// The lockstep streams deal a lattice state by state: S states are S dependent rounds, and a batch
// whose longest lattices are ten times its mean (BASELINE configs[4]: sizes log-uniform in 200 ..
// 20 000 states) waits for those few chains with the chip nine tenths idle.  ComputeLatticeAlphasAndBetas
// (called at latbin/lattice-to-word-frame-post.cc:88-90) is linear in the (+, x)
// semiring, so a lattice whose arcs span at most L states can be cut at any state b: everything behind
// the cut depends on what lies before it only through the values of the L states b-L .. b-1.
//
//   A  transfer   for every segment k and each of the c >= L states u before its first state: the weight
//                 of all paths from u to each of the segment's LAST c states whose first arc enters the
//                 segment (matrix Q_k, c x c), and to any final state inside it (Qf_k).  One LANE per
//                 (segment, u): all lanes of a wave walk their segments' arc records in lockstep, no
//                 cross-lane traffic, the last W states of a lane in its own LDS column.
//   B  combine    per lattice, serially over its segments, two small chains of c x c log-semiring
//                 matrix-vector products: forward alpha(last c states of segment k) = alpha(k-1) Q_k and
//                 tot_forward; backward the crossing contributions y_k = Q_k y_(k+1) + Qf_k.
//   C  sweep      every segment again, now one lane per (segment, direction), with the values behind its
//                 boundary known: alpha / beta of all its states, natural log, into the batch's rows.
//
// The chain is 2 x (segment length) arcs instead of S states; the work is (c + 2) / 2 times the arcs.
#ifndef KLAT_CHAIN_H_
#define KLAT_CHAIN_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"

namespace klat {

// An arc RECORD is the double exp(-cost) with the low nine bits of its fraction given to control bits
// (the weight keeps 43 fraction bits: 5.7e-14 relative, below 1e-9 in the log of a 20 000-arc path).
static const int kChainMaxSpan = 16;               // longest arc of a lattice that may be cut
static const uint32_t kChainRelMask = 31u;         // bits 0-4: states between the term's source and its target
static const uint32_t kChainLast = 1u << 5;        // last term of its target state
static const uint32_t kChainSpecial = 1u << 6;     // constant source: one (alpha[start], final weights) ...
static const uint32_t kChainWhich = 1u << 7;       // ... or zero (a state without a term, a dead arc)
static const uint32_t kChainFinal = 1u << 8;       // forward only, after the LAST of a final state: its final weight
static const uint32_t kChainCtlMask = 0x1FFu;
static const double kChainMaxCost = 690.0;         // |cost| above this does not fit the record's double

// why a sweep fell back to the streams (ChainArgs::flag, klat_batch_chain_fallback)
static const int kChainTroubleRange = 1;    // a lane's values left the doubles' range
static const int kChainTroubleSpread = 2;   // a live value more than 2^400 below its lane's largest
static const int kChainTroubleOne = 4;      // a constant term (a final weight) 2^900 away from its lane's scale
static const int kChainTroubleInject = 8;   // the same for the values injected at a cut
static const int kChainTroubleRows = 16;    // combine: the rows of a transfer matrix too far apart
static const int kChainTroubleVector = 32;  // combine: the entries of a boundary vector too far apart

struct ChainLattice {
  int32_t lattice, num_states, seg_begin, num_segs;
  uint32_t rec_f, rec_b;  // first forward / backward record
  int32_t span;           // the lattice's longest arc, in states
  int32_t grp_begin, num_grps, pad;  // its groups of kChainGroup segments (the two-level combine)
};
static const int kChainGroup = 16;
struct ChainGroup {
  int32_t clat, seg_first, nseg, pad;
};
struct ChainSegment {  // (filled by chain_prep_kernel)
  int32_t clat, k, first_state, num_states;
  uint32_t rf_begin, rf_count, rb_begin, rb_count;
};

struct ChainPrepArgs {
  const ChainLattice* lats;
  ChainSegment* segs;
  uint64_t* recs;
  const int64_t* state_off; const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic; const float* final_graph; const float* final_acoustic;
  // the stream packer's count pass: lattice u owns entries [state_off[u] + 2u, +S+2)
  const int32_t* cnt; const int32_t* row; const int32_t* in_arc; const int32_t* out_arc;
  int32_t* scratch_f; int32_t* scratch_b;  // same shape (the packer's r0f / r0b, not yet in use)
  uint64_t* arc_records;                  // [A] a temporary: every arc's record in the arcs' own order (chain_prep_kernel)
  ScaleParams sp;
  int32_t tropical;  // the records hold the costs themselves (klat_viterbi), not exp(-cost)
  int32_t* status;  // [1] set when a cost does not fit a record
  unsigned long long* maxcost;  // [1] the largest finite |cost| among the records, as the bits of a double (atomicMax)
};

struct ChainArgs {
  const ChainLattice* lats;
  const ChainSegment* segs;
  const uint64_t* recs;
  int32_t num_lats, num_segs;
  int32_t c, lgc, W;  // states behind a cut (power of two >= the longest arc), ring slots (2c)
  // Values between the kernels stay scaled: a plain double times two to an exponent.
  double* Qm; int32_t* Qe;        // [segments][c][c] rows of Q to [segments][c] exponents (one per row)
  double* Qfm; int32_t* Qfe;      // [segments][c] Qf, mantissa and exponent
  double* bnd_fm; int32_t* bnd_fe;  // [segments][c] alpha of the c states before the segment, to [segments] exponents
  double* bnd_bm; int32_t* bnd_be;  // [segments][c] what the arcs leaving the segment contribute to its last c states
  const ChainGroup* grps; int32_t num_grps;
  double* Gm; int32_t* Ge; double* Gfm; int32_t* Gfe;  // [groups]...: the product of a group's matrices, its final-state vector
  double* gb_fm; int32_t* gb_fe; double* gb_bm; int32_t* gb_be;  // [groups][c]: the running vector before / behind every group
  int32_t rescale_iters;  // a lane rescales every this many iterations of four records
  unsigned long long* dbg = nullptr;  // (KLAT_CHAIN_DEBUG) four words per wave: begin, end (100 MHz), HW_ID, XCC_ID
  size_t dbg_stride = 0;              //   the sweep's waves start here
  int32_t* flag;          // [1] raised when a lattice's values do not fit this arithmetic: the streams sweep them again
  int32_t* flag_next = nullptr;  // [1] or null: the NEXT sweep's flag, cleared by this one's first kernel (two flags take turns: no memset between sweeps)
  const int64_t* state_off;
  double* alpha; double* beta;  // padded rows (klat_kernels.h)
  const LnTabEntry* ln_table;
};

hipError_t launch_chain_prep(const ChainPrepArgs& a, int num_chain_lattices, hipStream_t stream);
// transfer, combine and sweep, one after the other on `stream`; tropical: the (min,+) twin over cost records -- alpha /
// beta are then the padded rows of the Viterbi forward / backward costs
hipError_t launch_chain_sweep(const ChainArgs& a, hipStream_t stream, bool tropical = false);

}  // namespace klat
#endif
