// klat_level.h -- argument blocks and launchers of the level-synchronous forward-backward
// (klat_level.hip): the path for lattices whose topological depth is far below their states.
#ifndef KLAT_LEVEL_H_
#define KLAT_LEVEL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_kernels.h"

namespace klat {

static constexpr int kLevelMinStates = 64;    // smaller lattices are no work either way
static constexpr int kLevelMaxStates = 8192;  // 8 B of LDS per state and wave
static constexpr int kLevelMinWidth = 8;      // states per level on average below which the streams win

// A lattice's LEVEL STREAM: one linear run of dwords, read once front to back -- the forward
// records (levels first to last), then the backward records (levels last to first).  A record is up
// to 64 states of one level (one per lane) with everything they fold:
//   [0] n  [1] nt  | n state numbers | n + 1 term offsets (0 .. nt) | backward only: n final weights
//   (graph, acoustic) | nt terms of 3 dwords
// A term is an arc as the folding state sees it: the state at its other end (forward: the source,
// backward: the destination; bit 31: ilabel != 0), graph cost, acoustic cost.
static constexpr int kLevelBlockDwords = 256;     // the stream moves to LDS in 1-KiB blocks (64 lanes x 16 B)
static constexpr int kLevelRingBlocks = 64;       // LDS ring of the sweep
static constexpr int kLevelMaxRecordTerms = 2600; // a record (3 + 4 n + 3 nt dwords at most) stays below half the ring

struct LevelPlanArgs {
  const int64_t* state_off;
  const int64_t* arc_off;
  const int32_t* src;
  const int32_t* dst;
  const int32_t* cnt;   // [S + 2 per lattice] first in-arc of every state (the stream packer's count pass)
  int32_t* counters;    // [S + 2 per lattice] scratch
  int32_t* order;       // [S] states level by level
  int32_t* level_off;   // [S + 2 per lattice] first entry of every level in `order`
  const int32_t* out_begin;  // [S + 2 per lattice] first out-arc of every state (the count pass's row table)
  int32_t* depth;       // [U] number of levels, or -1: not for this path
  int32_t* records;     // [U] records per direction (sum over the levels of ceil(states / 64))
  const int32_t* status;  // [2] the count pass's verdict: non-zero, a lattice of the batch is malformed and nothing is planned
  const uint8_t* only;    // [U] or null: plan only these lattices (the others have no tables of the general count pass)
};

// Writes the level streams of the lattices on the level path (after level_plan_kernel).
struct LevelBuildArgs {
  const int32_t* lattices;
  const int64_t* state_off;
  const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic;
  const float* final_graph; const float* final_acoustic;
  const int32_t* cnt; const int32_t* in_arc; const int32_t* order; const int32_t* level_off; const int32_t* out_begin;
  const int32_t* depth;
  const int64_t* stream_off;  // [level lattices] first block of the k-th lattice's stream
  int32_t* stream;
};

struct LevelSweepArgs {
  const int32_t* lattices;  // the lattices on this path
  const int64_t* state_off;
  const int64_t* arc_off;
  const float* final_graph; const float* final_acoustic;
  const int32_t* records;     // [U]
  const int64_t* stream_off;  // [level lattices], in blocks
  const int32_t* stream;
  double* alpha;  // padded rows (klat_kernels.h); Viterbi: forward costs
  double* beta;
  ScaleParams sp;
};

// dwords of a lattice's stream (both directions)
__host__ __device__ static inline int64_t level_stream_dwords(int64_t records, int64_t states, int64_t arcs) { return 2 * (3 * records + 2 * states + 3 * arcs) + 2 * states; }

hipError_t launch_level_plan(const LevelPlanArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_level_build(const LevelBuildArgs& a, int num_level_lattices, hipStream_t stream);
size_t level_sweep_lds_bytes(int max_states);  // LDS of one wave (= one lattice) of the sweep
hipError_t launch_level_sweep(const LevelSweepArgs& a, int num_level_lattices, int max_states, bool tropical, hipStream_t stream);

}  // namespace klat
#endif
