// klat_pack_device.h -- argument block of the device-side stream packer (klat_pack_device.hip).
#ifndef KLAT_PACK_DEVICE_H_
#define KLAT_PACK_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"

namespace klat {

// A tile of the fill pass: this many rounds of a job's stored lanes, one contiguous piece of each
// stream array.  A job's length is a whole number of tiles (kDeepPrefetchChunks chunks of 4 rounds).
static constexpr int kPackTileRounds = kDeepPrefetchChunks * kChunkRounds;
static constexpr int kPackTilesPerGroup = 8;  // tiles one workgroup of pack_fill_kernel writes, one after the other

// one workgroup of pack_fill_kernel: tiles [tile_begin, tile_begin + tile_count) of job `job`
struct PackEntry {
  int32_t job, tile_begin, tile_count, pad;
};

// LDS the count pass needs for a lattice's tables (in- and out-degree prefix: 2 (S + 2) words; arcs
// by destination as 16-bit arc numbers); above every class for a lattice of 65 536 arcs or more
static constexpr int kPackCountMaxLds = 150 * 1024;
static constexpr int kPackCountClasses = 3;
static constexpr int kPackCountClassLds[kPackCountClasses] = {36 * 1024, 76 * 1024, kPackCountMaxLds};  // 4, 2, 1 workgroups per CU
__host__ __device__ static inline int pack_count_lds_bytes(int64_t S, int64_t A) {
  if (A >= 65536 || S >= (1 << 20)) return 0x7FFFFFFF;
  return (int)(8 * (S + 2) + 2 * ((A + 1) & ~(int64_t)1));
}

// The lean passes (klat_pack_lean.hip), for lattices whose arcs are at most kPackLeanMaxSpan states long.  LDS of the
// lean count pass: in-degrees (4 (S + 2) bytes), first out-arcs and arcs by destination as 16-bit numbers, two bits a
// state, a byte an arc; three classes again (256 / 512 / 1 024 threads: sixteen waves a CU in each), a lattice above the
// last one takes the general passes.
static constexpr int kPackLeanClasses = 3;
static constexpr int kPackLeanClassLds[kPackLeanClasses] = {31 * 1024 + 512, 79 * 1024, kPackCountMaxLds};  // 5, 2, 1 workgroups per CU
__host__ __device__ static inline int pack_lean_lds_bytes(int64_t S, int64_t A) {
  if (A >= 65536 || S >= (1 << 20)) return 0x7FFFFFFF;
  return (int)(4 * (S + 2) + 8 * ((S + 31) / 32 + 1) + 2 * ((A + 1) & ~(int64_t)1) + 2 * ((S + 3) & ~(int64_t)1) + ((A + 3) & ~(int64_t)3));
}
// A forward tile of the lean fill pass reads the arcs of `longest arc` more states than it holds positions.
static constexpr int kPackLeanMaxSpan = 64;

// tiles one workgroup of pack_fill_lean_kernel writes (C2 fill, ms: 4 tiles 0.856, 8: 0.817, 16: 0.798, 24: 0.866, 36: 0.912)
#ifndef KLAT_LEAN_TILES_PER_GROUP
#define KLAT_LEAN_TILES_PER_GROUP 16
#endif
static constexpr int kLeanTilesPerGroup = KLAT_LEAN_TILES_PER_GROUP;
// rounds of a tile of the lean fill pass: kPackTileRounds (32) or twice that
#ifndef KLAT_LEAN_TILE_ROUNDS
#define KLAT_LEAN_TILE_ROUNDS 32
#endif
static constexpr int kLeanTileRounds = KLAT_LEAN_TILE_ROUNDS;
static_assert(kLeanTileRounds % kPackTileRounds == 0, "the tile tables hold a position per kPackTileRounds rounds");
// one workgroup of pack_fill_lean_kernel: tiles [tile_begin, tile_begin + tile_count) of a job, with what it needs of the job
struct LeanEntry {
  int64_t chunk_off;
  int32_t lane_tab, tag, dir, tile_begin, tile_count, job;
  int32_t rounds, pad;  // the job's rounds (a multiple of kPackTileRounds: its last tile may be a short one)
};

struct PackDevArgs {
  // the CSR batch
  const int64_t* state_off; const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic; const float* final_graph; const float* final_acoustic;
  // tables: lattice u owns entries [state_off[u] + 2u, +S+2) of cnt / row / r0f / r0b
  int32_t* cnt;     // first in-arc of every state (exclusive prefix of the in-degrees; [S] = A); [S+1] = A + number of final states
  int32_t* row;     // first out-arc of every state ([S] = A)
  int32_t* r0f;     // first forward round of position p = 0..S; [S+1] = number of rounds
  int32_t* r0b;     // first backward round of position p = 0..S-1 (bit 31: its state S-1-p is final); [S] = number of rounds
  int32_t* flist;   // [S] the lattice's final states in state order (cnt[S+1] - cnt[S] of them)
  int32_t* in_arc;  // [A] arcs sorted by destination (indices local to the lattice)
  int32_t* out_arc; // [A] the out-arcs of every state in stream order: farthest destination first, equal destinations in arc order
  LatticeRounds* lat_rounds;  // [U] out of the count pass
  const int8_t* lat_g;        // [U] log2(lanes) of every lattice (the plan), in for the later passes
  int32_t* status;  // [4] error code, its lattice, longest span (max_rel), FAR terms
  // fill pass
  const PackEntry* entries;
  const SweepJob* jobs;
  const LaneInfo* lanes;
  int32_t* sw; float* sg; float* sa;
  // fused costs (klat_pack_options.fuse_scales): sg holds (float)cost, sl the residual in units of lo_unit
  int16_t* sl;
  int32_t fused;
  ScaleParams fsp;
  double lo_inv_unit;
  unsigned long long* maxcost;  // [1] bits of the largest finite |cost| of the batch (pack_maxcost_kernel)
  unsigned long long* prof;  // [8] or null: phase cycles of the count pass (KLAT_PACK_PROFILE=1)
  // general passes over a part of the batch: only the lattices with only[u] != 0 (null: every lattice)
  const uint8_t* only;
  // lean passes (klat_pack_lean.hip); r0f / r0b then hold a position's first round in bits 0-23, its terms in bits 24-30
  const int8_t* lat_g_pred;  // [U] log2(lanes) the plan is expected to choose: the lean count pass leaves r0f / r0b for it
  int64_t num_arcs;          // of the batch
  uint32_t* meta;            // [A] arc length | place among its destination's in-arcs << 8 | among its source's out-arcs << 16 | word << 24
  int32_t* tpf;              // tile tables: the position that holds round 32 m of lattice u at [(state_off[u] >> 3) + 2 u + m]
  int32_t* tpb;
  const uint8_t* lean_skip;  // [U] or null: lattices the lean count pass leaves alone whatever their size (they take the general passes)
  uint8_t* lean_ok;          // [U] out: the lean count pass has packed the lattice's tables (0: it is left to the general passes)
  const uint8_t* reround;    // [U] pack_rounds_lean_kernel: the lattices whose lanes are not the expected ones
  int32_t fold_maxcost;      // the lean count pass also finds the batch's largest cost (a.maxcost): no pack_maxcost_kernel
  int32_t level_probe;       // lattices the level path may take (klat_level.hip reads the general tables) are left alone
  const LeanEntry* lentries;
};

// class_need[c]: the largest pack_count_lds_bytes among the lattices of LDS class c (0: none);
// any_global: some lattice is above the last class
hipError_t launch_pack_count(const PackDevArgs& a, int num_lattices, const int class_need[kPackCountClasses], bool any_global,
                             hipStream_t stream);
// the largest finite cost of the batch under a.fsp (arcs and final weights)
hipError_t launch_pack_maxcost(const PackDevArgs& a, int64_t num_arcs, int64_t num_states, hipStream_t stream);
// The lean passes: count (validation, tables, rounds of every lane count, the batch's largest cost when a.fused, r0f / r0b
// for a.lat_g_pred; class_need over kPackLeanClassLds), the prefix sums again for a.reround, fill over a.lentries.
hipError_t launch_pack_count_lean(const PackDevArgs& a, int num_lattices, const int class_need[kPackLeanClasses], hipStream_t stream);
hipError_t launch_pack_rounds_lean(const PackDevArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_pack_fill_lean(const PackDevArgs& a, const int count[3 * kNumLaneChoices] /* entries per (log2(lanes), forward | backward | both side by side), a.lentries sorted that way */, hipStream_t stream);
hipError_t launch_pack_rounds(const PackDevArgs& a, int num_lattices, hipStream_t stream);
// max_lpw: the most lattices any job holds (64 / its lanes per lattice)
hipError_t launch_pack_fill(const PackDevArgs& a, int num_entries, int max_lpw, hipStream_t stream);

}  // namespace klat
#endif
