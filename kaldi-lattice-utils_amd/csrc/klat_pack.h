// klat_pack.h -- host-side construction of the lockstep arc streams (flatten stage).
#ifndef KLAT_PACK_H_
#define KLAT_PACK_H_
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/klat.h"
#include "klat_device.h"

namespace klat {

struct PackedBatch {
  int G = 0, W = 0;
  std::vector<int32_t> sw;  // [chunks][64][4]
  std::vector<float> sg, sa;
  std::vector<SweepJob> jobs;    // sorted longest first
  std::vector<LaneInfo> lanes;
  std::vector<PostItem> items;
  int64_t fwd_slots = 0, bwd_slots = 0, far_terms = 0;
  int32_t max_rounds = 0;
};

// Validates topology (dst > src, ids in range, arcs grouped by src) and builds the streams.
// Returns KLAT_OK or an error code with *err filled.
int pack_batch(const klat_batch_desc& d, const klat_pack_options& opt, PackedBatch* out,
               std::string* err);

// Heuristics used when the options leave G / W at 0.
int choose_lanes_per_lattice(int64_t num_lattices, int64_t num_states, int64_t num_arcs);
int choose_ring_slots(int G, int64_t max_rel);

}  // namespace klat
#endif
