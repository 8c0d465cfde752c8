// klat_nbest.h -- argument block and launcher of the n-best kernel (klat_nbest.hip).
#ifndef KLAT_NBEST_H_
#define KLAT_NBEST_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klat {

static constexpr int kNbestWidth = 128;  // partial paths kept per state (= KLAT_NBEST_MAX)

struct NbestArgs {
  int32_t num_lattices;
  int32_t* next_lattice;  // work queue head, zeroed before the launch
  const int64_t* state_off;
  const int64_t* arc_off;
  const int32_t* src;
  const int32_t* dst;
  const int32_t* olabel;
  const float* final_graph;
  const float* final_acoustic;
  const float* arc_cost;  // [A] tropical costs, >= 0; +inf removes the arc
  int32_t k, max_labels;
  uint64_t* scratch;  // [workgroups][scratch_states][kNbestWidth]
  int64_t scratch_states;
  int32_t* num_paths;    // [U]
  float* path_cost;      // [U*k]
  int32_t* path_len;     // [U*k]
  int32_t* path_labels;  // [U*k*max_labels]
  int32_t* status;       // 0; 1+u: lattice u has a negative / NaN cost; -(1+u): a path of u has more labels than max_labels
};

hipError_t launch_nbest(const NbestArgs& a, int num_workgroups, hipStream_t stream);

}  // namespace klat
#endif
