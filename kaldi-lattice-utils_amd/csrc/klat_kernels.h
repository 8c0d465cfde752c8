// klat_kernels.h -- kernel argument blocks and host-callable launchers (HIP only).
#ifndef KLAT_KERNELS_H_
#define KLAT_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"

namespace klat {

struct ScaleParams {
  float gs_f, as_f;  // graph_scale, acoustic_scale (BaseFloat in the reference)
  float pen;
  int scaled;     // graph_scale != 1 || acoustic_scale != 1 (the reference's own test)
  int penalized;  // insertion_penalty != 0
};

struct SweepArgs {
  const SweepJob* jobs;
  const LaneInfo* lanes;
  const int32_t* sw;  // stream term words
  const float* sg;    // stream graph costs
  const float* sa;    // stream acoustic costs
  double* alpha;      // [S] natural-log forward values  (vector<double>* alpha of B2)
  double* beta;       // [S] natural-log backward values (vector<double>* beta of B2)
  double* tot_fwd;    // [U] forward total: the value of the virtual super-final state
  int32_t* job_flags;  // [num_jobs] set by the fast sweep when a job needs the exact sweep
  ScaleParams sp;
  int32_t ring_slots;
};

struct TotalsArgs {
  int32_t num_lattices;
  const int64_t* state_off;
  const double* beta;
  double* tot_fwd;  // in: as written by the sweep; rewritten to 0 for empty lattices
  double* total; double* tot_bwd;
};

struct PosteriorArgs {
  const PostItem* items;
  const int64_t* state_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic;
  const double* alpha; const double* beta;
  const double* total;
  double* logpost;
  ScaleParams sp;
};

hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream);
hipError_t launch_totals(const TotalsArgs& a, hipStream_t stream);
hipError_t launch_posterior(const PosteriorArgs& a, int num_items, hipStream_t stream);

}  // namespace klat
#endif
