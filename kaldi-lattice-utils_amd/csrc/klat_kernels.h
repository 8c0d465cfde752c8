// klat_kernels.h -- kernel argument blocks and host-callable launchers (HIP only).
#ifndef KLAT_KERNELS_H_
#define KLAT_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"

namespace klat {

struct ScaleParams {
  double gs, as;  // (double)(float)graph_scale, acoustic_scale
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
  double* fm; int32_t* fe;  // forward values per state (mantissa, exponent)
  double* bm; int32_t* be;  // backward values per state
  double* totf_m; int32_t* totf_e;  // forward total per lattice (virtual super-final state)
  ScaleParams sp;
  int32_t ring_slots;
};

struct TotalsArgs {
  int32_t num_lattices;
  const int64_t* state_off;
  const double* totf_m; const int32_t* totf_e;
  const double* bm; const int32_t* be;
  double* total; double* tot_fwd; double* tot_bwd;
};

struct PosteriorArgs {
  const PostItem* items;
  const int64_t* state_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic;
  const double* fm; const int32_t* fe;
  const double* bm; const int32_t* be;
  const double* total;
  double* logpost;
  ScaleParams sp;
};

struct ConvertArgs {
  int64_t num_states;
  const double* fm; const int32_t* fe;
  const double* bm; const int32_t* be;
  double* alpha; double* beta;
};

hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream);
hipError_t launch_totals(const TotalsArgs& a, hipStream_t stream);
hipError_t launch_posterior(const PosteriorArgs& a, int num_items, hipStream_t stream);
hipError_t launch_convert(const ConvertArgs& a, hipStream_t stream);

}  // namespace klat
#endif
