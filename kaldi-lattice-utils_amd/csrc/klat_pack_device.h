// klat_pack_device.h -- argument block of the device-side stream packer (klat_pack_device.hip).
#ifndef KLAT_PACK_DEVICE_H_
#define KLAT_PACK_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"

namespace klat {

// one workgroup of pack_fill_kernel: lattice `lattice` (or -1: an empty slot) in slot `slot` of job `job`
struct PackEntry {
  int32_t lattice, job, slot, pad;
};

struct PackDevArgs {
  // the CSR batch
  const int64_t* state_off; const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic; const float* final_graph; const float* final_acoustic;
  // scratch: lattice u owns entries [state_off[u] + 2u, +S+2) of cnt / r0f / r0b
  int32_t* cnt;     // in-degree, then first in-arc of every state (exclusive prefix)
  int32_t* r0f;     // first forward round of position p = 0..S; [S+1] = number of rounds
  int32_t* r0b;     // first backward round of position p = 0..S-1; [S] = number of rounds
  int32_t* frank;   // [S] rank of a final state among the lattice's final states
  int32_t* in_arc;  // [A] arcs sorted by destination (indices local to the lattice)
  LatticeRounds* lat_rounds;  // [U] out of the count pass
  const int8_t* lat_g;        // [U] log2(lanes) of every lattice (the plan), in for the later passes
  int32_t* status;  // [4] error code, its lattice, longest span (max_rel), FAR terms
  // fill pass
  const PackEntry* entries;
  const SweepJob* jobs;
  int32_t* sw; float* sg; float* sa;
};

hipError_t launch_pack_count(const PackDevArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_pack_rounds(const PackDevArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_pack_fill(const PackDevArgs& a, int num_entries, hipStream_t stream);

}  // namespace klat
#endif
