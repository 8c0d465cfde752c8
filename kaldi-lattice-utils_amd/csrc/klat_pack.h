// klat_pack.h -- host-side construction of the lockstep arc streams (flatten stage).
#ifndef KLAT_PACK_H_
#define KLAT_PACK_H_
#include <stdint.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/klat.h"
#include "klat_device.h"

namespace klat {

// Uninitialised host array: the streams are hundreds of megabytes, and std::vector's value
// initialisation would touch (page-fault) all of it on one thread before the packer's threads
// overwrite it -- that fault-in, not the packing, dominated klat_batch_create.
template <class T>
class HostBuf {
 public:
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { std::free(p_); }
  bool resize(size_t n) {  // contents unspecified
    std::free(p_);
    p_ = n ? static_cast<T*>(std::malloc(n * sizeof(T))) : nullptr;
    n_ = p_ ? n : 0;
    return n == 0 || p_ != nullptr;
  }
  T* data() { return p_; }
  const T* data() const { return p_; }
  size_t size() const { return n_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

struct PackedBatch {
  int G = 0, W = 0;
  HostBuf<int32_t> sw;  // [chunks][64][4]
  HostBuf<float> sg, sa;
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

// Job slicing shared by the host and the device packer: lattices sorted by round count, 64/G per
// job, jobs longest first; fills out->jobs / lanes / slot totals / max_rounds, returns the number
// of chunks and, per job, its lattices in slot order.
int64_t slice_jobs(const int64_t* state_off, int64_t num_lattices, const std::vector<int64_t>& rounds_fwd,
                   const std::vector<int64_t>& rounds_bwd, int G, int W, PackedBatch* out,
                   std::vector<std::vector<int64_t>>* job_lats);

// Heuristics used when the options leave G / W at 0.
int choose_lanes_per_lattice(int64_t num_lattices, int64_t num_states, int64_t num_arcs);
int choose_ring_slots(int G, int64_t max_rel);

}  // namespace klat
#endif
