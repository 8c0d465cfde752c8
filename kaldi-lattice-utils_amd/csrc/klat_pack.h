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

// Jobs [job_begin, job_end) share one layout: G lanes per lattice, W ring slots.  One sweep launch
// per class, the class with the longest chain first.
struct LayoutClass {
  int32_t G, W, job_begin, job_end;
  int32_t far;  // some lattice of the class has arcs (or final states) W or more positions long
  int32_t chain;  // the class holds the lattices that klat_chain.hip sweeps in segments: the log-semiring sweep leaves it out
};

// The posterior kernel's work items of lattice u (arcs [a0, a1)): at least one, the first one marked.
void append_post_items(std::vector<PostItem>* items, int64_t a0, int64_t a1, int64_t u);

struct PackedBatch {
  int G = 0, W = 0;  // of the class that holds most of the stream (klat_batch_info)
  std::vector<LayoutClass> classes;
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

// Layout planning and job slicing, shared by the host and the device packer.  plan_layout picks
// log2(lanes) of every lattice (-1: no states) from the round counts of the count pass; build_jobs
// groups the lattices by lanes into layout classes, slices every class into jobs (lattices sorted
// by round count, 64/G per job, longest first), fills out->classes / jobs / lanes / slot totals /
// max_rounds and returns the number of chunks and, per job, its lattices in slot order.
void plan_layout(int64_t num_lattices, const int64_t* state_off, const int64_t* arc_off, const LatticeRounds* rounds,
                 const klat_pack_options& opt, std::vector<int8_t>* lat_g,
                 const uint8_t* skip = nullptr /* [num_lattices] lattices that take no stream (level path) */);
// What plan_layout starts from, known from the offsets alone (the lean count pass computes every lattice's rounds for it
// before the plan exists): the lattice's fanout rounded up to a power of two, halved while the batch would need more
// than 8192 workgroups; a caller's fixed lanes.
void predict_lanes(int64_t num_lattices, const int64_t* state_off, const int64_t* arc_off, const klat_pack_options& opt,
                   std::vector<int8_t>* lat_g, const uint8_t* skip = nullptr);
int build_jobs(int64_t num_lattices, const int64_t* state_off, const LatticeRounds* rounds, const std::vector<int8_t>& lat_g,
               const klat_pack_options& opt, PackedBatch* out, std::vector<std::vector<int64_t>>* job_lats,
               int64_t* chunk_total, std::string* err,
               const uint8_t* on_chain = nullptr /* [num_lattices] lattices of the segment path: layout classes of their own */);
// The planner's model of a batch's sweep: cycles of the plan plan_layout would choose for the lattices not
// in `skip` (what klat_batch_create weighs the segment path against).
double plan_cycles(int64_t num_lattices, const int64_t* state_off, const int64_t* arc_off, const LatticeRounds* rounds,
                   const klat_pack_options& opt, const uint8_t* skip);
// log2(lanes) the planner starts from: the lattice's fanout rounded up to a power of two
int base_lane_choice(int64_t states, int64_t arcs);

// Ring slots when the options leave W at 0.
int choose_ring_slots(int G, int64_t max_rel);

}  // namespace klat
#endif
