// lattice-functions.h -- host-side mirror of the Kaldi functions the reference tools call on the
// hot path, with the reference's own names and argument meaning (interface B2), on
// top of the C ABI of include/klat.h.  The integer graph work (top-sort, state times, length
// disambiguation, flattening) runs on the host; everything numeric runs on the GPU.
#ifndef KLAT_HOST_LATTICE_FUNCTIONS_H_
#define KLAT_HOST_LATTICE_FUNCTIONS_H_
#include <cstdint>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "../../include/klat.h"
#include "archive-index.h"
#include "compact-lattice.h"

namespace klat {

// TopSortCompactLatticeIfNeeded (latbin/lattice-to-word-frame-post.cc:80): if some arc has
// nextstate <= state or Start() != 0, renumber states in topological order (reverse DFS finishing
// order, start first).  Throws (KLAT_ERR) on a cyclic lattice, as the reference does.
void TopSortCompactLatticeIfNeeded(CompactLattice* clat);

// CompactLatticeStateTimes (latbin/lattice-to-word-frame-post.cc:84-85): times[0]=0,
// times[dst]=times[src]+len(arc string); returns the utterance length (max over final states of
// times[s]+len(final string); 0 with a warning if there is no final state).  Throws if the
// lattice is not top-sorted or the times are inconsistent (KALDI_ERR / KALDI_ASSERT upstream).
int32_t CompactLatticeStateTimes(const CompactLattice& clat, std::vector<int32_t>* times);

// DisambiguateStateInputSequenceLength (fstext/fstext-utils2.h:109-215): equivalent lattice in
// which all paths reaching a state carry the same number of non-epsilon labels; new state ids are
// the ranks of the (length, old state) pairs; returns the maximum length.
int32_t DisambiguateStateInputSequenceLength(const CompactLattice& ifst, CompactLattice* ofst,
                                             std::vector<int32_t>* state_input_length, bool use_input = false);

// ArcSort(&clat, OLabelCompare()) as latbin/lattice-best-path2.cc:103 applies it before the
// disambiguation: the arcs of every state ordered by output label (stable).
void ArcSortByOlabel(CompactLattice* clat);

// AddSequenceLengthDismabiguationSymbol (fstext/fstext-utils2.h:219-271): appends the auxiliary
// chain aux[0..max_len] (aux[k] -> aux[k+1] with label dis_label and weight One, aux[max_len]
// final), and replaces every final weight by an epsilon arc carrying it to aux[length of the
// state], so that every complete path has max_len labels.  state_input_length grows by the
// auxiliary states (length k for aux[k]).
void AddSequenceLengthDisambiguationSymbol(CompactLattice* clat, std::vector<int32_t>* state_input_length,
                                           int32_t dis_label = -1);

// PruneLattice(beam, &clat) as the kwsbin2 tools apply it after ScaleLattice and
// AddWordInsPenToCompactLattice (kwsbin2/lattice-word-index-segment.cc:41-51): Viterbi forward /
// backward costs (structure mirrored in-repo at latbin/lattice-prune-dyn-beam.cc:35-87), final
// weights and arcs whose best path through them costs more than best + beam are dropped, then
// Connect() removes states that are no longer accessible and co-accessible (ids stay in order).
// The lattice keeps its ORIGINAL costs (the GPU applies the scales later); the pruning decision
// uses the costs the reference would see, i.e. scaled and penalised with its float rounding.
// Host-side graph surgery.  Returns false if the result is empty.
bool PruneLattice(float beam, float graph_scale, float acoustic_scale, float insertion_penalty,
                  CompactLattice* clat);

// ConvertToCost of a weight as the reference holds it after ScaleLattice / AddWordInsPenToCompactLattice.
double ScaledCost(float graph, float acoustic, bool word, float graph_scale, float acoustic_scale, float insertion_penalty);

// Connect() after pruning: keep[s][i] == 0 drops arc i of state s; states that are no longer
// accessible and co-accessible go, ids stay in order.  Returns false (and empties the lattice) if
// nothing survives.
bool ConnectKeeping(const std::vector<std::vector<uint8_t>>& keep, CompactLattice* clat);

// A batch of top-sorted lattices flattened to the CSR/SoA layout of klat_batch_desc.
// A growable array of plain values whose resize() leaves the new elements unset: the arrays of a batch are
// hundreds of megabytes that AddMany's threads overwrite at once, and std::vector would zero them on one
// thread first.
template <class T>
class PodVec {
 public:
  PodVec() = default;
  PodVec(const PodVec&) = delete;
  PodVec& operator=(const PodVec&) = delete;
  PodVec(PodVec&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr, o.n_ = o.cap_ = 0; }
  PodVec& operator=(PodVec&& o) noexcept {
    if (this != &o) {
      std::free(p_);
      p_ = o.p_, n_ = o.n_, cap_ = o.cap_;
      o.p_ = nullptr, o.n_ = o.cap_ = 0;
    }
    return *this;
  }
  ~PodVec() { std::free(p_); }
  void clear() { n_ = 0; }
  size_t size() const { return n_; }
  T* data() { return p_; }
  const T* data() const { return p_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
  void resize(size_t n) {  // new elements are NOT initialised
    if (n > cap_) grow(n);
    n_ = n;
  }
  void push_back(T v) {
    if (n_ == cap_) grow(n_ + 1);
    p_[n_++] = v;
  }

 private:
  void grow(size_t need) {
    size_t c = cap_ ? cap_ * 2 : 1024;
    if (c < need) c = need;
    T* q = static_cast<T*>(std::realloc(p_, c * sizeof(T)));
    if (!q) throw std::bad_alloc();
    p_ = q;
    cap_ = c;
  }
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

class LatticeBatch {
 public:
  void Clear();
  // appends clat (must be top-sorted with Start()==0, or empty); times may be null
  void Add(const CompactLattice& clat, const std::vector<int32_t>* times);
  // the same for many lattices at once, their arrays filled on `threads` threads (a push_back per field and
  // arc flattened 30 M arcs/s; the arcs of a C2-sized batch are 60 M)
  void AddMany(const std::vector<const CompactLattice*>& lats, const std::vector<const std::vector<int32_t>*>& times, int threads);
  // the same from the flat form (archive-index.h); every lattice top-sorted from state 0
  void AddManyFlat(const std::vector<FlatLattice>& lats, const std::vector<std::vector<int32_t>>& times, int threads);
  int32_t NumLattices() const { return (int32_t)state_off_.size() - 1; }
  int64_t NumArcs() const { return arc_off_.back(); }
  int64_t NumStates() const { return state_off_.back(); }
  klat_batch_desc Desc() const;
  const std::vector<int64_t>& StateOff() const { return state_off_; }
  const std::vector<int64_t>& ArcOff() const { return arc_off_; }

 private:
  std::vector<int64_t> state_off_{0}, arc_off_{0};
  PodVec<int32_t> src_, dst_, ilabel_, olabel_, state_time_;
  PodVec<float> graph_, acoustic_, final_graph_, final_acoustic_;
  bool has_times_ = true;
};

// RAII over klat_ctx / klat_batch; failures become exceptions carrying klat_last_error().
class Context {
 public:
  explicit Context(int device = 0);
  ~Context();
  klat_ctx* get() const { return ctx_; }

 private:
  klat_ctx* ctx_ = nullptr;
};
class DeviceBatch {
 public:
  DeviceBatch(Context& ctx, const LatticeBatch& batch, const klat_pack_options* opt = nullptr);
  ~DeviceBatch();
  klat_batch* get() const { return b_; }
  void ForwardBackward(float graph_scale, float acoustic_scale, float insertion_penalty);

 private:
  klat_batch* b_ = nullptr;
};
void Check(int rc, const char* what);

// PruneLattice for a whole batch with the decisions taken on the GPU (klat_viterbi +
// klat_prune_mask: the tropical twin of the sweep); the masks are applied and the
// lattices Connect()ed here.  Same result as PruneLattice() lattice by lattice, except for arcs
// whose best path lies within rounding of the cutoff.
void PruneLatticesOnDevice(Context& ctx, float beam, float graph_scale, float acoustic_scale, float insertion_penalty,
                           std::vector<CompactLattice>* lats);

// ComputeLatticeAlphasAndBetas (latbin/lattice-to-word-frame-post.cc:88-90): same signature and
// return value (0.5*(tot_forward+tot_backward)); viterbi = true (no call site of the reference
// passes it) runs the (min,+) twin of the sweep.  Runs the lattice as a one-element batch on the calling thread's GPU
// context; batch callers should use LatticeBatch/DeviceBatch directly.
double ComputeLatticeAlphasAndBetas(const CompactLattice& clat, bool viterbi, std::vector<double>* alpha,
                                    std::vector<double>* beta);
// The GPU the contexts behind ComputeLatticeAlphasAndBetas are created on (default 0); call before
// the first use.
void SetDefaultDevice(int device);

}  // namespace klat
#endif
