// compact-lattice.h -- host-side CompactLattice (a1 of SURVEY.md 8a) and its archive reader.
//
// Mirrors what the reference tools use of `kaldi::CompactLattice` =
// VectorFst<ArcTpl<CompactLatticeWeightTpl<LatticeWeightTpl<float>,int32>>> (used at
// latbin/lattice-to-word-frame-post.cc:71, kwsbin2/lattice-word-index-segment.cc:181): per-arc
// (ilabel==olabel word id, graph cost, acoustic cost, transition-id string, nextstate) and a
// final weight per state.  Only the LENGTH of each transition-id string is kept: nothing on the
// hot path reads the ids themselves (CompactLatticeStateTimes uses String().size()).  Tools that
// write lattices back (lattice-prune-dyn-beam) ask the reader to keep the ids in a per-lattice pool.
#ifndef KLAT_HOST_COMPACT_LATTICE_H_
#define KLAT_HOST_COMPACT_LATTICE_H_
#include <cstdint>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "kaldi-io.h"

namespace klat {

struct CompactLatticeArc {
  int32_t ilabel = 0, olabel = 0;
  float graph = 0.f, acoustic = 0.f;  // LatticeWeight Value1 / Value2
  int32_t string_len = 0;             // weight.String().size()
  int32_t nextstate = 0;
  int32_t string_off = -1;            // first id in the lattice's StringPool(), -1 = ids not kept
};

struct CompactLatticeState {
  float final_graph = std::numeric_limits<float>::infinity();     // Zero = (+inf,+inf)
  float final_acoustic = std::numeric_limits<float>::infinity();
  int32_t final_string_len = 0;
  int32_t final_string_off = -1;
  std::vector<CompactLatticeArc> arcs;
  bool IsFinal() const {
    return !(final_graph == std::numeric_limits<float>::infinity() &&
             final_acoustic == std::numeric_limits<float>::infinity());
  }
};

class CompactLattice {
 public:
  static constexpr int32_t kNoStateId = -1;
  int32_t Start() const { return start_; }
  void SetStart(int32_t s) { start_ = s; }
  int32_t NumStates() const { return (int32_t)states_.size(); }
  int64_t NumArcs() const;
  int32_t AddState() {
    states_.emplace_back();
    return NumStates() - 1;
  }
  CompactLatticeState& State(int32_t s) { return states_[s]; }
  const CompactLatticeState& State(int32_t s) const { return states_[s]; }
  void DeleteStates() {
    states_.clear();
    string_pool_.clear();
    start_ = kNoStateId;
  }
  // transition ids of all weights, addressed by string_off / final_string_off (empty unless the
  // reader was told to keep them)
  std::vector<int32_t>& StringPool() { return string_pool_; }
  const std::vector<int32_t>& StringPool() const { return string_pool_; }
  std::vector<CompactLatticeState>& States() { return states_; }
  const std::vector<CompactLatticeState>& States() const { return states_; }

 private:
  int32_t start_ = kNoStateId;
  std::vector<CompactLatticeState> states_;
  std::vector<int32_t> string_pool_;
};

// Text (Appendix B.2; Lattice-format B.3 entries go through ConvertLattice: linear chains merge) and binary
// (Appendix B.4, OpenFst VectorFst stream) readers for ONE lattice positioned after "key ".
bool ReadCompactLatticeText(std::istream& is, CompactLattice* clat, bool keep_strings = false);
bool ReadCompactLatticeBinary(std::istream& is, CompactLattice* clat, bool keep_strings = false);
// Binary writer (same stream layout), used by the round-trip test of the binary reader.
void WriteCompactLatticeBinary(std::ostream& os, const CompactLattice& clat);
void WriteCompactLatticeText(std::ostream& os, const CompactLattice& clat);

// SequentialCompactLatticeReader look-alike: for (; !r.Done(); r.Next()) { r.Key(); r.Value(); }
class SequentialCompactLatticeReader {
 public:
  explicit SequentialCompactLatticeReader(const std::string& rspecifier, bool keep_strings = false);
  bool Done() const { return done_; }
  const std::string& Key() const { return key_; }
  CompactLattice& Value() { return value_; }
  void FreeCurrent() { value_.DeleteStates(); }
  void Next();

 private:
  std::unique_ptr<Input> input_;
  std::string key_;
  CompactLattice value_;
  bool done_ = false;
  bool keep_strings_ = false;
};

}  // namespace klat
#endif
