// tool-common.h -- what the three tools share: the scale / filter options of SURVEY.md 2.3 plus
// the two knobs this implementation adds (--batch-size lattices per GPU batch, --device).
#ifndef KLAT_HOST_TOOL_COMMON_H_
#define KLAT_HOST_TOOL_COMMON_H_
#include <limits>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../kaldi-io.h"
#include "../lattice-functions.h"
#include "../table-writers.h"

namespace klat {

struct CommonOptions {
 public:
  float acoustic_scale = 1.0f, graph_scale = 1.0f, insertion_penalty = 0.0f;
  int32_t batch_size = 4096;      // lattices per GPU batch
  int32_t max_batch_arcs = 64 << 20;
  int32_t device = 0;
  int32_t lanes_per_lattice = 0;  // 0 = automatic
  void Register(ParseOptions* po) {
    po->Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods in the lattices.");
    po->Register("graph-scale", &graph_scale, "Scaling factor for graph probabilities in the lattices.");
    po->Register("insertion-penalty", &insertion_penalty,
                 "Add this penalty to the lattice arcs with non-epsilon output label (typically, equivalent to word insertion penalty).");
    po->Register("batch-size", &batch_size, "Number of lattices processed per GPU batch.");
    po->Register("max-batch-arcs", &max_batch_arcs, "Upper bound on arcs per GPU batch.");
    po->Register("device", &device, "GPU device index.");
    po->Register("lanes-per-lattice", &lanes_per_lattice, "Lanes per lattice of the sweep kernel (0 = choose from the batch).");
  }
  // The tools know their scales before they read a lattice: the stream packer applies them
  // (klat_pack_options.fuse_scales: 10-byte stream slots instead of 12).  Valid while this object lives.
  klat_pack_options PackOptions() const {
    fuse_ = {graph_scale, acoustic_scale, insertion_penalty};
    klat_pack_options o = {lanes_per_lattice, 0, 0, &fuse_};
    return o;
  }

 private:
  mutable klat_scales fuse_ = {1.0f, 1.0f, 0.0f};
};

struct IndexOptions {
  float beam = std::numeric_limits<float>::infinity();
  std::string exclude_str, include_str;
  int32_t num_threads = 1, num_threads_total = -1;
  bool device_prune = false;
  void Register(ParseOptions* po) {
    po->Register("beam", &beam, "Pruning beam (applied after acoustic scaling and adding the insertion penalty).");
    po->Register("device-prune", &device_prune,
                 "Take the --beam pruning decisions on the GPU (Viterbi sweep over the whole batch) instead of lattice by lattice on the host.");
    po->Register("exclude-words", &exclude_str, "Space-separated list of integers representing the words to exclude from the index.");
    po->Register("include-words", &include_str, "Space-separated list of integers representing the words to include in the index.");
    // TaskSequencerConfig options of the reference: accepted for command-line compatibility; the
    // GPU batch replaces the thread pool (one lattice per wave group instead of one per thread).
    po->Register("num-threads", &num_threads, "Accepted for compatibility; lattices are batched on the GPU instead.");
    po->Register("num-threads-total", &num_threads_total, "Accepted for compatibility.");
  }
  bool Prunes() const { return beam != std::numeric_limits<float>::infinity(); }
  void Check() const {
    if (Prunes() && !(beam > 0.0f)) KLAT_ERR("--beam must be positive");
  }
};

// Forward-backward over one batch, then the per-frame word posteriors of every lattice written
// as Posterior entries (latbin/lattice-to-word-frame-post.cc:93-139).  "Frames" are whatever the
// batch carries as state times: frames for lattice-to-word-frame-post, word positions for
// lattice-to-word-position-post.
inline void WritePosteriors(Context& ctx, const LatticeBatch& batch, const klat_pack_options& popt,
                            const CommonOptions& co, const std::vector<std::string>& keys,
                            const std::vector<int32_t>& num_frames, PosteriorWriter* writer) {
  DeviceBatch db(ctx, batch, &popt);
  db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
  int64_t T = 0;
  for (int32_t f : num_frames) T += f;
  std::vector<int64_t> frame_off((size_t)T + 1, 0);
  int64_t n = 0;
  Check(klat_frame_post(db.get(), num_frames.data(), frame_off.data(), nullptr, nullptr, 0, &n), "klat_frame_post");
  std::vector<int32_t> labels((size_t)std::max<int64_t>(n, 1));
  std::vector<float> logps((size_t)std::max<int64_t>(n, 1));
  Check(klat_frame_post(db.get(), num_frames.data(), frame_off.data(), labels.data(), logps.data(),
                        (int64_t)labels.size(), &n),
        "klat_frame_post");
  int64_t fb = 0;
  for (size_t u = 0; u < keys.size(); ++u) {
    writer->WriteFlat(keys[u], num_frames[u], frame_off.data() + fb, labels.data(), logps.data());
    fb += num_frames[u];
  }
}

// SequentialCompactLatticeReader with the reference's PruneLattice step applied
// (kwsbin2/lattice-word-index-segment.cc:49-51): lattices are read a batch ahead so that
// --device-prune can take the decisions of the whole batch in one Viterbi sweep on the GPU.
class PrunedLatticeReader {
 public:
  PrunedLatticeReader(const std::string& rspecifier, Context& ctx, const CommonOptions& co, const IndexOptions& io)
      : reader_(rspecifier), ctx_(ctx), co_(co), io_(io) {
    Fill();
  }
  bool Done() const { return pos_ >= keys_.size(); }
  const std::string& Key() const { return keys_[pos_]; }
  CompactLattice& Value() { return lats_[pos_]; }
  void FreeCurrent() { lats_[pos_].DeleteStates(); }
  void Next() {
    if (++pos_ >= keys_.size()) Fill();
  }

 private:
  void Fill() {
    keys_.clear();
    lats_.clear();
    pos_ = 0;
    int64_t arcs = 0;
    for (; !reader_.Done() && (int32_t)keys_.size() < co_.batch_size && arcs < co_.max_batch_arcs; reader_.Next()) {
      keys_.push_back(reader_.Key());
      arcs += reader_.Value().NumArcs();
      lats_.push_back(std::move(reader_.Value()));
      reader_.FreeCurrent();
    }
    if (!io_.Prunes() || keys_.empty()) return;
    if (io_.device_prune) {
      PruneLatticesOnDevice(ctx_, io_.beam, co_.graph_scale, co_.acoustic_scale, co_.insertion_penalty, &lats_);
    } else {
      for (auto& clat : lats_)
        if (clat.Start() != CompactLattice::kNoStateId)
          PruneLattice(io_.beam, co_.graph_scale, co_.acoustic_scale, co_.insertion_penalty, &clat);
    }
  }
  SequentialCompactLatticeReader reader_;
  Context& ctx_;
  const CommonOptions& co_;
  const IndexOptions& io_;
  std::vector<std::string> keys_;
  std::vector<CompactLattice> lats_;
  size_t pos_ = 0;
};

inline std::vector<int32_t> ParseIntList(const std::string& s, const char* what) {
  std::set<int32_t> out;
  std::istringstream is(s);
  std::string tok;
  while (is >> tok) {
    char* end = nullptr;
    const long v = std::strtol(tok.c_str(), &end, 10);
    if (end == tok.c_str() || *end) KLAT_ERR("Invalid integer \"" << tok << "\" in " << what);
    out.insert((int32_t)v);
  }
  return std::vector<int32_t>(out.begin(), out.end());
}

}  // namespace klat
#endif
