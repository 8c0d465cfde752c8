// tool-common.h -- what the three tools share: the scale / filter options of SURVEY.md 2.3 plus
// the two knobs this implementation adds (--batch-size lattices per GPU batch, --device).
#ifndef KLAT_HOST_TOOL_COMMON_H_
#define KLAT_HOST_TOOL_COMMON_H_
#include <atomic>
#include <exception>
#include <limits>
#include <mutex>
#include <set>
#include <thread>
#include <sstream>
#include <string>
#include <vector>

#include "../kaldi-io.h"
#include "../lattice-functions.h"
#include "../table-writers.h"

namespace klat {

// Host threads a rank spends on parsing and preparing its lattices (the archive entries are
// independent byte ranges; the GPU side of a C2-sized batch takes milliseconds, its parsing on one
// thread a third of a second): the machine's cores over the ranks, sixteen at most.
inline int DefaultHostThreads(int world) {
  const unsigned hc = std::thread::hardware_concurrency();
  const int t = hc ? (int)hc / std::max(1, world) : 1;
  return std::max(1, std::min(t, 16));
}

// f(i) for i in [0, n) on `threads` threads (dynamic schedule); the first exception is rethrown here.
template <class F>
inline void ParallelFor(int64_t n, int threads, F f) {
  if (threads <= 1 || n < 2) {
    for (int64_t i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int64_t> next(0);
  std::exception_ptr error;
  std::mutex mu;
  auto worker = [&]() {
    try {
      for (;;) {
        const int64_t i = next.fetch_add(1);
        if (i >= n) break;
        f(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      if (!error) error = std::current_exception();
      next.store(n);
    }
  };
  std::vector<std::thread> pool;
  const int nt = (int)std::min<int64_t>(threads, n);
  for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (error) std::rethrow_exception(error);
}

struct CommonOptions {
 public:
  float acoustic_scale = 1.0f, graph_scale = 1.0f, insertion_penalty = 0.0f;
  int32_t batch_size = 4096;      // lattices per GPU batch
  int32_t max_batch_arcs = 64 << 20;
  int32_t device = 0;
  int32_t lanes_per_lattice = 0;  // 0 = automatic
  int32_t host_threads = 0;       // 0 = DefaultHostThreads()
  void Register(ParseOptions* po) {
    po->Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods in the lattices.");
    po->Register("graph-scale", &graph_scale, "Scaling factor for graph probabilities in the lattices.");
    po->Register("insertion-penalty", &insertion_penalty,
                 "Add this penalty to the lattice arcs with non-epsilon output label (typically, equivalent to word insertion penalty).");
    po->Register("batch-size", &batch_size, "Number of lattices processed per GPU batch.");
    po->Register("max-batch-arcs", &max_batch_arcs, "Upper bound on arcs per GPU batch.");
    po->Register("device", &device, "GPU device index.");
    po->Register("lanes-per-lattice", &lanes_per_lattice, "Lanes per lattice of the sweep kernel (0 = choose from the batch).");
    po->Register("host-threads", &host_threads, "Threads that parse and prepare the lattices of a batch (0 = the machine's cores over the ranks, 16 at most).");
  }
  int HostThreads(int world) const { return host_threads > 0 ? host_threads : DefaultHostThreads(world); }
  // The tools know their scales before they read a lattice: the stream packer applies them
  // (klat_pack_options.fuse_scales: 10-byte stream slots instead of 12).  Valid while this object lives.
  klat_pack_options PackOptions() const {
    fuse_ = {graph_scale, acoustic_scale, insertion_penalty};
    klat_pack_options o = {lanes_per_lattice, 0, 0, &fuse_, 0};
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

// Page-locked buffers (klat_host_alloc) for results on their way from the GPU to the writer, recycled
// between batches: into fresh pageable memory 264 MB of records came back at 4 GB/s.
class PinnedPool {
 public:
  ~PinnedPool() {
    for (auto& b : free_) klat_host_free(b.first);
  }
  void* Acquire(size_t bytes, size_t* cap) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)  // the smallest buffer that is large enough
        if (free_[i].second >= bytes && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
      if (best < free_.size()) {
        void* p = free_[best].first;
        *cap = free_[best].second;
        free_.erase(free_.begin() + (long)best);
        return p;
      }
    }
    void* p = nullptr;
    *cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
    Check(klat_host_alloc(&p, *cap), "klat_host_alloc");
    return p;
  }
  void Release(void* p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(mu_);
    free_.emplace_back(p, cap);
    if (free_.size() > 8) {  // keep the larger ones
      size_t smallest = 0;
      for (size_t i = 1; i < free_.size(); ++i)
        if (free_[i].second < free_[smallest].second) smallest = i;
      klat_host_free(free_[smallest].first);
      free_.erase(free_.begin() + (long)smallest);
    }
  }

 private:
  std::mutex mu_;
  std::vector<std::pair<void*, size_t>> free_;
};

// Forward-backward over one batch, then the per-frame word posteriors of every lattice written
// as Posterior entries (latbin/lattice-to-word-frame-post.cc:93-139).  "Frames" are whatever the
// batch carries as state times: frames for lattice-to-word-frame-post, word positions for
// lattice-to-word-position-post.
inline void WritePosteriors(Context& ctx, const LatticeBatch& batch, const klat_pack_options& popt,
                            const CommonOptions& co, const std::vector<std::string>& keys,
                            const std::vector<int32_t>& num_frames, PosteriorWriter* writer, PinnedPool* pool = nullptr) {
  DeviceBatch db(ctx, batch, &popt);
  db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
  int64_t T = 0;
  for (int32_t f : num_frames) T += f;
  // results in page-locked buffers that the next batch reuses (or plain unset memory): a gigabyte of
  // posteriors into zeroed std::vectors cost more than the kernels that made them
  PinnedPool local;
  if (!pool) pool = &local;
  size_t cap_off = 0, cap_lab = 0, cap_lp = 0;
  int64_t* frame_off = static_cast<int64_t*>(pool->Acquire(sizeof(int64_t) * (size_t)(T + 1), &cap_off));
  int64_t n = 0;
  Check(klat_frame_post(db.get(), num_frames.data(), frame_off, nullptr, nullptr, 0, &n), "klat_frame_post");
  const int64_t cap = std::max<int64_t>(n, 1);
  int32_t* labels = static_cast<int32_t*>(pool->Acquire(sizeof(int32_t) * (size_t)cap, &cap_lab));
  float* logps = static_cast<float*>(pool->Acquire(sizeof(float) * (size_t)cap, &cap_lp));
  Check(klat_frame_post(db.get(), num_frames.data(), frame_off, labels, logps, cap, &n), "klat_frame_post");
  // formatted on the host threads (a text Posterior table is mostly number formatting), written in input order
  std::vector<int64_t> fb(keys.size() + 1, 0);
  for (size_t u = 0; u < keys.size(); ++u) fb[u + 1] = fb[u] + num_frames[u];
  std::vector<std::string> values(keys.size());
  ParallelFor((int64_t)keys.size(), co.HostThreads(1), [&](int64_t u) {
    PosteriorWriter::FormatFlat(writer->binary(), num_frames[(size_t)u], frame_off + fb[(size_t)u], labels, logps, &values[(size_t)u]);
  });
  pool->Release(frame_off, cap_off);
  pool->Release(labels, cap_lab);
  pool->Release(logps, cap_lp);
  for (size_t u = 0; u < keys.size(); ++u) writer->WriteFormatted(keys[u], values[u]);
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
