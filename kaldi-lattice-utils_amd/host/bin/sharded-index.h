// sharded-index.h -- the main loop the two kwsbin2 word indexers share, on one GPU or on several.
//
// The reference runs one LatticeScorerTask per lattice on a TaskSequencer thread pool and writes
// the records from the tasks' destructors, which run serially in submission order
// (kwsbin2/lattice-word-index-segment.cc:260-275, :96-128; -position.cc:275-290).  Here the
// workers are GPUs, one PROCESS each (--devices=0,1,..: the calling process is rank 0 and forks the
// other ranks before anything touches HIP): every rank reads the archive, a super-batch of
// world x batch lattices is dealt to the ranks as contiguous ranges of equal cumulative arc count
// (klat_partition_by_arcs), each rank runs its range through the sweep and the accumulator on its
// own GPU, the 24-byte records are gathered to rank 0 (klat_dist.h: RCCL send/recv from device
// buffers; --gather=host stages them through host memory and local sockets instead), and rank 0
// writes them in rank order = input order, one entry per key, empty ones included (:64-69,128).
#ifndef KLAT_HOST_SHARDED_INDEX_H_
#define KLAT_HOST_SHARDED_INDEX_H_
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>

#include "../../../include/klat_dist.h"
#include "../archive-index.h"
#include "tool-common.h"

namespace klat {

// "0,1,2", "0-7", "0-3,6"
inline std::vector<int32_t> ParseDeviceList(const std::string& s) {
  std::vector<int32_t> out;
  std::istringstream is(s);
  std::string tok;
  while (std::getline(is, tok, ',')) {
    if (tok.empty()) continue;
    const size_t dash = tok.find('-');
    char* end = nullptr;
    const long a = std::strtol(tok.c_str(), &end, 10);
    long b = a;
    if (dash != std::string::npos && dash > 0) {
      if (end != tok.c_str() + dash) KLAT_ERR("Invalid --devices entry \"" << tok << "\"");
      b = std::strtol(tok.c_str() + dash + 1, &end, 10);
    }
    if (end == tok.c_str() || *end || a < 0 || b < a) KLAT_ERR("Invalid --devices entry \"" << tok << "\"");
    for (long d = a; d <= b; ++d) out.push_back((int32_t)d);
  }
  return out;
}

namespace detail {
// SIGCHLD on the writing rank: a worker that ended badly must not leave the writer waiting in a
// collective.  Peeks (WNOWAIT) at every worker so that Finish() can still reap them.
inline pid_t g_workers[64];
inline int g_num_workers = 0;
inline void OnWorkerExit(int) {
  for (int i = 0; i < g_num_workers; ++i) {
    siginfo_t info;
    info.si_pid = 0;
    if (waitid(P_PID, (id_t)g_workers[i], &info, WEXITED | WNOHANG | WNOWAIT) == 0 && info.si_pid != 0 &&
        !(info.si_code == CLD_EXITED && info.si_status == 0)) {
      const char msg[] = "a worker rank failed: aborting\n";
      (void)!write(2, msg, sizeof(msg) - 1);
      _exit(255);  // the other workers die with their parent (PR_SET_PDEATHSIG)
    }
  }
}
}  // namespace detail

// The ranks of a --devices run.  Construct BEFORE the first HIP call of the process.
class DeviceRanks {
 public:
  DeviceRanks(const std::string& devices, int32_t default_device, const std::string& gather) {
    devices_ = ParseDeviceList(devices);
    if (devices_.empty()) devices_.push_back(default_device);
    world_ = (int32_t)devices_.size();
    host_gather_ = gather == "host";
    if (gather != "rccl" && gather != "host") KLAT_ERR("--gather must be rccl or host");
    if (world_ == 1) return;
    if (world_ > 64) KLAT_ERR("--devices: at most 64 GPUs");
    shared_ = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (shared_ == MAP_FAILED) KLAT_ERR("mmap: " << std::strerror(errno));
    new (shared_) Shared();
    socks_.assign((size_t)world_, -1);
    std::vector<int> child_end((size_t)world_, -1);
    if (host_gather_)
      for (int32_t r = 1; r < world_; ++r) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) KLAT_ERR("socketpair: " << std::strerror(errno));
        socks_[(size_t)r] = sv[0];
        child_end[(size_t)r] = sv[1];
      }
    const pid_t parent = getpid();
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_handler = &detail::OnWorkerExit;
    sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
    sigaction(SIGCHLD, &sa, nullptr);
    for (int32_t r = 1; r < world_; ++r) {
      const pid_t pid = fork();
      if (pid < 0) KLAT_ERR("fork: " << std::strerror(errno));
      if (pid == 0) {  // worker rank r
        signal(SIGCHLD, SIG_DFL);
        prctl(PR_SET_PDEATHSIG, SIGKILL);  // never outlive the writer
        if (getppid() != parent) _exit(255);
        rank_ = r;
        children_.clear();
        for (int32_t q = 1; q < world_; ++q) {
          if (socks_[(size_t)q] >= 0) close(socks_[(size_t)q]);
          if (q != r && child_end[(size_t)q] >= 0) close(child_end[(size_t)q]);
        }
        socks_.assign((size_t)world_, -1);
        socks_[0] = child_end[(size_t)r];
        return;
      }
      children_.push_back(pid);
      detail::g_workers[detail::g_num_workers++] = pid;
    }
    for (int fd : child_end)
      if (fd >= 0) close(fd);
  }
  ~DeviceRanks() {
    if (comm_) klat_comm_destroy(comm_);
  }
  int32_t rank() const { return rank_; }
  int32_t world() const { return world_; }
  int32_t device() const { return devices_[(size_t)rank_]; }
  bool host_gather() const { return host_gather_; }
  klat_comm* comm() { return comm_; }

  // after the rank's Context exists
  void InitComm(Context& ctx) {
    if (world_ == 1) return;
    if (host_gather_) {
      transport_.user = this;
      transport_.all_gather_i64 = &DeviceRanks::AllGather;
      transport_.group_begin = [](void*) { return 0; };
      transport_.send = &DeviceRanks::Send;
      transport_.recv = &DeviceRanks::Recv;
      transport_.group_end = [](void*) { return 0; };
      Check(klat_comm_create_custom(rank_, world_, &transport_, &comm_), "klat_comm_create_custom");
      return;
    }
    if (rank_ == 0) {
      Check(klat_rccl_unique_id(shared_->id), "klat_rccl_unique_id");
      shared_->ready.store(1, std::memory_order_release);
    } else {
      while (shared_->ready.load(std::memory_order_acquire) == 0) usleep(1000);
    }
    Check(klat_comm_create_rccl(ctx.get(), rank_, world_, shared_->id, &comm_), "klat_comm_create_rccl");
  }

  // Rank 0: waits for the workers; returns 0 if all of them succeeded.  Workers do not return.
  int Finish(int my_status) {
    if (world_ == 1) return my_status;
    // the orderly teardown of the communicator is for the success path: after a failure the other
    // ranks may sit inside the gather, and ncclCommDestroy could wait for them for ever
    if (comm_ && my_status == 0) klat_comm_destroy(comm_);
    comm_ = nullptr;
    if (rank_ != 0) {
      std::cout.flush();
      std::cerr.flush();
      _exit(my_status == 0 ? 0 : 255);
    }
    int bad = my_status;
    signal(SIGCHLD, SIG_DFL);
    for (pid_t pid : children_) {
      int st = 0;
      if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = bad ? bad : -1;
    }
    return bad;
  }
  // a rank that fails must not leave the others waiting in a collective: the writer takes its
  // workers down; a failing worker exits non-zero (Finish), which the writer's SIGCHLD handler sees
  void Abort() {
    if (world_ > 1 && rank_ == 0) {
      signal(SIGCHLD, SIG_DFL);
      for (pid_t pid : children_) kill(pid, SIGKILL);
    }
  }

 private:
  struct Shared {
    std::atomic<int> ready{0};
    unsigned char id[KLAT_UNIQUE_ID_BYTES];
  };
  static bool WriteAll(int fd, const void* p, int64_t n) {
    const char* c = static_cast<const char*>(p);
    while (n > 0) {
      const ssize_t k = write(fd, c, (size_t)n);
      if (k < 0 && errno == EINTR) continue;
      if (k <= 0) return false;
      c += k, n -= k;
    }
    return true;
  }
  static bool ReadAll(int fd, void* p, int64_t n) {
    char* c = static_cast<char*>(p);
    while (n > 0) {
      const ssize_t k = read(fd, c, (size_t)n);
      if (k < 0 && errno == EINTR) continue;
      if (k <= 0) return false;
      c += k, n -= k;
    }
    return true;
  }
  // host transport: a star around rank 0 (the only root the tools use)
  static int AllGather(void* user, const int64_t* send, int64_t* recv, int32_t count) {
    DeviceRanks* d = static_cast<DeviceRanks*>(user);
    const int64_t each = (int64_t)sizeof(int64_t) * count;
    if (d->rank_ == 0) {
      std::memcpy(recv, send, (size_t)each);
      for (int32_t r = 1; r < d->world_; ++r)
        if (!ReadAll(d->socks_[(size_t)r], recv + (int64_t)r * count, each)) return 1;
      for (int32_t r = 1; r < d->world_; ++r)
        if (!WriteAll(d->socks_[(size_t)r], recv, each * d->world_)) return 1;
      return 0;
    }
    return WriteAll(d->socks_[0], send, each) && ReadAll(d->socks_[0], recv, each * d->world_) ? 0 : 1;
  }
  static int Send(void* user, const void* buf, int64_t bytes, int32_t peer) {
    DeviceRanks* d = static_cast<DeviceRanks*>(user);
    return d->socks_[(size_t)peer] >= 0 && WriteAll(d->socks_[(size_t)peer], buf, bytes) ? 0 : 1;
  }
  static int Recv(void* user, void* buf, int64_t bytes, int32_t peer) {
    DeviceRanks* d = static_cast<DeviceRanks*>(user);
    return d->socks_[(size_t)peer] >= 0 && ReadAll(d->socks_[(size_t)peer], buf, bytes) ? 0 : 1;
  }

  std::vector<int32_t> devices_;
  int32_t rank_ = 0, world_ = 1;
  bool host_gather_ = false;
  Shared* shared_ = nullptr;
  std::vector<pid_t> children_;
  std::vector<int> socks_;  // host transport: rank 0 holds one socket per worker, a worker holds [0]
  klat_transport transport_ = {};
  klat_comm* comm_ = nullptr;
};

struct ShardOptions {
  std::string devices, gather = "rccl";
  void Register(ParseOptions* po) {
    po->Register("devices", &devices,
                 "GPUs to shard the lattices over, one process each (e.g. 0-7 or 0,2,5); empty = --device only.");
    po->Register("gather", &gather,
                 "How the index records reach the writing rank: rccl (device buffers over xGMI) or host (host memory over local sockets).");
  }
};

// ---- tools whose output needs no gather (lattice-to-word-frame-post, lattice-best-path2): with
// --devices every rank takes one contiguous range of the WHOLE archive (by bytes) and writes its own
// table, the wspecifier's "JOB" replaced by rank + 1 as Kaldi's parallel scripts name their pieces;
// the pieces concatenated in job order are the single-GPU output, entry for entry.  The loops are
// per-utterance independent in the reference, too (latbin/lattice-to-word-frame-post.cc:68,
// latbin/lattice-best-path2.cc:78). ----
inline std::string JobSpecifier(const std::string& wspecifier, int32_t rank, int32_t world) {
  const size_t at = wspecifier.find("JOB");
  if (at == std::string::npos) {
    if (world > 1)
      KLAT_ERR("--devices with more than one GPU writes one table per rank: put JOB into the wspecifier (e.g. ark:post.JOB.ark), "
               "it is replaced by 1.." << world);
    return wspecifier;
  }
  std::string out = wspecifier;
  out.replace(at, 3, std::to_string(rank + 1));
  return out;
}

// SequentialCompactLatticeReader look-alike over this rank's share of the archive.  A plain file is
// mapped and scanned (also on one GPU); the rank's entries are parsed kChunk at a time on `threads`
// threads, a chunk ahead of the caller's loop.
class RankArchiveReader {
 public:
  RankArchiveReader(const std::string& rspecifier, int32_t rank, int32_t world, int threads = 1) : threads_(threads) {
    if (!MappedArchive::Mappable(rspecifier)) {
      if (world > 1)
        KLAT_ERR("--devices with more than one GPU needs a plain archive file (every rank scans it for the entry boundaries), not "
                 << rspecifier);
      stream_.reset(new SequentialCompactLatticeReader(rspecifier));
      return;
    }
    archive_.reset(new MappedArchive(rspecifier));
    std::vector<int64_t> bytes;
    while (!archive_->Done()) {
      entries_.emplace_back();
      archive_->ScanNext(&entries_.back());
      bytes.push_back((int64_t)(entries_.back().end - entries_.back().begin));
    }
    std::vector<int64_t> bounds((size_t)world + 1);
    Check(klat_partition_by_arcs(bytes.data(), (int64_t)bytes.size(), world, bounds.data()), "klat_partition_by_arcs");
    pos_ = (size_t)bounds[(size_t)rank];
    end_ = (size_t)bounds[(size_t)rank + 1];
    total_ = entries_.size();
    chunk_begin_ = chunk_end_ = pos_;
  }
  bool Done() const { return stream_ ? stream_->Done() : pos_ >= end_; }
  const std::string& Key() const { return stream_ ? stream_->Key() : entries_[pos_].key; }
  CompactLattice& Value() {
    if (stream_) return stream_->Value();
    Load();  // (parses the chunk around pos_ when the caller first asks for a value of it)
    return chunk_[pos_ - chunk_begin_];
  }
  void FreeCurrent() {
    if (stream_) stream_->FreeCurrent();
    else if (pos_ >= chunk_begin_ && pos_ < chunk_end_) chunk_[pos_ - chunk_begin_].DeleteStates();
  }
  void Next() {
    if (stream_) {
      stream_->Next();
      return;
    }
    ++pos_;
  }
  size_t parsed() const { return parsed_; }
  size_t total() const { return total_; }
  // Mapped archives only (false otherwise: the caller walks Key() / Value()): the next up to `max` entries of
  // this rank as flat lattices, top-sorted from state 0 -- a binary entry straight from its bytes, any other
  // through the object form -- parsed on the reader's threads.  Call instead of, not mixed with, Next().
  bool mapped() const { return !stream_; }
  bool ReadFlatChunk(size_t max, int64_t max_arcs, std::vector<std::string>* keys, std::vector<FlatLattice>* flats) {
    keys->clear();
    flats->clear();
    if (stream_ || pos_ >= end_) return false;
    size_t n = 0;
    const size_t first = pos_;
    // (an entry's arcs are not known before it is parsed: 24 bytes is less than any arc takes in either format)
    for (int64_t bytes = 0; n < max && first + n < end_ && (n == 0 || bytes < max_arcs * 24); ++n)
      bytes += (int64_t)(entries_[first + n].end - entries_[first + n].begin);
    flats->resize(n);
    for (size_t i = 0; i < n; ++i) keys->push_back(entries_[first + i].key);
    ParallelFor((int64_t)n, threads_, [&](int64_t i) {
      FlatLattice& fl = (*flats)[(size_t)i];
      if (!archive_->ReadFlat(entries_[first + (size_t)i], &fl) || !IsTopSortedFromZero(fl)) {
        CompactLattice clat;
        archive_->Read(entries_[first + (size_t)i], &clat);
        if (clat.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&clat);
        ToFlat(clat, &fl);
      }
    });
    pos_ += n;
    parsed_ += n;
    chunk_.clear();
    chunk_begin_ = chunk_end_ = pos_;
    return true;
  }

 private:
  static constexpr size_t kChunk = 512;
  void Load() {
    if (pos_ < chunk_end_ || pos_ >= end_) return;
    chunk_begin_ = pos_;
    chunk_end_ = std::min(end_, pos_ + kChunk);
    chunk_.clear();
    chunk_.resize(chunk_end_ - chunk_begin_);
    ParallelFor((int64_t)chunk_.size(), threads_, [&](int64_t i) { archive_->Read(entries_[chunk_begin_ + (size_t)i], &chunk_[(size_t)i]); });
    parsed_ += chunk_.size();
  }
  std::unique_ptr<SequentialCompactLatticeReader> stream_;
  std::unique_ptr<MappedArchive> archive_;
  std::vector<ArchiveEntry> entries_;
  std::vector<CompactLattice> chunk_;
  size_t pos_ = 0, end_ = 0, parsed_ = 0, total_ = 0, chunk_begin_ = 0, chunk_end_ = 0;
  int threads_ = 1;
};

// A queue of at most `depth` items between two stages of the tools' pipeline (read | GPU | write).
template <class T>
class StageQueue {
 public:
  explicit StageQueue(size_t depth) : depth_(depth) {}
  void Push(T&& v) {
    std::unique_lock<std::mutex> lock(mu_);
    not_full_.wait(lock, [&] { return q_.size() < depth_; });
    q_.push_back(std::move(v));
    not_empty_.notify_one();
  }
  T Pop() {
    std::unique_lock<std::mutex> lock(mu_);
    not_empty_.wait(lock, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return v;
  }

 private:
  size_t depth_;
  std::mutex mu_;
  std::condition_variable not_full_, not_empty_;
  std::deque<T> q_;
};

// The lattices of a super-batch that this rank owns, ready for klat_batch_create: top-sort, [length
// disambiguation], state times per lattice on `threads` threads, then appended in input order.
template <bool POSITION>
inline void FlattenForIndex(std::vector<CompactLattice>* mine, LatticeBatch* batch, std::vector<int32_t>* state_len_all, int threads) {
  struct Prepared {
    CompactLattice dlat;
    std::vector<int32_t> times, state_len;
  };
  std::vector<Prepared> prep(mine->size());
  const auto tp0 = std::chrono::steady_clock::now();
  ParallelFor((int64_t)mine->size(), threads, [&](int64_t i) {
    CompactLattice& clat = (*mine)[(size_t)i];
    Prepared& p = prep[(size_t)i];
    if (POSITION) {
      if (clat.Start() != CompactLattice::kNoStateId) {
        TopSortCompactLatticeIfNeeded(&clat);
        DisambiguateStateInputSequenceLength(clat, &p.dlat, &p.state_len, false);  // kwsbin2/lattice-word-index-position.cc:64-65
        CompactLatticeStateTimes(p.dlat, &p.times);
      }
      clat.DeleteStates();
    } else if (clat.Start() != CompactLattice::kNoStateId) {
      TopSortCompactLatticeIfNeeded(&clat);
      CompactLatticeStateTimes(clat, &p.times);
    }
  });
  const auto tp1 = std::chrono::steady_clock::now();
  std::vector<const CompactLattice*> lats(mine->size());
  std::vector<const std::vector<int32_t>*> times(mine->size());
  for (size_t i = 0; i < mine->size(); ++i) {
    lats[i] = POSITION ? &prep[i].dlat : &(*mine)[i];
    times[i] = &prep[i].times;
    if (POSITION) state_len_all->insert(state_len_all->end(), prep[i].state_len.begin(), prep[i].state_len.end());
  }
  batch->AddMany(lats, times, threads);
  const auto tp2 = std::chrono::steady_clock::now();
  // (a lattice is a vector of arcs per state: freeing two thousand lattices one after the other took as long as parsing them)
  ParallelFor((int64_t)mine->size(), threads, [&](int64_t i) {
    (*mine)[(size_t)i].DeleteStates();
    prep[(size_t)i].dlat.DeleteStates();
  });
  if (std::getenv("KLAT_TOOL_PROFILE")) {
    const auto tp3 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "flatten: sort+times %.3f  add %.3f  free %.3f s\n", std::chrono::duration<double>(tp1 - tp0).count(),
                 std::chrono::duration<double>(tp2 - tp1).count(), std::chrono::duration<double>(tp3 - tp2).count());
  }
}

struct DevicesOption {
  std::string devices;
  void Register(ParseOptions* po) {
    po->Register("devices", &devices,
                 "GPUs to shard the lattices over, one process each (e.g. 0-7 or 0,2,5), every rank writing its own table "
                 "(JOB in the wspecifier = rank + 1); empty = --device only.");
  }
};

// POSITION = false: lattice-word-index-segment; true: lattice-word-index-position.
//
// Three stages run side by side in every rank: a reader thread (scan the next super-batch of the
// archive, parse THIS RANK'S entries only, prune / top-sort / state times / flatten), the calling
// thread (the GPU: stream packing, sweep, index, gather to rank 0) and, on rank 0, a writer thread
// (the records of the super-batch before).  A plain file is mapped and scanned (archive-index.h):
// the ranks agree on the entries' byte ranges without parsing them, deal them out by bytes, and
// each parses its own share -- the reference reads every lattice once
// (kwsbin2/lattice-word-index-segment.cc:261-274), and so does the node.  Standard input or a
// command cannot be scanned ahead: every rank then parses everything, as before.
template <bool POSITION, class Writer, class Record>
int RunIndexTool(const CommonOptions& co, const IndexOptions& io, const ShardOptions& so, const std::string& rspecifier,
                 const std::string& wspecifier) {
  static_assert(sizeof(Record) == 24, "index records are 24 bytes");
  const std::vector<int32_t> include = ParseIntList(io.include_str, "--include-words");
  const std::vector<int32_t> exclude = ParseIntList(io.exclude_str, "--exclude-words");
  const klat_label_filter filter = {include.data(), (int32_t)include.size(), exclude.data(), (int32_t)exclude.size()};
  DeviceRanks ranks(so.devices, co.device, so.gather);  // forks: nothing before this line touches HIP
  const int32_t world = ranks.world(), rank = ranks.rank();
  if (world > 1 && ParseSpecifier(rspecifier, false).filename == "-")
    KLAT_ERR("--devices with more than one GPU needs an input every rank can read (a file or a command), not standard input");
  int status = 0;
  struct SuperBatch {
    std::vector<std::string> keys;      // all entries of the super-batch, in input order
    int64_t lo = 0, hi = 0;             // this rank's range
    std::vector<CompactLattice> mine;   // --device-prune: still to be pruned and flattened by the GPU stage
    LatticeBatch batch;
    std::vector<int32_t> state_len_all;
    bool end = false;
    std::exception_ptr error;
  };
  struct WriteJob {
    std::vector<std::string> keys;
    Record* rec = nullptr;  // a buffer of `pinned`
    size_t rec_cap = 0;
    std::vector<int64_t> off;
    bool end = false;
  };
  PinnedPool pinned;
  StageQueue<std::unique_ptr<SuperBatch>> read_q(2);
  StageQueue<std::unique_ptr<WriteJob>> write_q(2);
  std::thread reader_thread, writer_thread;
  std::atomic<bool> stop{false};
  bool reader_ended = false;  // the reader's end marker has been taken off read_q (nothing more will come)
  std::exception_ptr writer_error;
  try {
    std::unique_ptr<Writer> writer;
    if (rank == 0) writer.reset(new Writer(wspecifier));  // only the writing rank opens (and truncates) the output
    Context ctx(ranks.device());
    ranks.InitComm(ctx);
    klat_pack_options popt = co.PackOptions();
    const bool mapped = MappedArchive::Mappable(rspecifier);
    const int host_threads = co.HostThreads(world);
    int64_t parsed_lattices = 0, parsed_arcs = 0, seen_lattices = 0;
    double t_scan = 0, t_parse = 0, t_prep = 0, t_gpu = 0, t_write = 0;  // seconds per stage (--verbose=1)
    double t_create = 0, t_sweep = 0, t_fetch = 0;                       // inside the GPU stage
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };

    reader_thread = std::thread([&]() {
      try {
        std::unique_ptr<MappedArchive> archive;
        std::unique_ptr<SequentialCompactLatticeReader> reader;
        if (mapped) archive.reset(new MappedArchive(rspecifier));
        else reader.reset(new SequentialCompactLatticeReader(rspecifier));
        auto done = [&]() { return mapped ? archive->Done() : reader->Done(); };
        while (!done() && !stop.load()) {
          // one super-batch: what `world` GPUs take in one go
          std::unique_ptr<SuperBatch> sb(new SuperBatch());
          std::vector<int64_t> weight;  // what the entries are dealt out by: bytes (scanned) or arcs (parsed)
          std::vector<ArchiveEntry> entries;
          std::vector<CompactLattice> lats;
          int64_t total = 0;
          // (a scanned entry's arcs are not known: 24 bytes is less than any arc takes in either format)
          const int64_t budget = (int64_t)co.max_batch_arcs * world * (mapped ? 24 : 1);
          auto t0 = now();
          while (!done() && (int64_t)sb->keys.size() < (int64_t)co.batch_size * world && total < budget) {
            if (mapped) {
              entries.emplace_back();
              archive->ScanNext(&entries.back());
              sb->keys.push_back(entries.back().key);
              weight.push_back((int64_t)(entries.back().end - entries.back().begin));
            } else {
              sb->keys.push_back(reader->Key());
              weight.push_back(reader->Value().NumArcs());
              lats.push_back(std::move(reader->Value()));
              reader->FreeCurrent();
              reader->Next();
            }
            total += weight.back();
          }
          t_scan += since(t0);
          t0 = now();
          const int64_t n = (int64_t)sb->keys.size();
          std::vector<int64_t> bounds((size_t)world + 1);
          Check(klat_partition_by_arcs(weight.data(), n, world, bounds.data()), "klat_partition_by_arcs");
          sb->lo = bounds[(size_t)rank], sb->hi = bounds[(size_t)rank + 1];
          std::vector<CompactLattice> mine;
          // The segment index without --beam needs nothing of a lattice but its arrays: a binary entry is read
          // straight into them (FlatLattice), no object per state; any other entry, and a lattice that is not
          // top-sorted yet, goes through the object form and is flattened afterwards.
          const bool flat_path = mapped && !POSITION && !io.Prunes();
          if (flat_path) {
            const int64_t m = sb->hi - sb->lo;
            std::vector<FlatLattice> flats((size_t)m);
            std::vector<std::vector<int32_t>> ftimes((size_t)m);
            ParallelFor(m, host_threads, [&](int64_t i) {
              FlatLattice& fl = flats[(size_t)i];
              if (!archive->ReadFlat(entries[(size_t)(sb->lo + i)], &fl) || !IsTopSortedFromZero(fl)) {
                CompactLattice clat;
                archive->Read(entries[(size_t)(sb->lo + i)], &clat);
                if (clat.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&clat);
                ToFlat(clat, &fl);
              }
              if (fl.NumStates() > 0) FlatLatticeStateTimes(fl, &ftimes[(size_t)i]);
            });
            t_parse += since(t0);
            t0 = now();
            for (const auto& fl : flats) parsed_arcs += fl.NumArcs();
            seen_lattices += n;
            parsed_lattices += m;
            sb->batch.AddManyFlat(flats, ftimes, host_threads);
            t_prep += since(t0);
            read_q.Push(std::move(sb));
            continue;
          }
          if (mapped) {
            mine.resize((size_t)(sb->hi - sb->lo));
            ParallelFor(sb->hi - sb->lo, host_threads, [&](int64_t i) { archive->Read(entries[(size_t)(sb->lo + i)], &mine[(size_t)i]); });
          } else {
            mine.assign(std::make_move_iterator(lats.begin() + sb->lo), std::make_move_iterator(lats.begin() + sb->hi));
            lats.clear();  // other ranks' lattices: only their arc counts mattered
          }
          t_parse += since(t0);
          t0 = now();
          seen_lattices += n;
          parsed_lattices += mapped ? sb->hi - sb->lo : n;
          if (mapped)
            for (const auto& clat : mine) parsed_arcs += clat.NumArcs();
          else
            parsed_arcs += total;
          if (io.Prunes() && io.device_prune) {
            sb->mine = std::move(mine);  // the GPU stage takes the pruning decisions of the whole batch
          } else {
            if (io.Prunes())  // PruneLattice(beam, &clat), kwsbin2/lattice-word-index-segment.cc:49-51
              ParallelFor((int64_t)mine.size(), host_threads, [&](int64_t i) {
                if (mine[(size_t)i].Start() != CompactLattice::kNoStateId)
                  PruneLattice(io.beam, co.graph_scale, co.acoustic_scale, co.insertion_penalty, &mine[(size_t)i]);
              });
            FlattenForIndex<POSITION>(&mine, &sb->batch, &sb->state_len_all, host_threads);
          }
          t_prep += since(t0);
          read_q.Push(std::move(sb));
        }
        std::unique_ptr<SuperBatch> last(new SuperBatch());
        last->end = true;
        read_q.Push(std::move(last));
      } catch (...) {
        std::unique_ptr<SuperBatch> last(new SuperBatch());
        last->end = true;
        last->error = std::current_exception();
        read_q.Push(std::move(last));
      }
    });
    if (rank == 0)
      writer_thread = std::thread([&]() {
        try {
          for (;;) {
            std::unique_ptr<WriteJob> job = write_q.Pop();
            if (job->end) break;
            if (writer_error) continue;
            const auto t0 = now();
            std::vector<std::string> values(job->keys.size());  // formatted on the host threads, written in input order
            ParallelFor((int64_t)values.size(), host_threads, [&](int64_t u) {
              Writer::Format(writer->binary(), job->rec + job->off[(size_t)u], job->off[(size_t)u + 1] - job->off[(size_t)u], &values[(size_t)u]);
            });
            pinned.Release(job->rec, job->rec_cap);
            job->rec = nullptr;
            for (size_t u = 0; u < job->keys.size(); ++u) writer->WriteFormatted(job->keys[u], values[u]);
            t_write += since(t0);
          }
        } catch (...) {
          writer_error = std::current_exception();
          for (;;)  // keep draining: the GPU stage must never block on a dead writer
            if (write_q.Pop()->end) break;
        }
      });

    int64_t num_lattices = 0;
    for (;;) {
      std::unique_ptr<SuperBatch> sb = read_q.Pop();
      if (sb->end) reader_ended = true;
      if (sb->error) std::rethrow_exception(sb->error);
      if (sb->end) break;
      if (writer_error) std::rethrow_exception(writer_error);
      const auto t_gpu0 = now();
      const int64_t n = (int64_t)sb->keys.size(), lo = sb->lo, hi = sb->hi;
      num_lattices += n;
      if (!sb->mine.empty()) {
        PruneLatticesOnDevice(ctx, io.beam, co.graph_scale, co.acoustic_scale, co.insertion_penalty, &sb->mine);
        FlattenForIndex<POSITION>(&sb->mine, &sb->batch, &sb->state_len_all, host_threads);
      }
      // this rank's range on its GPU
      std::unique_ptr<DeviceBatch> db;
      std::vector<int64_t> off((size_t)(hi - lo) + 1, 0);
      int64_t nrec = 0;
      if (hi > lo) {
        auto t0 = now();
        db.reset(new DeviceBatch(ctx, sb->batch, &popt));
        t_create += since(t0);
        t0 = now();
        db->ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
        if (POSITION)
          Check(klat_position_index(db->get(), sb->state_len_all.data(), &filter, off.data(), nullptr, 0, &nrec), "klat_position_index");
        else
          Check(klat_segment_index(db->get(), &filter, off.data(), nullptr, 0, &nrec), "klat_segment_index");
        t_sweep += since(t0);
      }
      const auto t_fetch0 = now();
      std::unique_ptr<WriteJob> job(new WriteJob());
      auto rec_buffer = [&](int64_t records) {
        job->rec = static_cast<Record*>(pinned.Acquire(sizeof(Record) * (size_t)std::max<int64_t>(records, 1), &job->rec_cap));
      };
      auto fetch_mine = [&]() {
        rec_buffer(nrec);
        if (hi <= lo) return;
        const int64_t cap = std::max<int64_t>(nrec, 1);
        if (POSITION)
          Check(klat_position_index(db->get(), sb->state_len_all.data(), &filter, off.data(),
                                    reinterpret_cast<klat_position_record*>(job->rec), cap, &nrec),
                "klat_position_index");
        else
          Check(klat_segment_index(db->get(), &filter, off.data(), reinterpret_cast<klat_segment_record*>(job->rec), cap, &nrec),
                "klat_segment_index");
      };
      if (world == 1) {
        fetch_mine();
        job->off = off;
      } else {
        int64_t tu = 0, tr = 0;
        if (ranks.host_gather()) {
          fetch_mine();
          Check(klat_gather_records(ranks.comm(), job->rec, 24, off.data(), hi - lo, 0, &tu, &tr), "klat_gather_records");
        } else if (hi > lo) {
          Check(klat_gather_index(ranks.comm(), db->get(), 0, &tu, &tr), "klat_gather_index");
        } else {
          Check(klat_gather_records(ranks.comm(), nullptr, 24, off.data(), 0, 0, &tu, &tr), "klat_gather_records");
        }
        if (rank == 0) {
          if (tu != n) KLAT_ERR("gathered " << tu << " utterances, expected " << n);
          job->off.assign((size_t)n + 1, 0);
          pinned.Release(job->rec, job->rec_cap);  // (this rank's own records, if it fetched them for the host transport)
          rec_buffer(tr);
          Check(klat_gather_fetch(ranks.comm(), job->off.data(), job->rec, std::max<int64_t>(tr, 1)), "klat_gather_fetch");
        }
      }
      t_fetch += since(t_fetch0);
      KLAT_VLOG(1, "Rank " << rank << "/" << world << ": lattices [" << lo << ", " << hi << ") of a super-batch of " << n
                           << ", " << sb->batch.NumArcs() << " arcs, " << nrec << " records");
      t_gpu += since(t_gpu0);
      if (rank == 0) {
        job->keys = std::move(sb->keys);
        write_q.Push(std::move(job));
      } else {
        pinned.Release(job->rec, job->rec_cap);
      }
    }
    reader_thread.join();
    if (rank == 0) {
      std::unique_ptr<WriteJob> last(new WriteJob());
      last->end = true;
      write_q.Push(std::move(last));
      writer_thread.join();
      if (writer_error) std::rethrow_exception(writer_error);
    }
    if (writer) writer->Close();
    KLAT_VLOG(1, "Rank " << rank << "/" << world << ": parsed " << parsed_lattices << " of " << seen_lattices << " lattices ("
                         << parsed_arcs << " arcs)" << (mapped ? ", archive scanned in place" : ", archive read as a stream"));
    KLAT_VLOG(1, "Rank " << rank << "/" << world << " stages (s, they overlap): scan " << t_scan << "  parse " << t_parse << "  prepare " << t_prep
                         << "  GPU + gather " << t_gpu << " (batch create " << t_create << ", sweep + index " << t_sweep << ", fetch / gather "
                         << t_fetch << ")  write " << t_write << "  (" << host_threads << " host threads)");
    KLAT_VLOG(1, "Done " << num_lattices << " lattices.");
  } catch (const std::exception& e) {
    if (world > 1) std::cerr << "rank " << rank << ": ";
    std::cerr << e.what();
    status = -1;
    ranks.Abort();
    // the stages must not outlive this frame: unblock and join them
    stop.store(true);
    if (reader_thread.joinable()) {
      // (a reader that failed -- a cyclic lattice, a truncated entry -- has handed over its end marker with the error and
      // is gone: there is nothing left to drain, and a drain would wait for a marker that never comes)
      std::thread drain([&]() {
        while (!reader_ended)
          if (read_q.Pop()->end) break;
      });
      reader_thread.join();
      drain.join();
    }
    if (writer_thread.joinable()) {
      std::unique_ptr<WriteJob> last(new WriteJob());
      last->end = true;
      write_q.Push(std::move(last));
      writer_thread.join();
    }
  }
  return ranks.Finish(status);
}

}  // namespace klat
#endif
