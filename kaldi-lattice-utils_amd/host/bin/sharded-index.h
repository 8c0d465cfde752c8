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
#include <cstring>

#include "../../../include/klat_dist.h"
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
    if (world_ > 64) KLAT_ERR("--devices: at most 64 GPUs");
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
    if (comm_) klat_comm_destroy(comm_), comm_ = nullptr;
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

// POSITION = false: lattice-word-index-segment; true: lattice-word-index-position.
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
  try {
    std::unique_ptr<Writer> writer;
    if (rank == 0) writer.reset(new Writer(wspecifier));  // only the writing rank opens (and truncates) the output
    Context ctx(ranks.device());
    ranks.InitComm(ctx);
    klat_pack_options popt = {co.lanes_per_lattice, 0, 0};
    SequentialCompactLatticeReader reader(rspecifier);
    int64_t num_lattices = 0;
    while (!reader.Done()) {
      // one super-batch: what `world` GPUs take in one go
      std::vector<std::string> keys;
      std::vector<CompactLattice> lats;
      std::vector<int64_t> arc_counts;
      int64_t arcs = 0;
      for (; !reader.Done() && (int64_t)keys.size() < (int64_t)co.batch_size * world && arcs < (int64_t)co.max_batch_arcs * world;
           reader.Next()) {
        keys.push_back(reader.Key());
        arc_counts.push_back(reader.Value().NumArcs());
        arcs += arc_counts.back();
        lats.push_back(std::move(reader.Value()));
        reader.FreeCurrent();
      }
      const int64_t n = (int64_t)keys.size();
      num_lattices += n;
      std::vector<int64_t> bounds((size_t)world + 1);
      Check(klat_partition_by_arcs(arc_counts.data(), n, world, bounds.data()), "klat_partition_by_arcs");
      const int64_t lo = bounds[(size_t)rank], hi = bounds[(size_t)rank + 1];
      for (int64_t u = 0; u < n; ++u)
        if (u < lo || u >= hi) lats[(size_t)u].DeleteStates();  // other ranks' lattices: only their arc counts mattered
      std::vector<CompactLattice> mine(std::make_move_iterator(lats.begin() + lo), std::make_move_iterator(lats.begin() + hi));
      lats.clear();
      if (io.Prunes() && !mine.empty()) {  // PruneLattice(beam, &clat), kwsbin2/lattice-word-index-segment.cc:49-51
        if (io.device_prune) {
          PruneLatticesOnDevice(ctx, io.beam, co.graph_scale, co.acoustic_scale, co.insertion_penalty, &mine);
        } else {
          for (auto& clat : mine)
            if (clat.Start() != CompactLattice::kNoStateId)
              PruneLattice(io.beam, co.graph_scale, co.acoustic_scale, co.insertion_penalty, &clat);
        }
      }
      LatticeBatch batch;
      std::vector<int32_t> state_len_all;
      for (auto& clat : mine) {
        std::vector<int32_t> times, state_len;
        if (POSITION) {
          CompactLattice dlat;
          if (clat.Start() != CompactLattice::kNoStateId) {
            TopSortCompactLatticeIfNeeded(&clat);
            DisambiguateStateInputSequenceLength(clat, &dlat, &state_len, false);  // kwsbin2/lattice-word-index-position.cc:64-65
            CompactLatticeStateTimes(dlat, &times);
          }
          batch.Add(dlat, &times);
          state_len_all.insert(state_len_all.end(), state_len.begin(), state_len.end());
        } else {
          if (clat.Start() != CompactLattice::kNoStateId) {
            TopSortCompactLatticeIfNeeded(&clat);
            CompactLatticeStateTimes(clat, &times);
          }
          batch.Add(clat, &times);
        }
        clat.DeleteStates();
      }
      // this rank's range on its GPU
      std::unique_ptr<DeviceBatch> db;
      std::vector<int64_t> off((size_t)(hi - lo) + 1, 0);
      int64_t nrec = 0;
      if (hi > lo) {
        db.reset(new DeviceBatch(ctx, batch, &popt));
        db->ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
        if (POSITION)
          Check(klat_position_index(db->get(), state_len_all.data(), &filter, off.data(), nullptr, 0, &nrec), "klat_position_index");
        else
          Check(klat_segment_index(db->get(), &filter, off.data(), nullptr, 0, &nrec), "klat_segment_index");
      }
      std::vector<Record> rec;
      std::vector<int64_t> all_off;
      auto fetch_mine = [&]() {
        rec.resize((size_t)std::max<int64_t>(nrec, 1));
        if (hi <= lo) return;
        if (POSITION)
          Check(klat_position_index(db->get(), state_len_all.data(), &filter, off.data(),
                                    reinterpret_cast<klat_position_record*>(rec.data()), (int64_t)rec.size(), &nrec),
                "klat_position_index");
        else
          Check(klat_segment_index(db->get(), &filter, off.data(), reinterpret_cast<klat_segment_record*>(rec.data()),
                                   (int64_t)rec.size(), &nrec),
                "klat_segment_index");
      };
      if (world == 1) {
        fetch_mine();
        all_off = off;
      } else {
        int64_t tu = 0, tr = 0;
        if (ranks.host_gather()) {
          fetch_mine();
          Check(klat_gather_records(ranks.comm(), rec.data(), 24, off.data(), hi - lo, 0, &tu, &tr), "klat_gather_records");
        } else if (hi > lo) {
          Check(klat_gather_index(ranks.comm(), db->get(), 0, &tu, &tr), "klat_gather_index");
        } else {
          Check(klat_gather_records(ranks.comm(), nullptr, 24, off.data(), 0, 0, &tu, &tr), "klat_gather_records");
        }
        if (rank == 0) {
          if (tu != n) KLAT_ERR("gathered " << tu << " utterances, expected " << n);
          all_off.assign((size_t)n + 1, 0);
          rec.resize((size_t)std::max<int64_t>(tr, 1));
          Check(klat_gather_fetch(ranks.comm(), all_off.data(), rec.data(), (int64_t)rec.size()), "klat_gather_fetch");
        }
      }
      if (rank == 0)
        for (int64_t u = 0; u < n; ++u)
          writer->Write(keys[(size_t)u], rec.data() + all_off[(size_t)u], all_off[(size_t)u + 1] - all_off[(size_t)u]);
      KLAT_VLOG(1, "Rank " << rank << "/" << world << ": lattices [" << lo << ", " << hi << ") of a super-batch of " << n
                           << ", " << batch.NumArcs() << " arcs, " << nrec << " records");
    }
    if (writer) writer->Close();
    KLAT_VLOG(1, "Done " << num_lattices << " lattices.");
  } catch (const std::exception& e) {
    if (world > 1) std::cerr << "rank " << rank << ": ";
    std::cerr << e.what();
    status = -1;
    ranks.Abort();
  }
  return ranks.Finish(status);
}

}  // namespace klat
#endif
