// klat-alphas-betas -- exercises interface B2 the way the reference's threaded tools do: the
// in-process function ComputeLatticeAlphasAndBetas(clat, false, &alpha, &beta) called concurrently
// from --num-threads worker threads, each on its own lattice
// (kwsbin2/lattice-word-index-segment.cc:181,260-275: TaskSequencer runs one LatticeScorerTask per
// lattice; results are emitted in submission order).  Prints, per lattice in input order,
//   key total alpha[S-1] beta[0] sum(alpha) sum(beta)     (finite entries only in the sums)
// Not one of the reference's tools: test plumbing for the re-entrancy of the C ABI underneath.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../lattice-functions.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Forward-backward of every lattice through ComputeLatticeAlphasAndBetas, from several threads.\n"
        "\n"
        "Usage: klat-alphas-betas [--num-threads=N] [--device=D] [--viterbi] <lattice-rspecifier>\n";
    ParseOptions po(usage);
    int32_t num_threads = 1, device = 0;
    bool viterbi = false;
    po.Register("viterbi", &viterbi, "Best-path (max) instead of sum: the routine's viterbi argument.");
    po.Register("num-threads", &num_threads, "Worker threads, one lattice each at a time.");
    po.Register("device", &device, "GPU device index.");
    po.Read(argc, argv);
    if (po.NumArgs() != 1) {
      po.PrintUsage();
      exit(1);
    }
    if (num_threads < 1) KLAT_ERR("--num-threads must be positive");
    SetDefaultDevice(device);
    std::vector<std::string> keys;
    std::vector<CompactLattice> lats;
    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next()) {
      keys.push_back(reader.Key());
      lats.push_back(std::move(reader.Value()));
      TopSortCompactLatticeIfNeeded(&lats.back());
    }
    struct Result {
      double total = 0, alpha_last = 0, beta0 = 0, sum_alpha = 0, sum_beta = 0;
      std::string error;
    };
    std::vector<Result> results(lats.size());
    // waves of short-lived threads, like TaskSequencer's: a thread per lattice, num_threads at once
    for (size_t begin = 0; begin < lats.size(); begin += (size_t)num_threads) {
      std::vector<std::thread> pool;
      for (size_t i = begin; i < std::min(lats.size(), begin + (size_t)num_threads); ++i)
        pool.emplace_back([&, i]() {
          try {
            std::vector<double> alpha, beta;
            Result& r = results[i];
            r.total = ComputeLatticeAlphasAndBetas(lats[i], viterbi, &alpha, &beta);
            r.alpha_last = alpha.back();
            r.beta0 = beta.front();
            for (double v : alpha)
              if (std::isfinite(v)) r.sum_alpha += v;
            for (double v : beta)
              if (std::isfinite(v)) r.sum_beta += v;
          } catch (const std::exception& e) {
            results[i].error = e.what();
          }
        });
      for (auto& t : pool) t.join();
    }
    for (size_t i = 0; i < lats.size(); ++i) {
      if (!results[i].error.empty()) KLAT_ERR("lattice " << keys[i] << ": " << results[i].error);
      std::printf("%s %.17g %.17g %.17g %.17g %.17g\n", keys[i].c_str(), results[i].total, results[i].alpha_last,
                  results[i].beta0, results[i].sum_alpha, results[i].sum_beta);
    }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
