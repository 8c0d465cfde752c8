// lattice-to-transcript-length-dist -- MI355X-native drop-in for
// latbin/lattice-to-transcript-length-dist.cc: same command line (:36-58), exit codes and
// Posterior output (one entry per utterance holding (length, float32 logp) pairs sorted by
// (logp desc, length asc) :117-124).
//
// The forward-backward over the length-disambiguated lattices (:94-95) runs on the GPU, batched;
// what the reference does afterwards touches only the final states (:98-109: acc[len(u)] =
// LogAdd(.., fw[u] - ConvertToCost(Final(u)))), a few doubles per lattice, and stays on the host.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

namespace {

// LogAdd(double,double) [upstream Kaldi base/kaldi-math.h, SURVEY.md A.2]
double LogAdd(double x, double y) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= std::log(2.220446049250313e-16)) return x + std::log1p(std::exp(diff));
  return x;  // also when diff is NaN (both -inf)
}

}  // namespace

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Compute the posterior log-probability of the transcription length (number of words) of "
        "each utterance. That is, we compute log P(|y| = n | x), for all lengths n.\n"
        "\n"
        "Usage: lattice-to-transcript-length-dist [options] lat-rspecifier post-wspecifier\n"
        " e.g.: lattice-to-transcript-length-dist --acoustic-scale=0.1 ark:1.lats ark:1.len.post\n";
    ParseOptions po(usage);
    CommonOptions co;
    co.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    PosteriorWriter posterior_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = co.PackOptions();
    const bool scaled = co.acoustic_scale != 1.0f || co.graph_scale != 1.0f;

    struct Final {
      int32_t state, length;
      double cost;  // ConvertToCost(Final(u)) after ScaleLattice (float rounding as the reference's)
    };
    std::vector<std::string> keys;
    std::vector<std::vector<Final>> finals;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      std::vector<double> alpha((size_t)std::max<int64_t>(batch.NumStates(), 1));
      std::vector<double> total((size_t)keys.size());
      Check(klat_get_alpha_beta(db.get(), alpha.data(), nullptr), "klat_get_alpha_beta");
      Check(klat_get_totals(db.get(), total.data(), nullptr, nullptr), "klat_get_totals");
      for (size_t u = 0; u < keys.size(); ++u) {
        const int64_t s0 = batch.StateOff()[u];
        std::map<int32_t, double> acc;  // states in order: the reference's StateIterator loop :98
        for (const Final& f : finals[u]) {
          const double lkh = alpha[(size_t)(s0 + f.state)] - f.cost;
          auto r = acc.emplace(f.length, lkh);
          if (!r.second) r.first->second = LogAdd(r.first->second, lkh);
        }
        PosteriorWriter::T posterior(1);
        for (const auto& kv : acc) posterior[0].emplace_back(kv.first, (float)(kv.second - total[u]));
        std::sort(posterior[0].begin(), posterior[0].end(),
                  [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& b) {
                    if (a.second != b.second) return b.second < a.second;
                    return a.first < b.first;
                  });
        posterior_writer.Write(keys[u], posterior);
      }
      keys.clear();
      finals.clear();
      batch.Clear();
    };

    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next()) {
      CompactLattice clat;
      std::vector<int32_t> state_input_length;
      {
        CompactLattice& tmp = reader.Value();
        if (tmp.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&tmp);
        DisambiguateStateInputSequenceLength(tmp, &clat, &state_input_length, false);
        reader.FreeCurrent();
      }
      std::vector<Final> fin;
      for (int32_t s = 0; s < clat.NumStates(); ++s) {
        const auto& st = clat.State(s);
        if (!st.IsFinal()) continue;
        float g = st.final_graph, a = st.final_acoustic;
        if (scaled) {  // ScaleLattice: float((double)scale * (double)x), one rounding; Value1 == +inf stays
          const bool zero = g == std::numeric_limits<float>::infinity();
          a = zero ? a : co.acoustic_scale * a;
          g = zero ? g : co.graph_scale * g;
        }
        fin.push_back(Final{s, state_input_length[(size_t)s], (double)g + (double)a});
      }
      keys.push_back(reader.Key());
      finals.push_back(std::move(fin));
      batch.Add(clat, &state_input_length);
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    posterior_writer.Close();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
