// lattice-word-index-utterance -- MI355X-native drop-in for kwsbin2/lattice-word-index-utterance.cc:
// same command line (:198-247), same output table (BasicTupleVectorHolder<int32,double>, records
// sorted by (logp desc, label asc) :291-300, one entry per input key).  The reference composes the
// lattice with a two-state query automaton once per word and runs a backward pass over each
// product (:143-183); here the lattices are batched through the C ABI and every word of a lattice
// is one lane of klat_utterance_scores (csrc/klat_utt.hip).  --rho-label and the TaskSequencer
// options are accepted for command-line compatibility and have nothing left to configure.
#include <algorithm>
#include <cstdlib>
#include <set>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Create an inverted index of the given lattices, where the score of each word is the "
        "probability that the word is in the transcription of the signal. I.e. P(R = 1 | x, v), "
        "where x is the image and v is the word.\n"
        "\n"
        "Usage: lattice-word-index-utterance [options] <lattice-rspecifier> <index-wspecifier>\n"
        " e.g.: lattice-word-index-utterance --acoustic-scale=0.1 ark:1.lats ark:1.index\n";
    ParseOptions po(usage);
    CommonOptions co;
    IndexOptions io;
    int32_t rho_label = std::numeric_limits<int32_t>::max();
    co.Register(&po);
    io.Register(&po);
    po.Register("rho-label", &rho_label, "Accepted for compatibility (no composition takes place here).");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    io.Check();
    const std::vector<int32_t> inc_v = ParseIntList(io.include_str, "--include-words");
    const std::vector<int32_t> exc_v = ParseIntList(io.exclude_str, "--exclude-words");
    const std::set<int32_t> include(inc_v.begin(), inc_v.end()), exclude(exc_v.begin(), exc_v.end());

    UtteranceScoreWriter score_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = co.PackOptions();

    std::vector<std::string> keys;
    std::vector<int64_t> word_off(1, 0);
    std::vector<int32_t> words;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      std::vector<double> logp(std::max<size_t>(words.size(), 1));
      Check(klat_utterance_scores(db.get(), word_off.data(), words.data(), logp.data()), "klat_utterance_scores");
      for (size_t u = 0; u < keys.size(); ++u) {
        std::vector<std::pair<int32_t, double>> rec;
        for (int64_t k = word_off[u]; k < word_off[u + 1]; ++k) {
          if (logp[(size_t)k] > 0.0)
            KLAT_WARN("Lattice " << keys[u] << ": found word (" << words[(size_t)k]
                                 << ") with higher likelihood than the total likelihood (by " << logp[(size_t)k] << ")");
          rec.emplace_back(words[(size_t)k], logp[(size_t)k]);
        }
        std::sort(rec.begin(), rec.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) {
          if (b.second != a.second) return b.second < a.second;
          return a.first < b.first;
        });
        score_writer.Write(keys[u], rec);
      }
      keys.clear();
      word_off.assign(1, 0);
      words.clear();
      batch.Clear();
    };

    for (PrunedLatticeReader reader(po.GetArg(1), ctx, co, io); !reader.Done(); reader.Next()) {
      CompactLattice& clat = reader.Value();
      std::set<int32_t> symbols;  // GetOutputSymbols(clat, false): distinct non-epsilon output labels
      if (clat.Start() != CompactLattice::kNoStateId) {
        TopSortCompactLatticeIfNeeded(&clat);
        for (const auto& st : clat.States())
          for (const auto& arc : st.arcs)
            if (arc.olabel != 0) symbols.insert(arc.olabel);
      }
      for (int32_t s : symbols)
        if (include.empty() ? exclude.count(s) == 0 : include.count(s) != 0) words.push_back(s);
      word_off.push_back((int64_t)words.size());
      keys.push_back(reader.Key());
      batch.Add(clat, nullptr);
      reader.FreeCurrent();
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    score_writer.Close();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return -1;
  }
}
