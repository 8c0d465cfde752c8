// lattice-word-index-position -- MI355X-native drop-in for kwsbin2/lattice-word-index-position.cc:
// same command line (:220-254) and output table (label,pos,t0,t1,logp :80-84,111).  Per lattice
// the host does TopSort -> DisambiguateStateInputSequenceLength -> state times (:61-67), the GPU
// the sweep and the (label,position) accumulator with the first-maximum segmentation (:165-184).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "This tool creates a positional inverted index of the given lattices, in the traditional "
        "meaning of \"position\" in the context of search engines: the probability that a word "
        "appears at some position within the transcription, for all transcriptions of the utterance.\n"
        "\n"
        "Usage: lattice-word-index-position [options] <lattice-rspecifier> <index-wspecifier>\n"
        " e.g.: lattice-word-index-position --acoustic-scale=0.1 ark:1.lats ark:1.index\n";
    ParseOptions po(usage);
    CommonOptions co;
    IndexOptions io;
    co.Register(&po);
    io.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    io.Check();
    const std::vector<int32_t> include = ParseIntList(io.include_str, "--include-words");
    const std::vector<int32_t> exclude = ParseIntList(io.exclude_str, "--exclude-words");
    const klat_label_filter filter = {include.data(), (int32_t)include.size(), exclude.data(), (int32_t)exclude.size()};

    PositionScoreWriter score_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = {co.lanes_per_lattice, 0, 0};

    std::vector<std::string> keys;
    std::vector<int32_t> state_len_all;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      std::vector<int64_t> off(keys.size() + 1, 0);
      int64_t n = 0;
      Check(klat_position_index(db.get(), state_len_all.data(), &filter, off.data(), nullptr, 0, &n), "klat_position_index");
      std::vector<klat_position_record> rec((size_t)std::max<int64_t>(n, 1));
      Check(klat_position_index(db.get(), state_len_all.data(), &filter, off.data(), rec.data(), (int64_t)rec.size(), &n),
            "klat_position_index");
      for (size_t u = 0; u < keys.size(); ++u) score_writer.Write(keys[u], rec.data() + off[u], off[u + 1] - off[u]);
      keys.clear();
      state_len_all.clear();
      batch.Clear();
    };

    for (PrunedLatticeReader reader(po.GetArg(1), ctx, co, io); !reader.Done(); reader.Next()) {
      CompactLattice& clat = reader.Value();
      CompactLattice dlat;
      std::vector<int32_t> times, state_len;
      if (clat.Start() != CompactLattice::kNoStateId) {
        TopSortCompactLatticeIfNeeded(&clat);
        DisambiguateStateInputSequenceLength(clat, &dlat, &state_len, false);
        CompactLatticeStateTimes(dlat, &times);
      }
      KLAT_VLOG(1, "Lattice " << reader.Key() << ": #states " << clat.NumStates() << " -> " << dlat.NumStates());
      keys.push_back(reader.Key());
      batch.Add(dlat, &times);
      state_len_all.insert(state_len_all.end(), state_len.begin(), state_len.end());
      reader.FreeCurrent();
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    score_writer.Close();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
