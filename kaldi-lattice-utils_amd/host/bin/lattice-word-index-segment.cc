// lattice-word-index-segment -- MI355X-native drop-in for kwsbin2/lattice-word-index-segment.cc:
// same command line (:205-239), same output table (BasicTupleVectorHolder<int32,int32,int32,double>,
// one record per input key in input order, empty ones included :64-69,128).  Lattices are
// top-sorted and timed on the host, then batched through the C ABI (sweep + segment accumulator).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "This tool is used to create a positional inverted index of the given lattices, where the "
        "score of each word in a segment is the probability that the word occurs in any of the "
        "transcriptions of the utterance at that specific time position (segment).\n"
        "\n"
        "Usage: lattice-word-index-segment [options] <lattice-rspecifier> <index-wspecifier>\n"
        " e.g.: lattice-word-index-segment --acoustic-scale=0.1 ark:1.lats ark:1.index\n";
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

    SegmentScoreWriter score_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = {co.lanes_per_lattice, 0, 0};

    std::vector<std::string> keys;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      std::vector<int64_t> off(keys.size() + 1, 0);
      int64_t n = 0;
      Check(klat_segment_index(db.get(), &filter, off.data(), nullptr, 0, &n), "klat_segment_index");
      std::vector<klat_segment_record> rec((size_t)std::max<int64_t>(n, 1));
      Check(klat_segment_index(db.get(), &filter, off.data(), rec.data(), (int64_t)rec.size(), &n), "klat_segment_index");
      for (size_t u = 0; u < keys.size(); ++u) score_writer.Write(keys[u], rec.data() + off[u], off[u + 1] - off[u]);
      KLAT_VLOG(1, "Processed a batch of " << keys.size() << " lattices, " << batch.NumArcs() << " arcs, " << n << " records");
      keys.clear();
      batch.Clear();
    };

    int64_t num_lattices = 0;
    for (PrunedLatticeReader reader(po.GetArg(1), ctx, co, io); !reader.Done(); reader.Next(), ++num_lattices) {
      CompactLattice& clat = reader.Value();
      KLAT_VLOG(1, "Lattice " << reader.Key() << ": #states " << clat.NumStates() << " #arcs " << clat.NumArcs());
      std::vector<int32_t> times;
      if (clat.Start() != CompactLattice::kNoStateId) {
        TopSortCompactLatticeIfNeeded(&clat);
        CompactLatticeStateTimes(clat, &times);
      }
      keys.push_back(reader.Key());
      batch.Add(clat, &times);
      reader.FreeCurrent();
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    score_writer.Close();
    KLAT_VLOG(1, "Done " << num_lattices << " lattices.");
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
