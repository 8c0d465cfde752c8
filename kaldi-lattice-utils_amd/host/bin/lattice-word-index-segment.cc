// lattice-word-index-segment -- MI355X-native drop-in for kwsbin2/lattice-word-index-segment.cc:
// same command line (:205-239), same output table (BasicTupleVectorHolder<int32,int32,int32,double>,
// one record per input key in input order, empty ones included :64-69,128).  Lattices are
// top-sorted and timed on the host, then batched through the C ABI (sweep + segment accumulator);
// --devices shards them over several GPUs and gathers the records to the writing rank (sharded-index.h).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "sharded-index.h"
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
    ShardOptions so;
    so.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    io.Check();
    return RunIndexTool<false, SegmentScoreWriter, klat_segment_record>(co, io, so, po.GetArg(1), po.GetArg(2));
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
