// lattice-word-index-position -- MI355X-native drop-in for kwsbin2/lattice-word-index-position.cc:
// same command line (:220-254) and output table (label,pos,t0,t1,logp :80-84,111).  Per lattice
// the host does TopSort -> DisambiguateStateInputSequenceLength -> state times (:61-67), the GPU
// the sweep and the (label,position) accumulator with the first-maximum segmentation (:165-184).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "sharded-index.h"
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
    ShardOptions so;
    so.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    io.Check();
    return RunIndexTool<true, PositionScoreWriter, klat_position_record>(co, io, so, po.GetArg(1), po.GetArg(2));
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
