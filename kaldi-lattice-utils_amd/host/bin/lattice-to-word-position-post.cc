// lattice-to-word-position-post -- MI355X-native drop-in for
// latbin/lattice-to-word-position-post.cc: same command line (:38-63), exit codes and Posterior
// output (entry n-1 of an utterance = the words at position n with float32 log-posteriors, sorted
// by (logp desc, label asc) :128-135; one entry per input key :138).
//
// The reference keys its accumulator by (state_input_length[nextstate], ilabel) (:104-117) after
// DisambiguateStateInputSequenceLength (:94-95).  In the disambiguated lattice a word arc leads
// from length n-1 to length n, so with the per-state lengths handed to the device as "state times"
// the arc covers exactly the one "frame" n-1: the frame-posterior kernels of
// lattice-to-word-frame-post compute this tool's output unchanged.
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Compute the posterior log-probability of each word at each position of the transcription. "
        "That is, we compute log P(y_n = v | x), for all positions n and words v.\n"
        "\n"
        "Usage: lattice-to-word-position-post [options] lat-rspecifier post-wspecifier\n"
        " e.g.: lattice-to-word-position-post --acoustic-scale=0.1 ark:1.lats ark:1.word.pos.post\n"
        "See also: lattice-to-word-frame-post\n";
    ParseOptions po(usage);
    CommonOptions co;
    co.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2 && po.NumArgs() != 3) {  // the reference accepts (and ignores) a third argument (:60)
      po.PrintUsage();
      exit(1);
    }
    PosteriorWriter posterior_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = co.PackOptions();

    std::vector<std::string> keys;
    std::vector<int32_t> max_len;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      WritePosteriors(ctx, batch, popt, co, keys, max_len, &posterior_writer);
      keys.clear();
      max_len.clear();
      batch.Clear();
    };

    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next()) {
      CompactLattice clat;
      std::vector<int32_t> state_input_length;
      int32_t len = 0;
      {
        CompactLattice& tmp = reader.Value();
        if (tmp.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&tmp);
        len = DisambiguateStateInputSequenceLength(tmp, &clat, &state_input_length, false);
        reader.FreeCurrent();
      }
      keys.push_back(reader.Key());
      max_len.push_back(len);
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
