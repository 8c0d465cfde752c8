// lattice-to-word-frame-post -- MI355X-native drop-in for latbin/lattice-to-word-frame-post.cc:
// same command line (:39-62), exit codes (:59-62,142-146) and Posterior output (log-posteriors as
// float32, per frame sorted by (logp desc, label asc) :121-135, one entry per input key :139).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Compute the posterior log-probability of each word for each given utterance frame. That is, "
        "we compute log P(a_i = v | x), for all possible utterance frames i and words v.\n"
        "\n"
        "Usage: lattice-to-word-frame-post [options] lat-rspecifier post-wspecifier\n"
        " e.g.: lattice-to-word-frame-post --acoustic-scale=0.1 ark:1.lats ark:1.word.pos.post\n"
        "See also: lattice-to-word-position-post\n";
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
    klat_pack_options popt = {co.lanes_per_lattice, 0, 0};

    std::vector<std::string> keys;
    std::vector<int32_t> num_frames;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      int64_t T = 0;
      for (int32_t f : num_frames) T += f;
      std::vector<int64_t> frame_off((size_t)T + 1, 0);
      int64_t n = 0;
      Check(klat_frame_post(db.get(), num_frames.data(), frame_off.data(), nullptr, nullptr, 0, &n), "klat_frame_post");
      std::vector<int32_t> labels((size_t)std::max<int64_t>(n, 1));
      std::vector<float> logps((size_t)std::max<int64_t>(n, 1));
      Check(klat_frame_post(db.get(), num_frames.data(), frame_off.data(), labels.data(), logps.data(), (int64_t)labels.size(), &n),
            "klat_frame_post");
      int64_t fb = 0;
      for (size_t u = 0; u < keys.size(); ++u) {
        PosteriorWriter::T post((size_t)num_frames[u]);
        for (int32_t k = 0; k < num_frames[u]; ++k)
          for (int64_t i = frame_off[fb + k]; i < frame_off[fb + k + 1]; ++i) post[k].emplace_back(labels[i], logps[i]);
        posterior_writer.Write(keys[u], post);
        fb += num_frames[u];
      }
      keys.clear();
      num_frames.clear();
      batch.Clear();
    };

    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next()) {
      CompactLattice& clat = reader.Value();
      std::vector<int32_t> times;
      int32_t total_frames = 0;
      if (clat.Start() != CompactLattice::kNoStateId) {
        TopSortCompactLatticeIfNeeded(&clat);
        total_frames = CompactLatticeStateTimes(clat, &times);
        bool misaligned = false;
        for (int32_t s = 0; s < clat.NumStates() && !misaligned; ++s)
          for (const auto& arc : clat.State(s).arcs)
            if (arc.ilabel != 0 && times[s] >= times[arc.nextstate]) {
              misaligned = true;
              break;
            }
        if (misaligned) KLAT_WARN("Lattice " << reader.Key() << " is misaligned, a word with zero duration was found!");
      }
      keys.push_back(reader.Key());
      num_frames.push_back(total_frames);
      batch.Add(clat, &times);
      reader.FreeCurrent();
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
