// lattice-to-word-frame-post -- MI355X-native drop-in for latbin/lattice-to-word-frame-post.cc:
// same command line (:39-62), exit codes (:59-62,142-146) and Posterior output (log-posteriors as
// float32, per frame sorted by (logp desc, label asc) :121-135, one entry per input key :139).
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "sharded-index.h"
#include "tool-common.h"

// --devices=0-7 (not in the reference): one process per GPU, each takes a contiguous share of the
// archive and writes its own Posterior table (JOB in the wspecifier); see sharded-index.h.
int main(int argc, char* argv[]) {
  using namespace klat;
  std::unique_ptr<DeviceRanks> ranks;
  try {
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
    DevicesOption dopt;
    dopt.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    ranks.reset(new DeviceRanks(dopt.devices, co.device, "host"));  // forks: nothing before this line touches HIP
    const int32_t rank = ranks->rank(), world = ranks->world();
    PosteriorWriter posterior_writer(JobSpecifier(po.GetArg(2), rank, world));
    Context ctx(ranks->device());
    klat_pack_options popt = co.PackOptions();

    std::vector<std::string> keys;
    std::vector<int32_t> num_frames;
    LatticeBatch batch;
    PinnedPool pool;  // (declared after ctx: released while the HIP runtime is up)
    auto flush = [&]() {
      if (keys.empty()) return;
      WritePosteriors(ctx, batch, popt, co, keys, num_frames, &posterior_writer, &pool);
      keys.clear();
      num_frames.clear();
      batch.Clear();
    };

    RankArchiveReader reader(po.GetArg(1), rank, world, co.HostThreads(world));
    if (reader.mapped()) {
      // a mapped archive: a batch's entries straight into arrays on the host threads (no object per state)
      std::vector<FlatLattice> flats;
      std::vector<std::vector<int32_t>> times;
      while (reader.ReadFlatChunk((size_t)co.batch_size, co.max_batch_arcs, &keys, &flats)) {
        times.assign(flats.size(), std::vector<int32_t>());
        num_frames.assign(flats.size(), 0);
        std::vector<uint8_t> misaligned(flats.size(), 0);
        ParallelFor((int64_t)flats.size(), co.HostThreads(world), [&](int64_t i) {
          const FlatLattice& fl = flats[(size_t)i];
          if (fl.NumStates() == 0) return;
          num_frames[(size_t)i] = FlatLatticeStateTimes(fl, &times[(size_t)i]);
          const std::vector<int32_t>& t = times[(size_t)i];
          for (int32_t s = 0; s < fl.NumStates() && !misaligned[(size_t)i]; ++s)
            for (int32_t k = fl.row[(size_t)s]; k < fl.row[(size_t)s + 1]; ++k)
              if (fl.ilabel[(size_t)k] != 0 && t[(size_t)s] >= t[(size_t)fl.dst[(size_t)k]]) {
                misaligned[(size_t)i] = 1;
                break;
              }
        });
        for (size_t i = 0; i < flats.size(); ++i)
          if (misaligned[i]) KLAT_WARN("Lattice " << keys[i] << " is misaligned, a word with zero duration was found!");
        batch.AddManyFlat(flats, times, co.HostThreads(world));
        flush();
      }
    }
    for (; !reader.Done(); reader.Next()) {
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
    if (world > 1) KLAT_VLOG(1, "Rank " << rank << "/" << world << ": parsed " << reader.parsed() << " of " << reader.total() << " lattices");
    return ranks->Finish(0);
  } catch (const std::exception& e) {
    std::cerr << e.what();
    if (ranks) {
      ranks->Abort();
      return ranks->Finish(-1);
    }
    return -1;
  }
}
