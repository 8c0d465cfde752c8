// lattice-best-path2 -- MI355X-native drop-in for latbin/lattice-best-path2.cc (SURVEY.md 8f f3,
// BASELINE configs[2]): the 1-best transcription that minimises the position-wise expected error
//   sum_k (1 - P(y_k = w | x))   over the length-disambiguated, length-padded lattice.
// Same command line (:55-70), Int32Vector output (:75,199), log lines and exit codes.
//
// GPU: forward-backward of the padded lattices, the (label, position) accumulation -- the
// position-index kernels with an empty filter compute the reference's word_pos_acc (:121-139) --
// and the tropical shortest paths over arc costs 1 - P (:149-188; klat_nbest, one wave per lattice).
// Host: graph surgery before (ArcSort, DisambiguateStateInputSequenceLength,
// AddSequenceLengthDismabiguationSymbol :103-110) and the mapping of the accumulated posteriors
// back to the arcs.  Among paths of equal cost the kernel's back-pointer order decides; OpenFst's
// ShortestPath may break such ties differently (its queue discipline is not part of the
// reference's contract).
//
// --nbest=N (N <= 128; the reference stops at ShortestPath(.., 1), :185; BASELINE configs[2] asks
// for 100): the N best paths per utterance under keys "<utt>-<n>", n = 1..N as Kaldi's
// lattice-to-nbest names them, and their costs to an optional third table.  --nbest=1 (default) is
// the reference's output.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "sharded-index.h"
#include "tool-common.h"

// --devices=0-7 (not in the reference; BASELINE configs[2] shards this tool over the node's GPUs): one
// process per GPU, each takes a contiguous share of the archive and writes its own tables (JOB in the
// wspecifiers); see sharded-index.h.
int main(int argc, char* argv[]) {
  using namespace klat;
  std::unique_ptr<DeviceRanks> ranks;
  try {
    const char* usage =
        "Generate especial 1-best path through lattices, which minimizes a different loss-function "
        "other than the 0-1 error on the sequence: the number of position-wise errors, an upper "
        "bound on the expected Levenshtein distance given p(y|x).\n"
        "\n"
        "Usage: lattice-best-path2 [options] lat-rspecifier [transcriptions-wspecifier]\n"
        "       lattice-best-path2 --nbest=N [options] lat-rspecifier [transcriptions-wspecifier [costs-wspecifier]]\n"
        " e.g.: lattice-best-path2 --acoustic-scale=0.1 ark:1.lats ark,t:1.tra\n";
    ParseOptions po(usage);
    CommonOptions co;
    co.Register(&po);
    int32_t nbest = 1;
    po.Register("nbest", &nbest, "Number of best paths per utterance (1 = the reference's output; up to 128).");
    DevicesOption dopt;
    dopt.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() < 1 || po.NumArgs() > 3) {
      po.PrintUsage();
      exit(1);
    }
    if (nbest < 1 || nbest > KLAT_NBEST_MAX) KLAT_ERR("--nbest must be in [1, " << KLAT_NBEST_MAX << "]");
    std::unique_ptr<Int32VectorWriter> transcriptions_writer;
    std::unique_ptr<BaseFloatWriter> costs_writer;
    ranks.reset(new DeviceRanks(dopt.devices, co.device, "host"));  // forks: nothing before this line touches HIP
    const int32_t rank = ranks->rank(), world = ranks->world();
    if (po.NumArgs() > 1) transcriptions_writer.reset(new Int32VectorWriter(JobSpecifier(po.GetArg(2), rank, world)));
    if (po.NumArgs() > 2) costs_writer.reset(new BaseFloatWriter(JobSpecifier(po.GetArg(3), rank, world)));
    Context ctx(ranks->device());
    klat_pack_options popt = co.PackOptions();

    double total_cost = 0.0;
    int64_t total_frames = 0;
    std::vector<std::string> keys;
    std::vector<int32_t> frames;
    std::vector<CompactLattice> lats;  // padded lattices: the shortest path walks their arcs
    std::vector<std::vector<int32_t>> lens;
    LatticeBatch batch;

    auto flush = [&]() {
      if (keys.empty()) return;
      DeviceBatch db(ctx, batch, &popt);
      db.ForwardBackward(co.graph_scale, co.acoustic_scale, co.insertion_penalty);
      const size_t U = keys.size();
      std::vector<int32_t> state_len((size_t)std::max<int64_t>(batch.NumStates(), 1));
      for (size_t u = 0; u < U; ++u)
        std::copy(lens[u].begin(), lens[u].end(), state_len.begin() + batch.StateOff()[u]);
      klat_label_filter filter = {nullptr, 0, nullptr, 0};
      std::vector<int64_t> record_off(U + 1, 0);
      int64_t n = 0;
      Check(klat_position_index(db.get(), state_len.data(), &filter, record_off.data(), nullptr, 0, &n), "klat_position_index");
      std::vector<klat_position_record> rec((size_t)std::max<int64_t>(n, 1));
      Check(klat_position_index(db.get(), state_len.data(), &filter, record_off.data(), rec.data(), (int64_t)rec.size(), &n),
            "klat_position_index");
      std::vector<double> total(U), tot_bwd(U);
      Check(klat_get_totals(db.get(), total.data(), nullptr, tot_bwd.data()), "klat_get_totals");

      // arc costs 1 - P(label, position) = exp(LogSub(0, log P)) stored as float (:171-173); log P
      // normalised by bw[Start()] and clipped at 0 (:143-146); epsilon arcs cost nothing (:166-168)
      std::vector<float> arc_cost((size_t)std::max<int64_t>(batch.NumArcs(), 1), 0.0f);
      int32_t max_labels = 1;
      for (size_t u = 0; u < U; ++u) {
        const CompactLattice& clat = lats[u];
        const std::vector<int32_t>& len = lens[u];
        std::map<std::pair<int32_t, int32_t>, double> post;
        for (int64_t i = record_off[u]; i < record_off[u + 1]; ++i)
          post[std::make_pair(rec[i].label, rec[i].pos)] = std::min(0.0, rec[i].logp + total[u] - tot_bwd[u]);
        int64_t a = batch.ArcOff()[u];
        for (int32_t s = 0; s < clat.NumStates(); ++s) {
          max_labels = std::max(max_labels, len[(size_t)s] + 2);
          for (const auto& arc : clat.State(s).arcs) {
            if (arc.olabel != 0) {
              const auto it = post.find(std::make_pair(arc.olabel, len[(size_t)arc.nextstate]));
              const double p = it == post.end() ? -std::numeric_limits<double>::infinity() : it->second;
              arc_cost[(size_t)a] = (float)std::exp(p >= 0.0 ? -std::numeric_limits<double>::infinity() : std::log1p(-std::exp(p)));
            }
            ++a;
          }
        }
      }
      std::vector<int32_t> num_paths(U), path_len(U * (size_t)nbest), path_labels(U * (size_t)nbest * (size_t)max_labels);
      std::vector<float> path_cost(U * (size_t)nbest);
      Check(klat_nbest(db.get(), arc_cost.data(), nbest, max_labels, num_paths.data(), path_cost.data(), path_len.data(),
                       path_labels.data()),
            "klat_nbest");

      for (size_t u = 0; u < U; ++u) {
        float best_cost = 0.0f;
        for (int32_t n = 0; n < std::max(1, num_paths[u]); ++n) {
          std::vector<int32_t> transcript;
          float cost = 0.0f;
          if (n < num_paths[u]) {
            const size_t slot = u * (size_t)nbest + (size_t)n;
            cost = path_cost[slot];
            // Filter out epsilons and special disambiguation symbols (:191-196).
            for (int32_t k = 0; k < path_len[slot]; ++k)
              if (path_labels[slot * (size_t)max_labels + (size_t)k] != -1) transcript.push_back(path_labels[slot * (size_t)max_labels + (size_t)k]);
          }
          if (n == 0) best_cost = cost;
          const std::string key = nbest == 1 ? keys[u] : keys[u] + "-" + std::to_string(n + 1);
          if (transcriptions_writer) transcriptions_writer->Write(key, transcript);
          if (costs_writer) costs_writer->Write(key, cost);
        }
        total_cost += best_cost;
        total_frames += frames[u];
        KLAT_LOG("For utterance " << keys[u] << ", best cost is " << best_cost << " over " << frames[u] << " frames.");
      }
      keys.clear();
      frames.clear();
      lats.clear();
      lens.clear();
      batch.Clear();
    };

    RankArchiveReader reader(po.GetArg(1), rank, world, co.HostThreads(world));
    for (; !reader.Done(); reader.Next()) {
      CompactLattice clat;
      std::vector<int32_t> state_input_length;
      int32_t lattice_frames = 0;
      {
        CompactLattice& tmp = reader.Value();
        if (tmp.Start() != CompactLattice::kNoStateId) {
          TopSortCompactLatticeIfNeeded(&tmp);
          std::vector<int32_t> state_times;
          lattice_frames = CompactLatticeStateTimes(tmp, &state_times);
          ArcSortByOlabel(&tmp);
        }
        DisambiguateStateInputSequenceLength(tmp, &clat, &state_input_length, false);
        AddSequenceLengthDisambiguationSymbol(&clat, &state_input_length);
        // The reference applies the insertion penalty before the padding exists (:93-96); here the
        // GPU applies it to arcs with a non-zero INPUT label, so the padding arcs keep only their
        // output label (everything downstream reads olabel, :128,165).
        for (auto& st : clat.States())
          for (auto& arc : st.arcs)
            if (arc.olabel == -1) arc.ilabel = 0;
        reader.FreeCurrent();
      }
      keys.push_back(reader.Key());
      frames.push_back(lattice_frames);
      batch.Add(clat, &state_input_length);
      lats.push_back(std::move(clat));
      lens.push_back(std::move(state_input_length));
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    if (transcriptions_writer) transcriptions_writer->Close();
    if (costs_writer) costs_writer->Close();
    KLAT_LOG("Overall cost per frame is " << (total_frames ? total_cost / total_frames : 0.0) << " over " << total_frames
                                          << " frames.");
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
