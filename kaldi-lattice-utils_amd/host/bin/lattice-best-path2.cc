// lattice-best-path2 -- MI355X-native drop-in for latbin/lattice-best-path2.cc (SURVEY.md 8f f3,
// BASELINE configs[2]): the 1-best transcription that minimises the position-wise expected error
//   sum_k (1 - P(y_k = w | x))   over the length-disambiguated, length-padded lattice.
// Same command line (:55-70), Int32Vector output (:75,199), log lines and exit codes.
//
// GPU: forward-backward of the padded lattices and the (label, position) accumulation -- the
// position-index kernels with an empty filter compute the reference's word_pos_acc (:121-139).
// Host: graph surgery before (ArcSort, DisambiguateStateInputSequenceLength,
// AddSequenceLengthDismabiguationSymbol :103-110) and, after, the tropical shortest path over arc
// costs 1 - P (:149-188), a single pass over the topologically sorted states.  Among paths of
// equal cost the first one found in (state, arc) order is kept; OpenFst's ShortestPath may break
// such ties differently (its queue discipline is not part of the reference's contract).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Generate especial 1-best path through lattices, which minimizes a different loss-function "
        "other than the 0-1 error on the sequence: the number of position-wise errors, an upper "
        "bound on the expected Levenshtein distance given p(y|x).\n"
        "\n"
        "Usage: lattice-best-path2 [options] lat-rspecifier [transcriptions-wspecifier]\n"
        " e.g.: lattice-best-path2 --acoustic-scale=0.1 ark:1.lats ark,t:1.tra\n";
    ParseOptions po(usage);
    CommonOptions co;
    co.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() < 1 || po.NumArgs() > 2) {
      po.PrintUsage();
      exit(1);
    }
    std::unique_ptr<Int32VectorWriter> transcriptions_writer;
    if (po.NumArgs() > 1) transcriptions_writer.reset(new Int32VectorWriter(po.GetArg(2)));
    Context ctx(co.device);
    klat_pack_options popt = {co.lanes_per_lattice, 0, 0};

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

      for (size_t u = 0; u < U; ++u) {
        const CompactLattice& clat = lats[u];
        const std::vector<int32_t>& len = lens[u];
        std::vector<int32_t> transcript;
        float transcript_cost = 0.0f;
        if (clat.NumStates() > 0) {
          // log P(w | x, k), normalised by bw[Start()] and clipped at 0 (:143-146)
          std::map<std::pair<int32_t, int32_t>, double> post;
          for (int64_t i = record_off[u]; i < record_off[u + 1]; ++i)
            post[std::make_pair(rec[i].label, rec[i].pos)] = std::min(0.0, rec[i].logp + total[u] - tot_bwd[u]);
          // tropical shortest path, arc cost 1 - P = exp(LogSub(0, log P)) stored as float (:171-173)
          const int32_t S = clat.NumStates();
          const float inf = std::numeric_limits<float>::infinity();
          std::vector<float> dist((size_t)S, inf);
          std::vector<int32_t> parent((size_t)S, -1), parent_label((size_t)S, 0);
          dist[0] = 0.0f;
          int32_t best_final = -1;
          float best_cost = inf;
          for (int32_t s = 0; s < S; ++s) {
            if (dist[(size_t)s] == inf) continue;
            const CompactLatticeState& st = clat.State(s);
            if (st.IsFinal() && dist[(size_t)s] < best_cost) {
              best_cost = dist[(size_t)s];
              best_final = s;
            }
            for (const auto& arc : st.arcs) {
              float w = 0.0f;
              if (arc.olabel != 0) {
                const auto it = post.find(std::make_pair(arc.olabel, len[(size_t)arc.nextstate]));
                const double p = it == post.end() ? -std::numeric_limits<double>::infinity() : it->second;
                w = (float)std::exp(p >= 0.0 ? -std::numeric_limits<double>::infinity() : std::log1p(-std::exp(p)));
              }
              const float nd = dist[(size_t)s] + w;
              if (nd < dist[(size_t)arc.nextstate]) {
                dist[(size_t)arc.nextstate] = nd;
                parent[(size_t)arc.nextstate] = s;
                parent_label[(size_t)arc.nextstate] = arc.olabel;
              }
            }
          }
          if (best_final >= 0) {
            transcript_cost = best_cost;
            for (int32_t s = best_final; parent[(size_t)s] >= 0; s = parent[(size_t)s])
              if (parent_label[(size_t)s] != 0 && parent_label[(size_t)s] != -1) transcript.push_back(parent_label[(size_t)s]);
            std::reverse(transcript.begin(), transcript.end());
          }
        }
        if (transcriptions_writer) transcriptions_writer->Write(keys[u], transcript);
        total_cost += transcript_cost;
        total_frames += frames[u];
        KLAT_LOG("For utterance " << keys[u] << ", best cost is " << transcript_cost << " over " << frames[u] << " frames.");
      }
      keys.clear();
      frames.clear();
      lats.clear();
      lens.clear();
      batch.Clear();
    };

    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next()) {
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
    KLAT_LOG("Overall cost per frame is " << (total_frames ? total_cost / total_frames : 0.0) << " over " << total_frames
                                          << " frames.");
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
