// lattice-prune-dyn-beam -- MI355X-native drop-in for latbin/lattice-prune-dyn-beam.cc (BASELINE
// configs[4], first half): reduce the pruning beam of every lattice by --beam-ratio until it has
// at most --max-arcs arcs and --max-states states (:169-186).  Same command line (:107-139) and
// CompactLattice output, written in the original scale (:188-196).
//
// GPU: the Viterbi forward / backward costs of the whole batch (klat_viterbi, the tropical twin of
// the sweep) and ComputeLatticeBeam (:166).  Pruning with a smaller beam never changes the best
// path nor the best cost through a surviving arc, so the costs of ONE sweep decide every iteration
// of the reference's loop: an arc survives beam b iff fw[src] + (cost + bw[dst]) <= best + b, a
// state iff it keeps an arc (or its final weight).  The loop itself (:169-186) then only counts,
// on the host; the masks of the last beam are applied and the lattice Connect()ed.
// The lattices keep their stored weights throughout (the reference scales, prunes and scales back
// :188-193, which perturbs the floats by an ulp or two).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "../lattice-functions.h"
#include "../table-writers.h"
#include "tool-common.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    const char* usage =
        "Iteratively reduce the beam of the lattice until a maximum number of arcs and states is achieved.\n"
        "\n"
        "Usage: lattice-prune-dyn-beam [options] lat-rspecifier lat-wspecifier\n"
        " e.g.: lattice-prune-dyn-beam --acoustic-scale=0.1 --max-arcs=1000 ark:1.lats ark:1.pruned.lats\n";
    ParseOptions po(usage);
    CommonOptions co;
    co.Register(&po);
    int32_t max_num_arcs = std::numeric_limits<int32_t>::max(), max_num_states = std::numeric_limits<int32_t>::max();
    float beam_ratio = 0.9f, min_beam = 1e-3f;
    po.Register("beam-ratio", &beam_ratio, "Reduce the maximum beam by this ratio at each iteration.");
    po.Register("min-beam", &min_beam, "Minimum beam threshold");
    po.Register("max-arcs", &max_num_arcs, "Maximum number of arcs of each lattice.");
    po.Register("max-states", &max_num_states, "Maximum number of states of each lattice.");
    po.Read(argc, argv);
    if (po.NumArgs() < 2) {
      po.PrintUsage();
      exit(1);
    }
    if (beam_ratio <= 0.0f || beam_ratio >= 1.0f) KLAT_ERR("--beam_ratio must be in the open range (0.0, 1.0).");
    CompactLatticeWriter lattice_writer(po.GetArg(2));
    Context ctx(co.device);
    klat_pack_options popt = co.PackOptions();
    const double inf = std::numeric_limits<double>::infinity();

    std::vector<std::string> keys;
    std::vector<CompactLattice> lats;
    LatticeBatch batch;
    auto flush = [&]() {
      if (keys.empty()) return;
      std::vector<double> fw((size_t)std::max<int64_t>(batch.NumStates(), 1)), bw(fw.size()), best(keys.size()), lattice_beam(keys.size());
      if (batch.NumStates() > 0) {
        DeviceBatch db(ctx, batch, &popt);
        klat_scales sc = {co.graph_scale, co.acoustic_scale, co.insertion_penalty};
        Check(klat_viterbi(db.get(), &sc), "klat_viterbi");
        Check(klat_get_viterbi(db.get(), fw.data(), bw.data(), best.data()), "klat_get_viterbi");
        Check(klat_prune_mask(db.get(), 1.0f, nullptr, nullptr, lattice_beam.data()), "klat_prune_mask");
      }
      for (size_t u = 0; u < keys.size(); ++u) {
        CompactLattice& clat = lats[u];
        const int32_t S = clat.NumStates();
        const int64_t s0 = batch.StateOff()[u];
        const int32_t original_num_states = S;
        const int64_t original_num_arcs = clat.NumArcs();
        const double original_beam = S > 0 ? lattice_beam[u] : 0.0;
        double beam = original_beam;
        int64_t num_arcs = original_num_arcs;
        int32_t num_states = original_num_states;
        if (S > 0 && beam > min_beam && (num_arcs > max_num_arcs || num_states > max_num_states)) {
          // cost of the best complete path through every arc / final weight, minus the best path
          std::vector<std::vector<double>> excess((size_t)S);
          std::vector<double> final_excess((size_t)S, inf), state_excess((size_t)S, inf);
          for (int32_t s = 0; s < S; ++s) {
            const CompactLatticeState& st = clat.State(s);
            const double f = fw[(size_t)(s0 + s)];
            excess[(size_t)s].resize(st.arcs.size());
            for (size_t i = 0; i < st.arcs.size(); ++i) {
              const auto& arc = st.arcs[i];
              const double c = ScaledCost(arc.graph, arc.acoustic, arc.ilabel != 0, co.graph_scale, co.acoustic_scale, co.insertion_penalty);
              excess[(size_t)s][i] = f + (c + bw[(size_t)(s0 + arc.nextstate)]);
            }
            if (st.IsFinal())
              final_excess[(size_t)s] = f + ScaledCost(st.final_graph, st.final_acoustic, false, co.graph_scale, co.acoustic_scale, 0.0f);
          }
          double cutoff = inf;
          auto count = [&]() {  // what PruneLattice + Connect leave at this cutoff
            std::fill(state_excess.begin(), state_excess.end(), inf);
            num_arcs = 0;
            for (int32_t s = 0; s < S; ++s) {
              if (final_excess[(size_t)s] <= cutoff && final_excess[(size_t)s] < inf) state_excess[(size_t)s] = 0.0;
              for (size_t i = 0; i < excess[(size_t)s].size(); ++i)
                if (excess[(size_t)s][i] <= cutoff && excess[(size_t)s][i] < inf) {  // Connect drops dead arcs
                  ++num_arcs;
                  state_excess[(size_t)s] = 0.0;
                  state_excess[(size_t)clat.State(s).arcs[i].nextstate] = 0.0;
                }
            }
            num_states = 0;
            for (int32_t s = 0; s < S; ++s) num_states += state_excess[(size_t)s] == 0.0;
          };
          for (size_t n_try = 1; beam > min_beam && (num_arcs > max_num_arcs || num_states > max_num_states); ++n_try) {
            if (n_try > 1 && beam == inf) {  // the reference never leaves this loop (:168-172)
              KLAT_LOG("WARNING: lattice " << keys[u] << " has arcs on no complete path (beam = inf) and is still over the limits "
                       "once they are removed; written as is");
              break;
            }
            beam = beam_ratio * beam;
            cutoff = best[u] + (double)(float)beam;  // PruneLattice takes the beam as a BaseFloat
            const int64_t prev_arcs = num_arcs;
            const int32_t prev_states = num_states;
            count();
            KLAT_VLOG(1, "Lattice " << keys[u] << " pruned with beam = " << beam << " (" << n_try << " trial):  pruned #states from "
                                    << prev_states << " to " << num_states << " and #arcs from " << prev_arcs << " to " << num_arcs);
          }
          std::vector<std::vector<uint8_t>> keep((size_t)S);
          for (int32_t s = 0; s < S; ++s) {
            CompactLatticeState& st = clat.State(s);
            keep[(size_t)s].resize(st.arcs.size());
            for (size_t i = 0; i < st.arcs.size(); ++i) keep[(size_t)s][i] = excess[(size_t)s][i] <= cutoff && excess[(size_t)s][i] < inf;
            if (st.IsFinal() && final_excess[(size_t)s] > cutoff) {
              st.final_graph = st.final_acoustic = std::numeric_limits<float>::infinity();
              st.final_string_len = 0;
              st.final_string_off = -1;
            }
          }
          ConnectKeeping(keep, &clat);
          num_arcs = clat.NumArcs();
          num_states = clat.NumStates();
        }
        lattice_writer.Write(keys[u], clat);
        if (original_num_states == num_states && original_num_arcs == num_arcs)
          KLAT_LOG("Lattice " << keys[u] << " was not pruned (beam = " << original_beam << ", # states = " << original_num_states
                              << ", # arcs = " << original_num_arcs << ")");
        else
          KLAT_LOG("Lattice " << keys[u] << " pruned #states from " << original_num_states << " to " << num_states
                              << " and #arcs from " << original_num_arcs << " to " << num_arcs << " (beam reduced from "
                              << original_beam << " to " << beam << ")");
      }
      keys.clear();
      lats.clear();
      batch.Clear();
    };

    for (SequentialCompactLatticeReader reader(po.GetArg(1), true /* keep the transition ids */); !reader.Done(); reader.Next()) {
      CompactLattice& clat = reader.Value();
      if (clat.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&clat);
      keys.push_back(reader.Key());
      batch.Add(clat, nullptr);
      lats.push_back(std::move(clat));
      reader.FreeCurrent();
      if ((int32_t)keys.size() >= co.batch_size || batch.NumArcs() >= co.max_batch_arcs) flush();
    }
    flush();
    lattice_writer.Close();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
