// lattice-functions.cc -- see lattice-functions.h.
#include "lattice-functions.h"

#include <mutex>

#include <algorithm>
#include <cstring>
#include <atomic>
#include <thread>
#include <map>
#include <queue>
#include <tuple>

namespace klat {

void Check(int rc, const char* what) {
  if (rc != KLAT_OK) KLAT_ERR(what << ": " << klat_last_error());
}

static bool IsTopSorted(const CompactLattice& clat) {
  for (int32_t s = 0; s < clat.NumStates(); ++s)
    for (const auto& arc : clat.State(s).arcs)
      if (arc.nextstate <= s) return false;
  return true;
}

void TopSortCompactLatticeIfNeeded(CompactLattice* clat) {
  const int32_t S = clat->NumStates();
  if (S == 0 || clat->Start() == CompactLattice::kNoStateId) return;
  if (clat->Start() == 0 && IsTopSorted(*clat)) return;
  // DFS from the start state, then from every unvisited state in numeric order; new id = rank in
  // reverse finishing order (what fst::TopSort produces).
  std::vector<uint8_t> color(S, 0);
  std::vector<int32_t> finish;
  finish.reserve(S);
  std::vector<std::pair<int32_t, size_t>> stack;
  for (int32_t pass = -1; pass < S; ++pass) {
    const int32_t root = pass < 0 ? clat->Start() : pass;
    if (color[root]) continue;
    color[root] = 1;
    stack.emplace_back(root, 0);
    while (!stack.empty()) {
      auto& top = stack.back();
      const auto& arcs = clat->State(top.first).arcs;
      if (top.second < arcs.size()) {
        const int32_t v = arcs[top.second++].nextstate;
        if (color[v] == 1) KLAT_ERR("Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles).");
        if (color[v] == 0) {
          color[v] = 1;
          stack.emplace_back(v, 0);
        }
      } else {
        color[top.first] = 2;
        finish.push_back(top.first);
        stack.pop_back();
      }
    }
  }
  std::vector<int32_t> order(S);
  for (int32_t k = 0; k < S; ++k) order[finish[S - 1 - k]] = k;
  std::vector<CompactLatticeState> ns(S);
  for (int32_t s = 0; s < S; ++s) {
    ns[order[s]] = std::move(clat->State(s));
    for (auto& arc : ns[order[s]].arcs) arc.nextstate = order[arc.nextstate];
  }
  clat->States() = std::move(ns);
  clat->SetStart(order[clat->Start()]);
}

int32_t CompactLatticeStateTimes(const CompactLattice& clat, std::vector<int32_t>* times) {
  if (!IsTopSorted(clat)) KLAT_ERR("Input lattice must be topologically sorted.");
  const int32_t S = clat.NumStates();
  times->clear();
  if (S == 0) return 0;
  if (clat.Start() != 0) KLAT_ERR("CompactLatticeStateTimes: start state must be 0");
  times->assign(S, -1);
  (*times)[0] = 0;
  int32_t utt_len = -1;
  for (int32_t s = 0; s < S; ++s) {
    const int32_t cur = (*times)[s];
    const CompactLatticeState& st = clat.State(s);
    for (const auto& arc : st.arcs) {
      if ((*times)[arc.nextstate] == -1) (*times)[arc.nextstate] = cur + arc.string_len;
      else if ((*times)[arc.nextstate] != cur + arc.string_len)
        KLAT_ERR("CompactLatticeStateTimes: inconsistent times at state " << arc.nextstate << " (lattice is not aligned)");
    }
    if (st.IsFinal()) {
      const int32_t this_len = cur + st.final_string_len;
      if (utt_len == -1) {
        utt_len = this_len;
      } else if (this_len != utt_len) {
        KLAT_WARN("Utterance does not seem to have a consistent length.");
        utt_len = std::max(utt_len, this_len);
      }
    }
  }
  if (utt_len == -1) {
    KLAT_WARN("Utterance does not have a final-state.");
    return 0;
  }
  return utt_len;
}

int32_t DisambiguateStateInputSequenceLength(const CompactLattice& ifst, CompactLattice* ofst,
                                             std::vector<int32_t>* state_input_length, bool use_input) {
  ofst->DeleteStates();
  if (state_input_length) state_input_length->clear();
  if (ifst.Start() < 0) return 0;
  typedef std::tuple<int32_t, int32_t> Pair;  // (length, old state)
  std::map<Pair, int32_t> state_map;
  std::queue<Pair> q;
  state_map[Pair(0, ifst.Start())] = -1;
  q.push(Pair(0, ifst.Start()));
  int32_t max_len = 0;
  while (!q.empty()) {
    const int32_t len = std::get<0>(q.front()), u = std::get<1>(q.front());
    q.pop();
    max_len = std::max(max_len, len);
    for (const auto& arc : ifst.State(u).arcs) {
      const int32_t nl = ((use_input ? arc.ilabel : arc.olabel) == 0) ? len : len + 1;
      if (state_map.emplace(Pair(nl, arc.nextstate), -1).second) q.push(Pair(nl, arc.nextstate));
    }
  }
  // ids = rank in (length, state) order; state_input_length follows the same order
  int32_t next_id = 0;
  for (auto& kv : state_map) {
    kv.second = next_id++;
    ofst->AddState();
    if (state_input_length) state_input_length->push_back(std::get<0>(kv.first));
  }
  ofst->SetStart(0);
  for (const auto& kv : state_map) {
    const int32_t len = std::get<0>(kv.first), u = std::get<1>(kv.first), u2 = kv.second;
    const CompactLatticeState& in = ifst.State(u);
    CompactLatticeState& out = ofst->State(u2);
    out.final_graph = in.final_graph;
    out.final_acoustic = in.final_acoustic;
    out.final_string_len = in.final_string_len;
    out.final_string_off = -1;  // ids stay with the input lattice
    out.arcs.reserve(in.arcs.size());
    for (auto arc : in.arcs) {
      const int32_t nl = ((use_input ? arc.ilabel : arc.olabel) == 0) ? len : len + 1;
      arc.nextstate = state_map[Pair(nl, arc.nextstate)];
      arc.string_off = -1;
      if (arc.nextstate <= u2) KLAT_ERR("DisambiguateStateInputSequenceLength: output would not be topologically sorted (cyclic input?)");
      out.arcs.push_back(arc);
    }
  }
  return max_len;
}

void ArcSortByOlabel(CompactLattice* clat) {
  for (auto& st : clat->States())
    std::stable_sort(st.arcs.begin(), st.arcs.end(),
                     [](const CompactLatticeArc& a, const CompactLatticeArc& b) { return a.olabel < b.olabel; });
}

void AddSequenceLengthDisambiguationSymbol(CompactLattice* clat, std::vector<int32_t>* state_input_length,
                                           int32_t dis_label) {
  if ((size_t)clat->NumStates() != state_input_length->size())
    KLAT_ERR("AddSequenceLengthDisambiguationSymbol() expects the input length of every state ("
             << state_input_length->size() << " vs. " << clat->NumStates() << ")");
  if (clat->NumStates() == 0) return;
  const int32_t num_orig = clat->NumStates();
  const int32_t max_length = *std::max_element(state_input_length->begin(), state_input_length->end());
  std::vector<int32_t> aux((size_t)max_length + 1);
  for (int32_t k = 0; k <= max_length; ++k) aux[(size_t)k] = clat->AddState();
  CompactLatticeState& last = clat->State(aux[(size_t)max_length]);
  last.final_graph = last.final_acoustic = 0.0f;  // Weight::One()
  last.final_string_len = 0;
  for (int32_t k = 0; k <= max_length; ++k) {
    state_input_length->push_back(k);
    if (k < max_length) {
      CompactLatticeArc arc;
      arc.ilabel = arc.olabel = dis_label;
      arc.graph = arc.acoustic = 0.0f;
      arc.string_len = 0;
      arc.nextstate = aux[(size_t)k + 1];
      clat->State(aux[(size_t)k]).arcs.push_back(arc);
    }
  }
  for (int32_t u = 0; u < num_orig; ++u) {
    CompactLatticeState& st = clat->State(u);
    if (!st.IsFinal()) continue;
    CompactLatticeArc arc;
    arc.ilabel = arc.olabel = 0;
    arc.graph = st.final_graph;
    arc.acoustic = st.final_acoustic;
    arc.string_len = st.final_string_len;
    arc.nextstate = aux[(size_t)(*state_input_length)[(size_t)u]];
    st.arcs.push_back(arc);
    st.final_graph = st.final_acoustic = std::numeric_limits<float>::infinity();
    st.final_string_len = 0;
  }
}

// ConvertToCost of a weight after ScaleLattice / AddWordInsPen, with the reference's rounding:
// products formed in double and stored as float, penalty added in float, sum in double.
double ScaledCost(float g, float a, bool word, float gs, float as, float pen) {
  const float inf = std::numeric_limits<float>::infinity();
  if (as != 1.0f || gs != 1.0f) {
    if (g == inf) {
      a = inf;
    } else {
      g = (float)((double)gs * (double)g);
      a = (float)((double)as * (double)a);
    }
  }
  if (pen != 0.0f && word) g = g + pen;
  return (double)g + (double)a;
}

bool PruneLattice(float beam, float gs, float as, float pen, CompactLattice* clat) {
  if (!(beam > 0.0f)) KLAT_ERR("PruneLattice: beam must be positive");
  TopSortCompactLatticeIfNeeded(clat);
  const int32_t S = clat->NumStates();
  if (S == 0 || clat->Start() == CompactLattice::kNoStateId) return false;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> cost(S, inf);  // Viterbi forward, then shared with the backward costs
  cost[clat->Start()] = 0.0;
  double best_final = inf;
  for (int32_t s = 0; s < S; ++s) {
    const CompactLatticeState& st = clat->State(s);
    const double fc = cost[s];
    for (const auto& arc : st.arcs) {
      const double nc = fc + ScaledCost(arc.graph, arc.acoustic, arc.ilabel != 0, gs, as, pen);
      if (cost[arc.nextstate] > nc) cost[arc.nextstate] = nc;
    }
    const double tf = fc + ScaledCost(st.final_graph, st.final_acoustic, false, gs, as, 0.0f);
    if (tf < best_final) best_final = tf;
  }
  const double cutoff = best_final + beam;
  std::vector<std::vector<uint8_t>> keep(S);
  for (int32_t s = S - 1; s >= 0; --s) {
    CompactLatticeState& st = clat->State(s);
    const double fc = cost[s];
    double bc = ScaledCost(st.final_graph, st.final_acoustic, false, gs, as, 0.0f);
    if (bc + fc > cutoff && bc != inf) {
      st.final_graph = st.final_acoustic = std::numeric_limits<float>::infinity();
      st.final_string_len = 0;
      st.final_string_off = -1;
    }
    keep[s].assign(st.arcs.size(), 1);
    for (size_t i = 0; i < st.arcs.size(); ++i) {
      const auto& arc = st.arcs[i];
      const double abc = ScaledCost(arc.graph, arc.acoustic, arc.ilabel != 0, gs, as, pen) + cost[arc.nextstate];
      if (abc < bc) bc = abc;
      if (fc + abc > cutoff) keep[s][i] = 0;  // the reference redirects the arc to a dead state
    }
    cost[s] = bc;
  }
  return ConnectKeeping(keep, clat);
}

bool ConnectKeeping(const std::vector<std::vector<uint8_t>>& keep, CompactLattice* clat) {
  const int32_t S = clat->NumStates();
  // Connect(): keep states that are accessible from the start and co-accessible to a final state
  std::vector<uint8_t> acc(S, 0), coacc(S, 0);
  acc[clat->Start()] = 1;
  for (int32_t s = 0; s < S; ++s) {
    if (!acc[s]) continue;
    const auto& arcs = clat->State(s).arcs;
    for (size_t i = 0; i < arcs.size(); ++i)
      if (keep[s][i]) acc[arcs[i].nextstate] = 1;
  }
  for (int32_t s = S - 1; s >= 0; --s) {
    const CompactLatticeState& st = clat->State(s);
    bool c = st.IsFinal();
    for (size_t i = 0; i < st.arcs.size() && !c; ++i)
      if (keep[s][i] && coacc[st.arcs[i].nextstate]) c = true;
    coacc[s] = c;
  }
  std::vector<int32_t> newid(S, -1);
  int32_t n = 0;
  for (int32_t s = 0; s < S; ++s)
    if (acc[s] && coacc[s]) newid[s] = n++;
  if (n == 0 || newid[clat->Start()] < 0) {
    clat->DeleteStates();
    return false;
  }
  std::vector<CompactLatticeState> ns(n);
  for (int32_t s = 0; s < S; ++s) {
    if (newid[s] < 0) continue;
    CompactLatticeState& src = clat->State(s);
    CompactLatticeState& dst = ns[newid[s]];
    dst.final_graph = src.final_graph;
    dst.final_acoustic = src.final_acoustic;
    dst.final_string_len = src.final_string_len;
    dst.final_string_off = src.final_string_off;
    for (size_t i = 0; i < src.arcs.size(); ++i) {
      if (!keep[s][i] || newid[src.arcs[i].nextstate] < 0) continue;
      CompactLatticeArc arc = src.arcs[i];
      arc.nextstate = newid[arc.nextstate];
      dst.arcs.push_back(arc);
    }
  }
  const int32_t new_start = newid[clat->Start()];
  clat->States() = std::move(ns);
  clat->SetStart(new_start);
  return true;
}

void PruneLatticesOnDevice(Context& ctx, float beam, float gs, float as, float pen, std::vector<CompactLattice>* lats) {
  if (!(beam > 0.0f)) KLAT_ERR("PruneLattice: beam must be positive");
  LatticeBatch batch;
  for (auto& clat : *lats) {
    if (clat.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&clat);
    batch.Add(clat, nullptr);
  }
  if (batch.NumArcs() == 0 && batch.NumStates() == 0) return;
  DeviceBatch db(ctx, batch, nullptr);
  klat_scales sc = {gs, as, pen};
  Check(klat_viterbi(db.get(), &sc), "klat_viterbi");
  std::vector<uint8_t> keep_arc((size_t)std::max<int64_t>(batch.NumArcs(), 1)), keep_final((size_t)std::max<int64_t>(batch.NumStates(), 1));
  Check(klat_prune_mask(db.get(), beam, keep_arc.data(), keep_final.data(), nullptr), "klat_prune_mask");
  for (size_t u = 0; u < lats->size(); ++u) {
    CompactLattice& clat = (*lats)[u];
    const int32_t S = clat.NumStates();
    if (S == 0) continue;
    const int64_t s0 = batch.StateOff()[u];
    int64_t a = batch.ArcOff()[u];
    std::vector<std::vector<uint8_t>> keep((size_t)S);
    for (int32_t s = 0; s < S; ++s) {
      CompactLatticeState& st = clat.State(s);
      if (st.IsFinal() && !keep_final[(size_t)(s0 + s)]) {
        st.final_graph = st.final_acoustic = std::numeric_limits<float>::infinity();
        st.final_string_len = 0;
        st.final_string_off = -1;
      }
      keep[(size_t)s].assign(keep_arc.begin() + a, keep_arc.begin() + a + (int64_t)st.arcs.size());
      a += (int64_t)st.arcs.size();
    }
    ConnectKeeping(keep, &clat);
  }
}

// ---- LatticeBatch ----
void LatticeBatch::Clear() {
  state_off_.assign(1, 0);
  arc_off_.assign(1, 0);
  src_.clear(), dst_.clear(), ilabel_.clear(), olabel_.clear(), state_time_.clear();
  graph_.clear(), acoustic_.clear(), final_graph_.clear(), final_acoustic_.clear();
  has_times_ = true;
}

void LatticeBatch::Add(const CompactLattice& clat, const std::vector<int32_t>* times) {
  const int32_t S = clat.Start() == CompactLattice::kNoStateId ? 0 : clat.NumStates();
  if (S > 0 && clat.Start() != 0) KLAT_ERR("LatticeBatch::Add: lattice must be top-sorted with start state 0");
  if (!times) has_times_ = false;
  for (int32_t s = 0; s < S; ++s) {
    const CompactLatticeState& st = clat.State(s);
    final_graph_.push_back(st.final_graph);
    final_acoustic_.push_back(st.final_acoustic);
    if (times) state_time_.push_back((*times)[s]);
    for (const auto& arc : st.arcs) {
      src_.push_back(s);
      dst_.push_back(arc.nextstate);
      ilabel_.push_back(arc.ilabel);
      olabel_.push_back(arc.olabel);
      graph_.push_back(arc.graph);
      acoustic_.push_back(arc.acoustic);
    }
  }
  state_off_.push_back((int64_t)final_graph_.size());
  arc_off_.push_back((int64_t)src_.size());
}

void LatticeBatch::AddMany(const std::vector<const CompactLattice*>& lats, const std::vector<const std::vector<int32_t>*>& times, int threads) {
  const size_t n = lats.size();
  std::vector<int64_t> so(n + 1), ao(n + 1);
  so[0] = state_off_.back(), ao[0] = arc_off_.back();
  bool all_times = true;
  for (size_t i = 0; i < n; ++i) {
    const CompactLattice& clat = *lats[i];
    const int32_t S = clat.Start() == CompactLattice::kNoStateId ? 0 : clat.NumStates();
    if (S > 0 && clat.Start() != 0) KLAT_ERR("LatticeBatch::Add: lattice must be top-sorted with start state 0");
    so[i + 1] = so[i] + S;
    ao[i + 1] = ao[i] + (S > 0 ? clat.NumArcs() : 0);
    if (!times[i]) all_times = false;
  }
  if (!all_times) has_times_ = false;
  final_graph_.resize((size_t)so[n]);
  final_acoustic_.resize((size_t)so[n]);
  if (has_times_) state_time_.resize((size_t)so[n]);
  for (PodVec<int32_t>* v : {&src_, &dst_, &ilabel_, &olabel_}) v->resize((size_t)ao[n]);
  graph_.resize((size_t)ao[n]);
  acoustic_.resize((size_t)ao[n]);
  std::atomic<size_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) break;
      const CompactLattice& clat = *lats[i];
      const int32_t S = (int32_t)(so[i + 1] - so[i]);
      int64_t a = ao[i];
      for (int32_t s = 0; s < S; ++s) {
        const CompactLatticeState& st = clat.State(s);
        final_graph_[(size_t)(so[i] + s)] = st.final_graph;
        final_acoustic_[(size_t)(so[i] + s)] = st.final_acoustic;
        if (has_times_) state_time_[(size_t)(so[i] + s)] = (*times[i])[(size_t)s];
        for (const auto& arc : st.arcs) {
          src_[(size_t)a] = s;
          dst_[(size_t)a] = arc.nextstate;
          ilabel_[(size_t)a] = arc.ilabel;
          olabel_[(size_t)a] = arc.olabel;
          graph_[(size_t)a] = arc.graph;
          acoustic_[(size_t)a] = arc.acoustic;
          ++a;
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < std::min<int>(threads, (int)n); ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  for (size_t i = 0; i < n; ++i) state_off_.push_back(so[i + 1]), arc_off_.push_back(ao[i + 1]);
}

void LatticeBatch::AddManyFlat(const std::vector<FlatLattice>& lats, const std::vector<std::vector<int32_t>>& times, int threads) {
  const size_t n = lats.size();
  std::vector<int64_t> so(n + 1), ao(n + 1);
  so[0] = state_off_.back(), ao[0] = arc_off_.back();
  for (size_t i = 0; i < n; ++i) {
    if (!IsTopSortedFromZero(lats[i])) KLAT_ERR("LatticeBatch::Add: lattice must be top-sorted with start state 0");
    so[i + 1] = so[i] + lats[i].NumStates();
    ao[i + 1] = ao[i] + lats[i].NumArcs();
  }
  final_graph_.resize((size_t)so[n]);
  final_acoustic_.resize((size_t)so[n]);
  if (has_times_) state_time_.resize((size_t)so[n]);
  for (PodVec<int32_t>* v : {&src_, &dst_, &ilabel_, &olabel_}) v->resize((size_t)ao[n]);
  graph_.resize((size_t)ao[n]);
  acoustic_.resize((size_t)ao[n]);
  std::atomic<size_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) break;
      const FlatLattice& lat = lats[i];
      const int32_t S = lat.NumStates();
      const size_t A = (size_t)lat.NumArcs();
      if (S > 0) {
        std::memcpy(final_graph_.data() + so[i], lat.final_graph.data(), sizeof(float) * (size_t)S);
        std::memcpy(final_acoustic_.data() + so[i], lat.final_acoustic.data(), sizeof(float) * (size_t)S);
        if (has_times_) std::memcpy(state_time_.data() + so[i], times[i].data(), sizeof(int32_t) * (size_t)S);
      }
      if (A > 0) {
        std::memcpy(dst_.data() + ao[i], lat.dst.data(), sizeof(int32_t) * A);
        std::memcpy(ilabel_.data() + ao[i], lat.ilabel.data(), sizeof(int32_t) * A);
        std::memcpy(olabel_.data() + ao[i], lat.olabel.data(), sizeof(int32_t) * A);
        std::memcpy(graph_.data() + ao[i], lat.graph.data(), sizeof(float) * A);
        std::memcpy(acoustic_.data() + ao[i], lat.acoustic.data(), sizeof(float) * A);
        for (int32_t s = 0; s < S; ++s)
          for (int32_t k = lat.row[(size_t)s]; k < lat.row[(size_t)s + 1]; ++k) src_[(size_t)(ao[i] + k)] = s;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < std::min<int>(threads, (int)n); ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  for (size_t i = 0; i < n; ++i) state_off_.push_back(so[i + 1]), arc_off_.push_back(ao[i + 1]);
}

klat_batch_desc LatticeBatch::Desc() const {
  klat_batch_desc d;
  d.num_lattices = NumLattices();
  d.state_off = state_off_.data();
  d.arc_off = arc_off_.data();
  d.src = src_.data();
  d.dst = dst_.data();
  d.ilabel = ilabel_.data();
  d.olabel = olabel_.data();
  d.graph = graph_.data();
  d.acoustic = acoustic_.data();
  d.final_graph = final_graph_.data();
  d.final_acoustic = final_acoustic_.data();
  d.state_time = (has_times_ && (int64_t)state_time_.size() == NumStates()) ? state_time_.data() : nullptr;
  return d;
}

Context::Context(int device) { Check(klat_ctx_create(device, &ctx_), "klat_ctx_create"); }
Context::~Context() { klat_ctx_destroy(ctx_); }

DeviceBatch::DeviceBatch(Context& ctx, const LatticeBatch& batch, const klat_pack_options* opt) {
  const klat_batch_desc d = batch.Desc();
  Check(klat_batch_create(ctx.get(), &d, opt, &b_), "klat_batch_create");
}
DeviceBatch::~DeviceBatch() { klat_batch_destroy(b_); }
void DeviceBatch::ForwardBackward(float graph_scale, float acoustic_scale, float insertion_penalty) {
  const klat_scales s = {graph_scale, acoustic_scale, insertion_penalty};
  // none of the tools reads per-arc posteriors: the accumulators form them from alpha / beta
  Check(klat_forward_backward_ex(b_, &s, KLAT_FB_NO_ARC_POSTERIORS), "klat_forward_backward");
}

namespace {
// The GPU contexts behind ComputeLatticeAlphasAndBetas: every calling thread leases one for its
// lifetime (one HIP stream per thread: the B2 contract of kwsbin2/lattice-word-index-segment.cc:
// 181,264-272, where each TaskSequencer worker owns its lattice) and hands it back when the thread
// ends, so a pool of short-lived workers creates as many contexts as run at once, not one per
// thread.  The pool itself is never destroyed: a context must not be torn down from a static or
// thread_local destructor at process exit, when the HIP runtime may already be gone.
struct ContextPool {
  std::mutex mu;
  std::vector<Context*> idle;
  int device = 0;
  Context* Acquire() {
    {
      std::lock_guard<std::mutex> lock(mu);
      if (!idle.empty()) {
        Context* c = idle.back();
        idle.pop_back();
        return c;
      }
    }
    return new Context(device);
  }
  void Release(Context* c) {
    std::lock_guard<std::mutex> lock(mu);
    idle.push_back(c);
  }
};
ContextPool& Pool() {
  static ContextPool* pool = new ContextPool();  // intentionally leaked, see above
  return *pool;
}
struct ContextLease {
  Context* ctx;
  ContextLease() : ctx(Pool().Acquire()) {}
  ~ContextLease() { Pool().Release(ctx); }
};
}  // namespace

void SetDefaultDevice(int device) {
  std::lock_guard<std::mutex> lock(Pool().mu);
  Pool().device = device;
}

double ComputeLatticeAlphasAndBetas(const CompactLattice& clat, bool viterbi, std::vector<double>* alpha,
                                    std::vector<double>* beta) {
  if (clat.NumStates() == 0 || clat.Start() != 0) KLAT_ERR("ComputeLatticeAlphasAndBetas: lattice must be top-sorted with start state 0");
  static thread_local ContextLease lease;  // one HIP stream per calling thread (re-entrant, B2 contract)
  Context& ctx = *lease.ctx;
  LatticeBatch lb;
  lb.Add(clat, nullptr);
  DeviceBatch db(ctx, lb);
  if (viterbi) {
    // LogAddOrMax(viterbi = true, ...) of upstream's routine: alpha / beta are the best partial paths'
    // log-likelihoods = minus the Viterbi costs of the sweep's (min,+) twin (klat_viterbi)
    const klat_scales unit = {1.0f, 1.0f, 0.0f};
    Check(klat_viterbi(db.get(), &unit), "klat_viterbi");
    alpha->assign(clat.NumStates(), 0.0);
    beta->assign(clat.NumStates(), 0.0);
    double best = 0.0;
    Check(klat_get_viterbi(db.get(), alpha->data(), beta->data(), &best), "klat_get_viterbi");
    for (double& v : *alpha) v = -v;  // (+inf cost: unreachable -> log-zero)
    for (double& v : *beta) v = -v;
    return -best;  // tot_forward = tot_backward = the best complete path
  }
  db.ForwardBackward(1.0f, 1.0f, 0.0f);
  alpha->assign(clat.NumStates(), 0.0);
  beta->assign(clat.NumStates(), 0.0);
  double total = 0.0, tf = 0.0, tb = 0.0;
  Check(klat_get_alpha_beta(db.get(), alpha->data(), beta->data()), "klat_get_alpha_beta");
  Check(klat_get_totals(db.get(), &total, &tf, &tb), "klat_get_totals");
  if (std::abs(tf - tb) > 1e-8 * std::max(std::abs(tf), std::abs(tb)))
    KLAT_WARN("Total forward probability over lattice = " << tf << ", while total backward probability = " << tb);
  return total;
}

}  // namespace klat
