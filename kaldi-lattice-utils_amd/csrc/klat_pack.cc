// klat_pack.cc -- builds the forward / backward lockstep arc streams from the CSR batch.
//
// This is the "flatten" stage: it runs once per batch on the host, outside the timed region,
// like the preprocessing the reference's own timers exclude
// (kwsbin2/lattice-word-index-segment.cc:140-141 "Timer does not include preprocessing time").
//
// Forward stream of one lattice: for every position p = 0..S (p == S is a virtual super-final
// state whose value is tot_forward), the terms (source position, cost) that feed it:
//   p == 0        : the constant-one term (alpha[start] = 0)
//   0 < p < S     : one term per in-arc (src -> p)
//   p == S        : one term per final state s, cost = final weight
// Backward stream: position q stands for state s = S-1-q; terms are the out-arcs of s (source
// = position of dst) plus, if s is final, a constant-one term carrying the final cost.
// A position with no term gets one NIL term so that every state is finalised exactly once.
// Terms of a position are dealt to the lattice's G lanes round by round; the last round of a
// position carries LAST on all G lanes, the first round FIRST.  Within a position the term whose
// source was finished most recently comes last (it is the one on the dependency chain).
#include "klat_pack.h"

#include <algorithm>
#include <limits>
#include <atomic>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

namespace klat {

namespace {

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <class F>
void parallel_for(int64_t n, int threads, F f) {
  if (threads <= 1 || n < 2) {
    for (int64_t i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int64_t> next(0);
  auto worker = [&]() {
    for (;;) {
      int64_t i = next.fetch_add(1);
      if (i >= n) break;
      f(i);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
}

inline bool is_final(const klat_batch_desc& d, int64_t gs) {
  return !(d.final_graph[gs] == INFINITY && d.final_acoustic[gs] == INFINITY);
}

}  // namespace

int choose_ring_slots(int G, int64_t max_rel) {
  const int LPW = 64 / G;
  // 32 slots: the sweep then stages two chunks per barrier and never has to wait for its own
  // stores (klat_sweep.hip).  The ring grows with the longest arc span only while the workgroup
  // stays within the LDS that lets five of them share a CU (1024 ring cells: W = 64 at G = 4):
  // longer arcs are FAR terms, whose sources the producer prefetches one period ahead, and that
  // costs less than a CU that holds three workgroups instead of five (tools/span_scan.py: 30 % of
  // the arcs stretched up to 1000 states, W = 32 / 64 / 128: 0.84 / 0.83 / 1.26 ms).
  int W = 32;
  while (W <= max_rel && 2 * W * LPW <= 1024) W *= 2;
  return W;
}

// ---- layout planner -------------------------------------------------------------------------
//
// Every lattice gets its own number of lanes G.  Its terms are dealt state by state, so a
// lattice of S states with in-degree d takes sum ceil(d / G) >= S rounds, each of which depends
// on the one before: G ~ fanout makes the chain one round per state, a larger G only pads.  The
// starting point is therefore G = fanout rounded up to a power of two (at least 2 for a branching
// lattice: with one lane the chain is one round per ARC and the sweep is slower even when the
// chip is full many times over -- tools/layout_scan.py: 140 000 lattices x 150 states, G = 1:
// 92 G arcs/s, G = 2: 112, G = 4: 110), halved while the batch would need more than 8192
// workgroups.  A skewed batch is then bounded by its longest lattices, not by
// the chip: the planner lowers the round budget R step by step, gives every lattice the smallest
// G that meets R, and keeps the R with the smallest modelled time
//   max( longest chain x t_lone(G),  job-rounds / resident workgroups x t_lone(G) x f(occupancy) )
//   + launch overhead per layout class
// with t_lone and f measured on MI355X (tools/occupancy_scan.py, tools/config_scan.py).
namespace {

const double kLoneCycles[kNumLaneChoices] = {300, 300, 308, 324, 359, 400, 450};  // cycles per round, one workgroup alone

double occupancy_factor(double jobs) {  // slow-down of a round against workgroups per CU
  static const double f[6] = {1.0, 1.0, 1.15, 1.47, 1.49, 1.78};
  const double w = std::min(5.0, std::max(0.0, jobs / 256.0));
  const int i = (int)w;
  return i >= 5 ? f[5] : f[i] + (f[i + 1] - f[i]) * (w - i);
}

struct PlanCost {
  double cycles, jobs;
};

PlanCost plan_cost(int64_t U, const LatticeRounds* lr, const std::vector<int8_t>& g) {
  double chain = 0, work[kNumLaneChoices] = {0}, count[kNumLaneChoices] = {0};
  for (int64_t u = 0; u < U; ++u) {
    if (g[(size_t)u] < 0) continue;
    const int k = g[(size_t)u];
    chain = std::max(chain, (double)std::max(lr[u].fwd[k], lr[u].bwd[k]) * kLoneCycles[k]);
    work[k] += ((double)lr[u].fwd[k] + (double)lr[u].bwd[k]) * (double)(1 << k) / 64.0 * kLoneCycles[k];
    count[k] += 1;
  }
  double jobs = 0, cyc = 0, launches = 0;
  for (int k = 0; k < kNumLaneChoices; ++k) {
    jobs += 2.0 * std::ceil(count[k] * (double)(1 << k) / 64.0);
    cyc += work[k];
    launches += count[k] > 0 ? 1 : 0;
  }
  if (jobs <= 0) return PlanCost{0, 0};
  const double kLaunchCycles = 30000;  // one more layout class = one more pair of launches
  return PlanCost{std::max(chain, cyc / std::min(jobs, 1280.0) * occupancy_factor(jobs)) + kLaunchCycles * launches, jobs};
}

}  // namespace

void predict_lanes(int64_t U, const int64_t* state_off, const int64_t* arc_off, const klat_pack_options& opt, std::vector<int8_t>* lat_g,
                   const uint8_t* skip) {
  std::vector<int8_t>& g = *lat_g;
  g.assign((size_t)U, -1);
  if (opt.lanes_per_lattice != 0) {
    int k = 0;
    while ((1 << k) < opt.lanes_per_lattice) ++k;
    for (int64_t u = 0; u < U; ++u)
      if (state_off[u + 1] > state_off[u] && !(skip && skip[u])) g[(size_t)u] = (int8_t)k;
    return;
  }
  double jobs = 0;
  for (int64_t u = 0; u < U; ++u) {
    const int64_t S = state_off[u + 1] - state_off[u], A = arc_off[u + 1] - arc_off[u];
    if (S <= 0 || (skip && skip[u])) continue;
    const double fan = (double)A / (double)S;
    int k = 0;
    while ((1 << k) < fan && k < kNumLaneChoices - 1) ++k;
    g[(size_t)u] = (int8_t)k;
    jobs += 2.0 * (double)(1 << k) / 64.0;
  }
  while (jobs > 8192) {
    double nj = 0;
    bool changed = false;
    for (int64_t u = 0; u < U; ++u) {
      if (g[(size_t)u] > 1) g[(size_t)u]--, changed = true;
      if (g[(size_t)u] >= 0) nj += 2.0 * (double)(1 << g[(size_t)u]) / 64.0;
    }
    jobs = nj;
    if (!changed) break;
  }
}

void plan_layout(int64_t U, const int64_t* state_off, const int64_t* arc_off, const LatticeRounds* lr,
                 const klat_pack_options& opt, std::vector<int8_t>* lat_g, const uint8_t* skip) {
  std::vector<int8_t>& g = *lat_g;
  predict_lanes(U, state_off, arc_off, opt, lat_g, skip);
  if (opt.lanes_per_lattice != 0) return;
  // round budgets below the longest chain
  const std::vector<int8_t> base(g);
  std::vector<int8_t> best(g), cur(g);
  PlanCost best_cost = plan_cost(U, lr, base);
  const double base_cycles = best_cost.cycles;
  int64_t longest = 0, floor_rounds = 0;
  for (int64_t u = 0; u < U; ++u) {
    if (base[(size_t)u] < 0) continue;
    longest = std::max<int64_t>(longest, std::max(lr[u].fwd[base[(size_t)u]], lr[u].bwd[base[(size_t)u]]));
    floor_rounds = std::max<int64_t>(floor_rounds, std::max(lr[u].fwd[kNumLaneChoices - 1], lr[u].bwd[kNumLaneChoices - 1]));
  }
  for (double R = 0.92 * (double)longest; R >= 0.999 * (double)floor_rounds && R > 0; R *= 0.92) {
    for (int64_t u = 0; u < U; ++u) {
      int k = base[(size_t)u];
      if (k < 0) continue;
      while (k < kNumLaneChoices - 1 && std::max(lr[u].fwd[k], lr[u].bwd[k]) > R &&
             std::max(lr[u].fwd[k + 1], lr[u].bwd[k + 1]) < std::max(lr[u].fwd[k], lr[u].bwd[k]))
        ++k;
      cur[(size_t)u] = (int8_t)k;
    }
    const PlanCost c = plan_cost(U, lr, cur);
    if (c.jobs > 8192) break;
    if (c.cycles < best_cost.cycles) best_cost = c, best = cur;
  }
  if (best_cost.cycles < 0.95 * base_cycles) g = best;  // otherwise not worth the extra lanes
}

int base_lane_choice(int64_t S, int64_t A) {
  const double fan = S > 0 ? (double)A / (double)S : 0.0;
  int k = 0;
  while ((1 << k) < fan && k < kNumLaneChoices - 1) ++k;
  return k;
}

double plan_cycles(int64_t U, const int64_t* state_off, const int64_t* arc_off, const LatticeRounds* lr,
                   const klat_pack_options& opt, const uint8_t* skip) {
  std::vector<int8_t> g;
  plan_layout(U, state_off, arc_off, lr, opt, &g, skip);
  return plan_cost(U, lr, g).cycles;
}

// keys in ascending order of their upper 32 bits, equal ones as they stood (the lower bits here count upwards: the whole
// key ascending); three stable passes over the digits that differ (std::sort: 0.2 ms for 10 000 keys, this: a tenth)
static void radix_sort_high32(std::vector<uint64_t>* keys, std::vector<uint64_t>* tmp) {
  const size_t n = keys->size();
  if (n < 64) {
    std::sort(keys->begin(), keys->end());
    return;
  }
  tmp->resize(n);
  uint64_t lo = ~0ull, hi = 0;
  for (uint64_t k : *keys) lo = std::min(lo, k >> 32), hi = std::max(hi, k >> 32);
  std::vector<uint64_t>*from = keys, *to = tmp;
  for (int shift = 0; shift < 33 && ((hi - lo) >> shift) != 0; shift += 11) {
    size_t count[2049] = {0};
    for (uint64_t k : *from) ++count[(((k >> 32) - lo) >> shift & 2047) + 1];
    for (int i = 0; i < 2048; ++i) count[i + 1] += count[i];
    for (uint64_t k : *from) (*to)[count[((k >> 32) - lo) >> shift & 2047]++] = k;
    std::swap(from, to);
  }
  if (from != keys) keys->swap(*tmp);
}

int64_t slice_jobs(const int64_t* state_off, const LatticeRounds* lr, const std::vector<int64_t>& members,
                   const std::vector<int64_t>& rf, const std::vector<int64_t>& rb, int G, int W, int64_t chunk_base,
                   PackedBatch* out, std::vector<std::vector<int64_t>>* job_lats) {
  const int LPW = 64 / G;
  const int D = kDeepPrefetchChunks;
  // ---- slicing: lattices of similar length share a wave ----
  // (sorted by rounds, longest first, equal rounds in the members' order: one 64-bit key a lattice -- a stable sort
  // through a comparison that looks the rounds up took 0.3 ms of klat_batch_create on 10 000 lattices)
  struct ProtoJob {
    int dir;
    int64_t rounds;
    std::vector<int64_t> lat;
  };
  std::vector<ProtoJob> proto;
  // (ONE order for both directions -- by the rounds of the two streams together -- so that a group of lattices has a
  // forward and a backward job over the same lattices: the stream packer's fill pass then writes a group's two jobs
  // side by side and reads the group's arcs from HBM once, the second time from the Infinity Cache.  Sorted by each
  // direction's own rounds the jobs were a little tighter where a lattice's two streams differ much in length.)
  std::vector<uint64_t> keys(members.size()), keys_tmp;
  for (size_t i = 0; i < members.size(); ++i) keys[i] = ((uint64_t)(0x7FFFFFFF - (rf[members[i]] + rb[members[i]]) / 2) << 32) | (uint64_t)i;
  radix_sort_high32(&keys, &keys_tmp);
  static const bool own_order = std::getenv("KLAT_JOB_ORDER_OWN") != nullptr;  // (A/B: every direction sorted by its own rounds, as until round 5)
  for (int dir = 0; dir < 2; ++dir) {
    const std::vector<int64_t>& r = dir ? rb : rf;
    if (own_order) {
      for (size_t i = 0; i < members.size(); ++i) keys[i] = ((uint64_t)(0x7FFFFFFF - r[members[i]]) << 32) | (uint64_t)i;
      radix_sort_high32(&keys, &keys_tmp);
    }
    for (size_t i = 0; i < keys.size(); i += LPW) {
      ProtoJob pj;
      pj.dir = dir;
      pj.rounds = 0;
      pj.lat.reserve(LPW);
      for (size_t j = i; j < std::min(keys.size(), i + (size_t)LPW); ++j) {
        const int64_t u = members[(size_t)(keys[j] & 0xFFFFFFFFu)];
        pj.lat.push_back(u);
        pj.rounds = std::max(pj.rounds, r[u]);
      }
      proto.push_back(std::move(pj));
    }
  }
  std::stable_sort(proto.begin(), proto.end(),
                   [](const ProtoJob& a, const ProtoJob& b) { return a.rounds > b.rounds; });

  PackedBatch& P = *out;
  const size_t j0 = P.jobs.size();
  const size_t l0 = P.lanes.size();
  P.jobs.resize(j0 + proto.size());
  P.lanes.resize(l0 + proto.size() * LPW, LaneInfo{-1, 0, -1});
  job_lats->resize(j0 + proto.size());
  int64_t chunk_total = chunk_base;
  for (size_t j = 0; j < proto.size(); ++j) {
    int64_t nch = ceil_div(proto[j].rounds, kChunkRounds);
    nch = ceil_div(nch, D) * D;
    SweepJob& job = P.jobs[j0 + j];
    // lanes per lattice that ever carry a term: only their units are stored
    int A = 1;
    for (int64_t u : proto[j].lat) A = std::max(A, (int)((proto[j].dir ? (uint32_t)lr[u].max_terms >> 16 : (uint32_t)lr[u].max_terms & 0xFFFFu)));
    A = std::min(A, G);
    job.chunk_off = chunk_total;  // in units of 16 B
    job.nchunks = (int32_t)nch;
    job.dir = proto[j].dir;
    job.lane_tab = (int32_t)(l0 + j * LPW);
    job.tag = job_tag(0, G, W, A);
    chunk_total += nch * A * LPW;
    (proto[j].dir ? P.bwd_slots : P.fwd_slots) += nch * kChunkRounds * A * LPW;
    P.max_rounds = std::max<int32_t>(P.max_rounds, (int32_t)(nch * kChunkRounds));
    for (size_t k = 0; k < proto[j].lat.size(); ++k) {
      const int64_t u = proto[j].lat[k];
      LaneInfo& li = P.lanes[l0 + j * LPW + k];
      li.state_base = state_off[u];
      li.num_states = (int32_t)(state_off[u + 1] - state_off[u]);
      li.lattice = (int32_t)u;
    }
    (*job_lats)[j0 + j] = std::move(proto[j].lat);
  }
  P.classes.push_back(LayoutClass{G, W, (int32_t)j0, (int32_t)(j0 + proto.size()), 0, 0});
  return chunk_total;
}

int build_jobs(int64_t U, const int64_t* state_off, const LatticeRounds* lr, const std::vector<int8_t>& lat_g,
               const klat_pack_options& opt, PackedBatch* out, std::vector<std::vector<int64_t>>* job_lats,
               int64_t* chunk_total, std::string* err, const uint8_t* on_chain) {
  PackedBatch& P = *out;
  struct Proto {
    int k, chain;
    int64_t longest, max_rel, slots;
    std::vector<int64_t> members;
  };
  std::vector<Proto> cls;
  for (int chain = 0; chain < (on_chain ? 2 : 1); ++chain)
    for (int k = 0; k < kNumLaneChoices; ++k) {
      Proto c{k, chain, 0, 1, 0, {}};
      for (int64_t u = 0; u < U; ++u)
        if (lat_g[(size_t)u] == k && (on_chain ? on_chain[u] != 0 : false) == (chain != 0)) {
          c.members.push_back(u);
          c.longest = std::max<int64_t>(c.longest, std::max(lr[u].fwd[k], lr[u].bwd[k]));
          c.max_rel = std::max<int64_t>(c.max_rel, lr[u].max_rel);
          c.slots += ((int64_t)lr[u].fwd[k] + lr[u].bwd[k]) << k;
        }
      if (!c.members.empty()) cls.push_back(std::move(c));
    }
  // launch order: the class with the longest chain first
  std::stable_sort(cls.begin(), cls.end(), [](const Proto& a, const Proto& b) { return a.longest > b.longest; });
  std::vector<int64_t> rf((size_t)U, 0), rb((size_t)U, 0);
  int64_t chunks = 0, most_slots = -1;
  for (const Proto& c : cls) {
    const int G = 1 << c.k, LPW = 64 / G;
    // The ring only has to cover arcs; final terms (rel = S - s) may be far without harm, so the
    // automatic W follows the arcs' span, which the count pass folded into max_rel with the finals.
    const int W = opt.ring_slots ? opt.ring_slots : choose_ring_slots(G, c.max_rel);
    if ((int64_t)W * LPW > kMaxRingCells) {
      *err = "ring_slots * (64 / lanes_per_lattice) must not exceed 4032 ring cells per workgroup";
      return KLAT_ERR_INVALID_ARG;
    }
    for (int64_t u : c.members) rf[(size_t)u] = lr[u].fwd[c.k], rb[(size_t)u] = lr[u].bwd[c.k];
    chunks = slice_jobs(state_off, lr, c.members, rf, rb, G, W, chunks, &P, job_lats);
    P.classes.back().far = c.max_rel >= W ? 1 : 0;
    P.classes.back().chain = c.chain;
    if (c.slots > most_slots) most_slots = c.slots, P.G = G, P.W = W;  // what klat_batch_info reports
  }
  if (cls.empty()) {
    P.G = opt.lanes_per_lattice ? opt.lanes_per_lattice : 1;
    P.W = opt.ring_slots ? opt.ring_slots : choose_ring_slots(P.G, 1);
  }
  // wave priorities: in a skewed batch the long chains outrank the waves they share a SIMD with
  // (the jobs the log-semiring sweep runs side by side: those outside the segment path's classes)
  std::vector<uint8_t> in_chain(P.jobs.size(), 0);
  for (const LayoutClass& lc : P.classes)
    if (lc.chain)
      for (int32_t j = lc.job_begin; j < lc.job_end; ++j) in_chain[(size_t)j] = 1;
  int64_t sum_rounds = 0, longest = 0, count = 0;
  for (size_t j = 0; j < P.jobs.size(); ++j)
    if (!in_chain[j]) sum_rounds += P.jobs[j].nchunks, longest = std::max<int64_t>(longest, P.jobs[j].nchunks), ++count;
  if (count > 0 && longest * count > 2 * sum_rounds)
    for (size_t j = 0; j < P.jobs.size(); ++j)
      if (!in_chain[j]) P.jobs[j].tag |= (int32_t)std::min<int64_t>(3, 4 * (int64_t)P.jobs[j].nchunks / (longest + 1));
  *chunk_total = chunks;
  return KLAT_OK;
}

void append_post_items(std::vector<PostItem>* items, int64_t a0, int64_t a1, int64_t u) {
  int32_t first = kPostItemFirst;
  for (int64_t a = a0; a < a1 || first; a += kPostItemArcs) {
    items->push_back(PostItem{a, (int32_t)std::max<int64_t>(0, std::min<int64_t>(kPostItemArcs, a1 - a)) | first, (int32_t)u});
    first = 0;
  }
}

int pack_batch(const klat_batch_desc& d, const klat_pack_options& opt, PackedBatch* out,
               std::string* err) {
  const int64_t U = d.num_lattices;
  if (U < 0 || (U > 0 && (!d.state_off || !d.arc_off))) {
    *err = "klat_batch_desc: null offsets";
    return KLAT_ERR_INVALID_ARG;
  }
  const int64_t S_tot = U ? d.state_off[U] : 0, A_tot = U ? d.arc_off[U] : 0;
  if (A_tot > 0 && (!d.src || !d.dst || !d.graph || !d.acoustic || !d.ilabel)) {
    *err = "klat_batch_desc: null arc arrays";
    return KLAT_ERR_INVALID_ARG;
  }
  if (S_tot > 0 && (!d.final_graph || !d.final_acoustic)) {
    *err = "klat_batch_desc: null final-weight arrays";
    return KLAT_ERR_INVALID_ARG;
  }
  int threads = opt.host_threads > 0 ? opt.host_threads : (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;

  const int G_opt = opt.lanes_per_lattice;
  if (G_opt < 0 || G_opt > 64 || (G_opt & (G_opt - 1))) {
    *err = "lanes_per_lattice must be a power of two in [1,64]";
    return KLAT_ERR_INVALID_ARG;
  }
  if (opt.ring_slots != 0 &&
      (opt.ring_slots < 8 || opt.ring_slots > 1024 || (opt.ring_slots & (opt.ring_slots - 1)))) {
    *err = "ring_slots must be a power of two in [8,1024]";
    return KLAT_ERR_INVALID_ARG;
  }

  // ---- pass 1: validate, count the rounds of every lattice and direction for every G ----
  std::vector<LatticeRounds> lr((size_t)U);
  std::atomic<int> bad(0);
  std::atomic<int64_t> bad_lat(-1);
  parallel_for(U, threads, [&](int64_t u) {
    const int64_t s0 = d.state_off[u], a0 = d.arc_off[u];
    const int64_t S = d.state_off[u + 1] - s0, A = d.arc_off[u + 1] - a0;
    std::memset(&lr[(size_t)u], 0, sizeof(LatticeRounds));
    if (S < 0 || A < 0 || S >= (int64_t)kRelMask || (S == 0 && A > 0)) {
      bad = KLAT_ERR_INVALID_ARG;
      bad_lat = u;
      return;
    }
    if (S == 0) return;
    std::vector<int32_t> in_cnt(S, 0), out_cnt(S, 0);
    int64_t mr = 1;
    int32_t prev_src = 0;
    for (int64_t i = 0; i < A; ++i) {
      const int32_t s = d.src[a0 + i], t = d.dst[a0 + i];
      if (s < 0 || s >= S || t <= s || t >= S || s < prev_src) {
        bad = (s < prev_src || s < 0 || s >= S) ? KLAT_ERR_INVALID_ARG : KLAT_ERR_BAD_TOPOLOGY;
        bad_lat = u;
        return;
      }
      prev_src = s;
      in_cnt[t]++;
      out_cnt[s]++;
      mr = std::max<int64_t>(mr, t - s);
    }
    lr[(size_t)u].max_span = (int32_t)mr;
    int64_t nfinal = 0, f[kNumLaneChoices] = {0}, b[kNumLaneChoices] = {0};
    int64_t mtf = 1, mtb = 1;
    for (int64_t s = 0; s < S; ++s) {
      const bool fin = is_final(d, s0 + s);
      if (fin) {
        nfinal++;
        mr = std::max<int64_t>(mr, S - s);
      }
      const int64_t nf = (s == 0) ? in_cnt[0] + 1 : std::max<int64_t>(in_cnt[s], 1);
      const int64_t nb = std::max<int64_t>(out_cnt[s] + (fin ? 1 : 0), 1);
      mtf = std::max(mtf, nf);
      mtb = std::max(mtb, nb);
      for (int k = 0; k < kNumLaneChoices; ++k) {
        f[k] += ceil_div(nf, (int64_t)1 << k);
        b[k] += ceil_div(nb, (int64_t)1 << k);
      }
    }
    for (int k = 0; k < kNumLaneChoices; ++k) {
      lr[(size_t)u].fwd[k] = (int32_t)(f[k] + ceil_div(std::max<int64_t>(nfinal, 1), (int64_t)1 << k));
      lr[(size_t)u].bwd[k] = (int32_t)b[k];
    }
    lr[(size_t)u].max_rel = (int32_t)mr;
    mtf = std::max(mtf, nfinal);  // the virtual super-final position holds one term per final state
    lr[(size_t)u].max_terms = (int32_t)(std::min<int64_t>(mtf, 64) | (std::min<int64_t>(mtb, 64) << 16));
  });
  if (bad.load() != 0) {
    *err = "lattice " + std::to_string(bad_lat.load()) +
           (bad.load() == KLAT_ERR_BAD_TOPOLOGY
                ? ": arc with dst <= src or out of range (lattice is not topologically sorted)"
                : ": malformed CSR (state ids out of range or arcs not grouped by source)");
    return bad.load();
  }

  PackedBatch& P = *out;
  std::vector<int8_t> lat_g;
  plan_layout(U, d.state_off, d.arc_off, lr.data(), opt, &lat_g);
  std::vector<std::vector<int64_t>> job_lats;
  int64_t chunk_total = 0;
  const int rc = build_jobs(U, d.state_off, lr.data(), lat_g, opt, &P, &job_lats, &chunk_total, err);
  if (rc != KLAT_OK) return rc;
  const size_t nslots = (size_t)chunk_total * kChunkRounds;  // chunk_total counts 16-byte units
  if (!P.sw.resize(nslots) || !P.sg.resize(nslots) || !P.sa.resize(nslots)) {
    *err = "out of host memory for the arc streams";
    return KLAT_ERR_OOM;
  }

  // ---- pass 2: fill ----
  std::atomic<int64_t> far(0);
  parallel_for((int64_t)job_lats.size(), threads, [&](int64_t j) {
    const SweepJob& job = P.jobs[j];
    const int G = job_lanes(job.tag), W = job_ring(job.tag), AL = job_stored_lanes(job.tag), LPW = 64 / G;
    const size_t b0 = (size_t)job.chunk_off * kChunkRounds;
    const size_t n = (size_t)job.nchunks * (size_t)(AL * LPW) * kChunkRounds;
    // padding: zero-weight terms that read cell 0 and park their (unused) result in the lane's
    // own dummy cell behind the ring, so that the consumer's ring write needs no predicate
    for (size_t i = 0; i < n; ++i) {
      const int unit = (int)((i / kChunkRounds) % (size_t)(AL * LPW));
      const int lane = (unit % LPW) * G + unit / LPW;
      P.sw[b0 + i] = (int32_t)(kNil | ((uint32_t)(W * LPW + lane / G) << kWrShift));
    }
    std::fill(P.sg.data() + b0, P.sg.data() + b0 + n, std::numeric_limits<float>::infinity());
    std::memset(&P.sa[b0], 0, n * sizeof(float));
    int64_t far_local = 0;
    auto slot_index = [&](int64_t r, int lane) -> size_t {
      return (size_t)stream_slot(job.chunk_off, G, AL, r / kChunkRounds, lane, (int)(r % kChunkRounds));
    };
    std::vector<int64_t> in_begin;
    std::vector<int32_t> in_arc, out_ord;
    for (size_t k = 0; k < job_lats[j].size(); ++k) {
      const int64_t u = job_lats[j][k];
      const int64_t s0 = d.state_off[u], a0 = d.arc_off[u];
      const int64_t S = d.state_off[u + 1] - s0, A = d.arc_off[u + 1] - a0;
      const int lane0 = (int)k * G;
      int64_t r = 0;  // next free round of this lattice
      int64_t nterm = 0;
      int64_t pos = 0;  // stream position of the state being fed (forward: state id; backward: S-1-id)
      auto begin_pos = [&](int64_t p) {
        nterm = 0;
        pos = p;
      };
      auto ring_cell = [&](int64_t p) -> uint32_t { return (uint32_t)((p & (W - 1)) * LPW + (int64_t)k); };
      auto emit = [&](int64_t rel, bool word, float g, float a, bool nil) {
        const int64_t rr = r + nterm / G;
        const int lane = lane0 + (int)(nterm % G);
        uint32_t w;
        if (nil) {
          w = kNil | ring_cell(pos);  // reads a valid cell; the weight is zero
        } else if (rel == 0) {
          w = kOne | ring_cell(pos);
        } else if (rel >= W) {
          w = kFar | (uint32_t)rel;  // the kernel resolves the cells of FAR terms
          far_local++;
        } else {
          w = ring_cell(pos - rel);
          if (rel == 1 && nterm < G) w |= kBypass;  // first round: the source was finished just before
        }
        if (word) w |= kWord;
        if ((w & kFar) == 0) w |= (uint32_t)(W * LPW + lane / G) << kWrShift;  // the lattice's dummy cell unless COMMIT (end_pos)
        const size_t ix = slot_index(rr, lane);
        P.sw[ix] = (int32_t)w;
        P.sg[ix] = nil ? std::numeric_limits<float>::infinity() : g;
        P.sa[ix] = a;
        nterm++;
      };
      auto end_pos = [&]() {
        if (nterm == 0) emit(0, false, 0.f, 0.f, true);
        const int64_t nr = ceil_div(nterm, G);
        const int64_t lastr = r + nr - 1;
        for (int g = 0; g < AL; ++g) {  // (the lanes beyond AL are not stored: the kernel copies these two bits)
          P.sw[slot_index(r, lane0 + g)] |= (int32_t)kFirst;
          P.sw[slot_index(lastr, lane0 + g)] |= (int32_t)kLast;
        }
        // the lattice's first lane stores the state's value at the end of its last round
        int32_t& cw = P.sw[slot_index(lastr, lane0)];
        if (((uint32_t)cw & kFar) == 0)
          cw = (int32_t)((((uint32_t)cw) & ~(kCellMask << kWrShift)) | kCommit | (ring_cell(pos) << kWrShift));
        r += nr;
      };
      if (job.dir == 0) {
        // in-arc lists by counting sort on dst (stable: sources ascending)
        in_begin.assign(S + 1, 0);
        for (int64_t i = 0; i < A; ++i) in_begin[d.dst[a0 + i] + 1]++;
        for (int64_t s = 0; s < S; ++s) in_begin[s + 1] += in_begin[s];
        in_arc.resize(A);
        {
          std::vector<int64_t> cur(in_begin.begin(), in_begin.end() - 1);
          for (int64_t i = 0; i < A; ++i) in_arc[cur[d.dst[a0 + i]]++] = (int32_t)i;
        }
        for (int64_t p = 0; p < S; ++p) {
          begin_pos(p);
          if (p == 0) emit(0, false, 0.f, 0.f, false);  // alpha[start] = 0
          for (int64_t q = in_begin[p]; q < in_begin[p + 1]; ++q) {
            const int64_t i = a0 + in_arc[q];
            emit(p - d.src[i], d.ilabel[i] != 0, d.graph[i], d.acoustic[i], false);
          }
          end_pos();
        }
        begin_pos(S);  // virtual super-final state
        for (int64_t s = 0; s < S; ++s)
          if (is_final(d, s0 + s)) emit(S - s, false, d.final_graph[s0 + s], d.final_acoustic[s0 + s], false);
        end_pos();
      } else {
        int64_t hi = A;  // arcs of state s are [lo,hi) scanning sources downwards
        for (int64_t s = S - 1; s >= 0; --s) {
          int64_t lo = hi;
          while (lo > 0 && d.src[a0 + lo - 1] == s) --lo;
          begin_pos(S - 1 - s);
          if (is_final(d, s0 + s)) emit(0, false, d.final_graph[s0 + s], d.final_acoustic[s0 + s], false);
          // farthest destination first, nearest last: the term that depends on the most
          // recently finished state is the last one to enter the accumulator
          out_ord.resize(hi - lo);
          for (int64_t i = lo; i < hi; ++i) out_ord[i - lo] = (int32_t)i;
          std::stable_sort(out_ord.begin(), out_ord.end(),
                           [&](int32_t x, int32_t y) { return d.dst[a0 + x] > d.dst[a0 + y]; });
          for (int32_t i : out_ord)
            emit(d.dst[a0 + i] - s, d.ilabel[a0 + i] != 0, d.graph[a0 + i], d.acoustic[a0 + i], false);
          end_pos();
          hi = lo;
        }
      }
    }
    far += far_local;
  });
  P.far_terms = far.load();

  // ---- posterior work items ----
  for (int64_t u = 0; u < U; ++u) {
    const int64_t a0 = d.arc_off[u], a1 = d.arc_off[u + 1];
    append_post_items(&P.items, a0, a1, u);
  }
  return KLAT_OK;
}

}  // namespace klat
