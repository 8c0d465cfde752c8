// klat_synth.cc -- synthetic lattice generator of SURVEY.md Appendix D (bench/test plumbing).
//
// Every "rng" in the Appendix-D pseudo-code is one fresh SplitMix64 draw, taken left to right:
//   times  : S-1 draws (one per state after the first), before any arc draw
//   per arc: [dst draw if j>0], label draw r (r%20==0 -> epsilon, else one more draw for the
//            word id), graph draw, acoustic draw.
// tests/test_synth.py holds an independent pure-Python restatement that must agree bit for bit.
//
// klat_synth_wide_*: this build's second shape, for lattices whose topological depth is far below
// their number of states (real word lattices: depth = number of words).  States come in levels of
// `width` states after the start state (level(s) = 1 + (s-1)/width, the last level may be partial);
// the start state has one arc to every state of level 1, every state of a later level but the last
// has `fanout` arcs into the NEXT level (the first to its own column, or the level's last state if
// that column does not exist; the others to a drawn column); the states of the last level are final
// with weight (0,0).  All states of a level share their time.  Draw order: one time draw per level
// 1..L, then per arc as above ([column draw if j>0 and src != 0], label, [word], graph, acoustic).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#include "../../include/klat.h"
#include "klat_internal.h"

namespace {
struct SplitMix64 {
  uint64_t x;
  explicit SplitMix64(uint64_t s) : x(s) {}
  uint64_t next() {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
};
inline int64_t lattice_arcs(int32_t S, int32_t fan) { return S >= 2 ? (int64_t)fan * (S - 1) : 0; }
struct WideShape {
  int32_t S, w, L;  // states, level width, index of the last level (0 if S == 1)
  explicit WideShape(int32_t S_, int32_t w_) : S(S_), w(w_), L(S_ <= 1 ? 0 : 1 + (S_ - 2) / w_) {}
  int32_t level(int32_t s) const { return s == 0 ? 0 : 1 + (s - 1) / w; }
  int32_t first(int32_t l) const { return l == 0 ? 0 : 1 + (l - 1) * w; }        // first state of level l
  int32_t count(int32_t l) const { return l == 0 ? 1 : std::min(w, S - first(l)); }  // states in level l
  int64_t arcs(int32_t fan) const { return S <= 1 ? 0 : (int64_t)count(1) + (int64_t)fan * (int64_t)(L - 1) * w; }
};
}  // namespace

extern "C" {

int klat_synth_wide_sizes(int32_t n, const int32_t* num_states, const int32_t* fanout, int32_t width,
                          int64_t* total_states, int64_t* total_arcs) {
  if (n < 0 || width < 1 || (n > 0 && (!num_states || !fanout)) || !total_states || !total_arcs)
    return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_wide_sizes: bad argument");
  int64_t S = 0, A = 0;
  for (int32_t u = 0; u < n; ++u) {
    if (num_states[u] < 1 || fanout[u] < 1)
      return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_wide_sizes: states and fanout must be >= 1");
    S += num_states[u];
    A += WideShape(num_states[u], width).arcs(fanout[u]);
  }
  *total_states = S;
  *total_arcs = A;
  return KLAT_OK;
}

int klat_synth_wide_generate(uint64_t seed, int64_t first_utt, int32_t n, const int32_t* num_states,
                             const int32_t* fanout, int32_t width, int64_t* state_off, int64_t* arc_off,
                             int32_t* src, int32_t* dst, int32_t* ilabel, int32_t* olabel, float* graph,
                             float* acoustic, float* final_graph, float* final_acoustic, int32_t* state_time,
                             int32_t host_threads) {
  int64_t St = 0, At = 0;
  int rc = klat_synth_wide_sizes(n, num_states, fanout, width, &St, &At);
  if (rc != KLAT_OK) return rc;
  if (!state_off || !arc_off || (At > 0 && (!src || !dst || !ilabel || !olabel || !graph || !acoustic)) ||
      (St > 0 && (!final_graph || !final_acoustic || !state_time)))
    return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_wide_generate: null output array");
  state_off[0] = 0;
  arc_off[0] = 0;
  for (int32_t u = 0; u < n; ++u) {
    state_off[u + 1] = state_off[u] + num_states[u];
    arc_off[u + 1] = arc_off[u] + WideShape(num_states[u], width).arcs(fanout[u]);
  }
  int threads = host_threads > 0 ? host_threads : (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  std::atomic<int32_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const int32_t u = next.fetch_add(1);
      if (u >= n) break;
      const WideShape sh(num_states[u], width);
      const int32_t S = sh.S, fan = fanout[u];
      const int64_t s0 = state_off[u];
      int64_t a = arc_off[u];
      SplitMix64 rng(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(first_utt + u + 1)));
      int32_t t = 0;
      state_time[s0] = 0;
      for (int32_t l = 1; l <= sh.L; ++l) {
        t += 1 + (int32_t)(rng.next() % 4);
        for (int32_t k = 0; k < sh.count(l); ++k) state_time[s0 + sh.first(l) + k] = t;
      }
      for (int32_t s = 0; s < S; ++s) {
        const bool fin = sh.level(s) == sh.L;
        final_graph[s0 + s] = fin ? 0.f : INFINITY;
        final_acoustic[s0 + s] = fin ? 0.f : INFINITY;
      }
      auto emit = [&](int32_t s, int32_t d) {
        const uint64_t r = rng.next();
        const int32_t label = (r % 20 == 0) ? 0 : 1 + (int32_t)(rng.next() % 1000);
        const double ug = (double)(rng.next() >> 11) * (1.0 / 9007199254740992.0);
        const double ua = (double)(rng.next() >> 11) * (1.0 / 9007199254740992.0);
        src[a] = s, dst[a] = d, ilabel[a] = label, olabel[a] = label;
        graph[a] = (float)(ug * 8.0);
        acoustic[a] = (float)(ua * 40.0);
        ++a;
      };
      if (S >= 2)
        for (int32_t k = 0; k < sh.count(1); ++k) emit(0, 1 + k);
      for (int32_t s = 1; s < S; ++s) {
        const int32_t l = sh.level(s);
        if (l >= sh.L) break;
        const int32_t nf = sh.first(l + 1), nc = sh.count(l + 1), col = s - sh.first(l);
        for (int32_t j = 0; j < fan; ++j) {
          const int32_t d = (j == 0) ? nf + std::min(col, nc - 1) : nf + (int32_t)(rng.next() % (uint64_t)nc);
          emit(s, d);
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  return KLAT_OK;
}

int klat_synth_sizes(int32_t n, const int32_t* num_states, const int32_t* fanout,
                     int64_t* total_states, int64_t* total_arcs) {
  if (n < 0 || (n > 0 && (!num_states || !fanout)) || !total_states || !total_arcs)
    return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_sizes: bad argument");
  int64_t S = 0, A = 0;
  for (int32_t u = 0; u < n; ++u) {
    if (num_states[u] < 1 || fanout[u] < 1)
      return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_sizes: states and fanout must be >= 1");
    S += num_states[u];
    A += lattice_arcs(num_states[u], fanout[u]);
  }
  *total_states = S;
  *total_arcs = A;
  return KLAT_OK;
}

int klat_synth_generate(uint64_t seed, int64_t first_utt, int32_t n, const int32_t* num_states,
                        const int32_t* fanout, int64_t* state_off, int64_t* arc_off, int32_t* src,
                        int32_t* dst, int32_t* ilabel, int32_t* olabel, float* graph,
                        float* acoustic, float* final_graph, float* final_acoustic,
                        int32_t* state_time, int32_t host_threads) {
  int64_t St = 0, At = 0;
  int rc = klat_synth_sizes(n, num_states, fanout, &St, &At);
  if (rc != KLAT_OK) return rc;
  if (!state_off || !arc_off || (At > 0 && (!src || !dst || !ilabel || !olabel || !graph || !acoustic)) ||
      (St > 0 && (!final_graph || !final_acoustic || !state_time)))
    return klat::set_error(KLAT_ERR_INVALID_ARG, "klat_synth_generate: null output array");
  state_off[0] = 0;
  arc_off[0] = 0;
  for (int32_t u = 0; u < n; ++u) {
    state_off[u + 1] = state_off[u] + num_states[u];
    arc_off[u + 1] = arc_off[u] + lattice_arcs(num_states[u], fanout[u]);
  }
  int threads = host_threads > 0 ? host_threads : (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  std::atomic<int32_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const int32_t u = next.fetch_add(1);
      if (u >= n) break;
      const int32_t S = num_states[u], fan = fanout[u];
      const int64_t s0 = state_off[u];
      int64_t a = arc_off[u];
      SplitMix64 rng(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(first_utt + u + 1)));
      state_time[s0] = 0;
      for (int32_t s = 1; s < S; ++s) state_time[s0 + s] = state_time[s0 + s - 1] + 1 + (int32_t)(rng.next() % 4);
      for (int32_t s = 0; s < S; ++s) {
        final_graph[s0 + s] = INFINITY;
        final_acoustic[s0 + s] = INFINITY;
      }
      final_graph[s0 + S - 1] = 0.f;
      final_acoustic[s0 + S - 1] = 0.f;
      for (int32_t s = 0; s + 1 < S; ++s) {
        for (int32_t j = 0; j < fan; ++j, ++a) {
          const int32_t span = (S - 2 - s) < 7 ? (S - 2 - s) : 7;
          const int32_t d = (j == 0) ? s + 1 : s + 1 + (int32_t)(rng.next() % (uint64_t)(span + 1));
          const uint64_t r = rng.next();
          const int32_t label = (r % 20 == 0) ? 0 : 1 + (int32_t)(rng.next() % 1000);
          const double ug = (double)(rng.next() >> 11) * (1.0 / 9007199254740992.0);
          const double ua = (double)(rng.next() >> 11) * (1.0 / 9007199254740992.0);
          src[a] = s;
          dst[a] = d;
          ilabel[a] = label;
          olabel[a] = label;
          graph[a] = (float)(ug * 8.0);
          acoustic[a] = (float)(ua * 40.0);
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  return KLAT_OK;
}

}  // extern "C"
