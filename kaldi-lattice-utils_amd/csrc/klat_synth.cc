// klat_synth.cc -- synthetic lattice generator of SURVEY.md Appendix D (bench/test plumbing).
//
// Every "rng" in the Appendix-D pseudo-code is one fresh SplitMix64 draw, taken left to right:
//   times  : S-1 draws (one per state after the first), before any arc draw
//   per arc: [dst draw if j>0], label draw r (r%20==0 -> epsilon, else one more draw for the
//            word id), graph draw, acoustic draw.
// tests/test_synth.py holds an independent pure-Python restatement that must agree bit for bit.
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
}  // namespace

extern "C" {

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
