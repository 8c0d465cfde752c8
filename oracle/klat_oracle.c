/*
 * klat_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference algorithm on the forward-backward /
 * posterior hot path of jpuigcerver/kaldi-lattice-utils.  It is the parity checker for
 * the HIP path and the "port" CPU baseline of bench.py.  Nothing under
 * kaldi_lattice_utils_amd/ (the product) may include, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity pin: the numbers this file produces are checked by tests/test_oracle_golden.py
 * against the reference's own known-answer vectors kwsbin2/README.md:75 (segment index),
 * kwsbin2/README.md:122 (position index), kwsbin2/README.md:25 (utterance total) and the
 * state times of kwsbin2/README.md:61-64.  lattice-to-word-frame-post has no known-answer
 * vector in the reference -> that one accumulator is "parity unpinned";
 * it shares alpha/beta/total with the pinned tools.
 *
 * The arithmetic the reference delegates to upstream Kaldi (kaldi-asr/kaldi, unpinned:
 * travis/install_kaldi.sh:8 clones HEAD) is restated from its published algorithm
 * (SURVEY.md Appendix A); every function cites the reference call site that depends on it.
 *
 * Lattice representation used here (one lattice, "stored order"):
 *   S states, arcs grouped by source state in stored order (CSR): arc_begin[S+1];
 *   per arc: dst, ilabel, olabel, graph (Value1), acoustic (Value2), dur (= length of the
 *   transition-id string); per state: final_graph/final_acoustic (+inf,+inf == Zero ==
 *   non-final), final_dur.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KOR_OK 0
#define KOR_ERR_NOT_TOPSORTED 1
#define KOR_ERR_TIMES_INCONSISTENT 2
#define KOR_ERR_CYCLIC 3
#define KOR_ERR_ALLOC 4

/* ---- A.2  LogAdd(double,double) [UPSTREAM base/kaldi-math.h]; direct call sites
 * latbin/lattice-to-word-frame-post.cc:114, kwsbin2/lattice-word-index-segment.cc:170,
 * kwsbin2/lattice-word-index-position.cc:174. */
double kor_log_add(double x, double y) {
  /* kMinLogDiffDouble = Log(DBL_EPSILON) */
  static double min_log_diff = 0.0;
  if (min_log_diff == 0.0) min_log_diff = log(DBL_EPSILON);
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= min_log_diff) {
    return x + log1p(exp(diff));
  }
  return x; /* also the (-inf,-inf) and NaN-diff case */
}

/* LogSub [UPSTREAM]; only call site latbin/lattice-best-path2.cc:175. */
double kor_log_sub(double x, double y) {
  if (y >= x) {
    if (y == x) return -INFINITY;
    return NAN; /* KALDI_ERR in the reference */
  }
  double diff = y - x;
  double res = x + log(1.0 - exp(diff));
  if (res != res) return -INFINITY;
  return res;
}

/* ---- A.3  ScaleLattice(LatticeScale(g,a)) + AddWordInsPenToCompactLattice(pen)
 * [UPSTREAM fstext/lattice-utils-inl.h, lat/lattice-functions.cc]; call sites
 * latbin/lattice-to-word-frame-post.cc:64,74-78, kwsbin2/lattice-word-index-segment.cc:41-47.
 * Scales are BaseFloat (float) promoted to double; products are formed in double and
 * stored back as float.  Weights whose Value1 is +inf become Zero.  The scale step is
 * skipped when both scales == 1.0 and the penalty step when pen == 0 (caller's if). */
static void scale_pair(double gs, double as, float *g, float *a) {
  if (*g == INFINITY) {
    *g = INFINITY;
    *a = INFINITY;
    return;
  }
  double v1 = (double)*g, v2 = (double)*a;
  float ng = (float)(gs * v1 + 0.0 * v2);
  float na = (float)(0.0 * v1 + as * v2);
  *g = ng;
  *a = na;
}

void kor_scale_and_penalize(int64_t num_arcs, int64_t num_states, const int32_t *ilabel,
                            float *graph, float *acoustic, float *final_graph,
                            float *final_acoustic, float graph_scale, float acoustic_scale,
                            float insertion_penalty) {
  if (acoustic_scale != 1.0f || graph_scale != 1.0f) {
    double gs = (double)graph_scale, as = (double)acoustic_scale;
    for (int64_t i = 0; i < num_arcs; ++i) scale_pair(gs, as, &graph[i], &acoustic[i]);
    for (int64_t s = 0; s < num_states; ++s) {
      /* finals: only if != Zero (Zero stays Zero either way) */
      scale_pair(gs, as, &final_graph[s], &final_acoustic[s]);
    }
  }
  if (insertion_penalty != 0.0f) {
    for (int64_t i = 0; i < num_arcs; ++i)
      if (ilabel[i] != 0) graph[i] = graph[i] + insertion_penalty; /* float add */
  }
}

/* ---- A.6  top-sorted test used by TopSortCompactLatticeIfNeeded
 * (latbin/lattice-to-word-frame-post.cc:80); invariant asserted in-repo at
 * latbin/lattice-prune-dyn-beam.cc:48,76. */
int kor_is_top_sorted(int32_t S, const int64_t *arc_begin, const int32_t *dst) {
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i)
      if (dst[i] <= s || dst[i] >= S) return 0;
  return 1;
}

/* OpenFst TopSort numbering [UPSTREAM]: DFS from `start`, then from every still-unvisited
 * state in numeric order; new id = rank in reverse finishing order.  order[old] = new.
 * Returns KOR_ERR_CYCLIC on a back edge (KALDI_ERR in the reference). */
int kor_top_sort(int32_t S, int32_t start, const int64_t *arc_begin, const int32_t *dst,
                 int32_t *order) {
  if (S == 0) return KOR_OK;
  uint8_t *color = (uint8_t *)calloc((size_t)S, 1); /* 0 white 1 grey 2 black */
  int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
  int64_t *iter = (int64_t *)malloc(sizeof(int64_t) * (size_t)S);
  int32_t *finish = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
  if (!color || !stack || !iter || !finish) return KOR_ERR_ALLOC;
  int32_t nfin = 0, rc = KOR_OK;
  for (int32_t pass = -1; pass < S && rc == KOR_OK; ++pass) {
    int32_t root = pass < 0 ? start : pass;
    if (root < 0 || root >= S || color[root]) continue;
    int32_t sp = 0;
    stack[sp] = root;
    iter[root] = arc_begin[root];
    color[root] = 1;
    while (sp >= 0) {
      int32_t u = stack[sp];
      if (iter[u] < arc_begin[u + 1]) {
        int32_t v = dst[iter[u]++];
        if (color[v] == 1) {
          rc = KOR_ERR_CYCLIC;
          break;
        }
        if (color[v] == 0) {
          color[v] = 1;
          iter[v] = arc_begin[v];
          stack[++sp] = v;
        }
      } else {
        color[u] = 2;
        finish[nfin++] = u;
        --sp;
      }
    }
  }
  if (rc == KOR_OK)
    for (int32_t k = 0; k < S; ++k) order[finish[S - 1 - k]] = k;
  free(color);
  free(stack);
  free(iter);
  free(finish);
  return rc;
}

/* ---- A.5  CompactLatticeStateTimes [UPSTREAM lat/lattice-functions.cc]; call sites
 * latbin/lattice-to-word-frame-post.cc:84-85, kwsbin2/lattice-word-index-segment.cc:62,
 * kwsbin2/lattice-word-index-position.cc:67.  Returns utterance length through *utt_len
 * (0 when no final state).  *warn_flags bit0 = inconsistent utterance length,
 * bit1 = no final state. */
int kor_state_times(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                    const int32_t *dur, const float *final_graph, const float *final_acoustic,
                    const int32_t *final_dur, int32_t *times, int32_t *utt_len_out,
                    int32_t *warn_flags) {
  if (warn_flags) *warn_flags = 0;
  if (!kor_is_top_sorted(S, arc_begin, dst)) return KOR_ERR_NOT_TOPSORTED;
  for (int32_t s = 0; s < S; ++s) times[s] = -1;
  if (S > 0) times[0] = 0;
  int32_t utt_len = -1;
  for (int32_t s = 0; s < S; ++s) {
    int32_t cur = times[s];
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      int32_t d = dst[i];
      if (times[d] == -1)
        times[d] = cur + dur[i];
      else if (times[d] != cur + dur[i])
        return KOR_ERR_TIMES_INCONSISTENT; /* KALDI_ASSERT */
    }
    int is_final = !(final_graph[s] == INFINITY && final_acoustic[s] == INFINITY);
    if (is_final) {
      int32_t this_len = times[s] + (final_dur ? final_dur[s] : 0);
      if (utt_len == -1) {
        utt_len = this_len;
      } else if (this_len != utt_len) {
        if (warn_flags) *warn_flags |= 1;
        if (this_len > utt_len) utt_len = this_len;
      }
    }
  }
  if (utt_len == -1) {
    if (warn_flags) *warn_flags |= 2;
    utt_len = 0;
  }
  *utt_len_out = utt_len;
  return KOR_OK;
}

/* ---- A.1  ComputeLatticeAlphasAndBetas(lat, viterbi=false, &alpha, &beta)
 * [UPSTREAM lat/lattice-functions.cc]; call sites latbin/lattice-to-word-frame-post.cc:88-90,
 * kwsbin2/lattice-word-index-segment.cc:63, kwsbin2/lattice-word-index-position.cc:66.
 * Loop structure and ConvertToCost convention corroborated by the in-repo tropical twin
 * latbin/lattice-prune-dyn-beam.cc:35-87. */
double kor_alphas_betas(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                        const float *graph, const float *acoustic, const float *final_graph,
                        const float *final_acoustic, double *alpha, double *beta,
                        double *tot_fwd_out, double *tot_bwd_out) {
  for (int32_t s = 0; s < S; ++s) {
    alpha[s] = -INFINITY;
    beta[s] = -INFINITY;
  }
  if (S == 0) {
    if (tot_fwd_out) *tot_fwd_out = -INFINITY;
    if (tot_bwd_out) *tot_bwd_out = -INFINITY;
    return -INFINITY;
  }
  alpha[0] = 0.0;
  double tot_fwd = -INFINITY;
  for (int32_t s = 0; s < S; ++s) {
    double this_alpha = alpha[s];
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double arc_like = -((double)graph[i] + (double)acoustic[i]);
      alpha[dst[i]] = kor_log_add(alpha[dst[i]], this_alpha + arc_like);
    }
    if (!(final_graph[s] == INFINITY && final_acoustic[s] == INFINITY)) {
      double final_like = this_alpha - ((double)final_graph[s] + (double)final_acoustic[s]);
      tot_fwd = kor_log_add(tot_fwd, final_like);
    }
  }
  for (int32_t s = S - 1; s >= 0; --s) {
    double this_beta = -((double)final_graph[s] + (double)final_acoustic[s]);
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double arc_like = -((double)graph[i] + (double)acoustic[i]);
      this_beta = kor_log_add(this_beta, beta[dst[i]] + arc_like);
    }
    beta[s] = this_beta;
  }
  double tot_bwd = beta[0];
  if (tot_fwd_out) *tot_fwd_out = tot_fwd;
  if (tot_bwd_out) *tot_bwd_out = tot_bwd;
  return 0.5 * (tot_bwd + tot_fwd);
}

/* A.7 ComputeCompactLatticeBetas (kwsbin2/lattice-word-index-utterance.cc:121-123). */
double kor_betas(int32_t S, const int64_t *arc_begin, const int32_t *dst, const float *graph,
                 const float *acoustic, const float *final_graph, const float *final_acoustic,
                 double *beta) {
  for (int32_t s = S - 1; s >= 0; --s) {
    double this_beta = -((double)final_graph[s] + (double)final_acoustic[s]);
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double arc_like = -((double)graph[i] + (double)acoustic[i]);
      this_beta = kor_log_add(this_beta, beta[dst[i]] + arc_like);
    }
    beta[s] = this_beta;
  }
  return S > 0 ? beta[0] : -INFINITY;
}

/* ---- a8  per-arc through-likelihood, double cost form:
 * kwsbin2/lattice-word-index-segment.cc:160-163, kwsbin2/lattice-word-index-position.cc:160-163
 * (fw[src] + (-ConvertToCost(w)) + bw[dst]). */
void kor_arc_through(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                     const float *graph, const float *acoustic, const double *alpha,
                     const double *beta, double *through) {
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double arc_lkh = -((double)graph[i] + (double)acoustic[i]);
      through[i] = alpha[s] + arc_lkh + beta[dst[i]];
    }
}

/* label filter of kwsbin2/lattice-word-index-segment.cc:148-154 (same at
 * kwsbin2/lattice-word-index-position.cc:148-154).  Sets are sorted int32 arrays. */
static int in_sorted(const int32_t *set, int32_t n, int32_t v) {
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    int32_t mid = lo + (hi - lo) / 2;
    if (set[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < n && set[lo] == v;
}
static int valid_label(int32_t label, const int32_t *inc, int32_t ninc, const int32_t *exc,
                       int32_t nexc) {
  return label != 0 &&
         ((ninc > 0 && in_sorted(inc, ninc, label)) || (ninc == 0 && !in_sorted(exc, nexc, label)));
}

/* ---- generic helper: records carrying their arrival rank, stable grouping by key ---- */
typedef struct {
  int32_t k0, k1, k2; /* key fields */
  int32_t t0, t1;     /* payload (position index) */
  int64_t rank;       /* arrival order == reference map-insert order */
  double x;
} kor_rec;

static int cmp_key_rank(const void *pa, const void *pb) {
  const kor_rec *a = (const kor_rec *)pa, *b = (const kor_rec *)pb;
  if (a->k0 != b->k0) return a->k0 < b->k0 ? -1 : 1;
  if (a->k1 != b->k1) return a->k1 < b->k1 ? -1 : 1;
  if (a->k2 != b->k2) return a->k2 < b->k2 ? -1 : 1;
  if (a->rank != b->rank) return a->rank < b->rank ? -1 : 1;
  return 0;
}

/* ---- a10 + a12  segment index: kwsbin2/lattice-word-index-segment.cc:143-174 (accumulate
 * acc[label][(t0,t1)] = LogAdd(...), in arc order) and :98-126 (subtract total, sort by
 * (logp desc, label, t0, t1)).  Output arrays need room for num_arcs records.
 * Returns the number of records. */
static int cmp_segment(const void *pa, const void *pb) {
  const kor_rec *a = (const kor_rec *)pa, *b = (const kor_rec *)pb;
  if (b->x != a->x) return b->x < a->x ? -1 : 1;
  if (a->k0 != b->k0) return a->k0 < b->k0 ? -1 : 1;
  if (a->k1 != b->k1) return a->k1 < b->k1 ? -1 : 1;
  if (a->k2 != b->k2) return a->k2 < b->k2 ? -1 : 1;
  return 0;
}

int64_t kor_segment_index(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                          const int32_t *olabel, const float *graph, const float *acoustic,
                          const int32_t *times, const double *alpha, const double *beta,
                          double total, const int32_t *inc, int32_t ninc, const int32_t *exc,
                          int32_t nexc, int32_t *out_label, int32_t *out_t0, int32_t *out_t1,
                          double *out_logp) {
  int64_t A = S > 0 ? arc_begin[S] : 0;
  if (A == 0) return 0;
  kor_rec *r = (kor_rec *)malloc(sizeof(kor_rec) * (size_t)A);
  int64_t n = 0;
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      if (!valid_label(olabel[i], inc, ninc, exc, nexc)) continue;
      double arc_lkh = -((double)graph[i] + (double)acoustic[i]);
      r[n].k0 = olabel[i];
      r[n].k1 = times[s];
      r[n].k2 = times[dst[i]];
      r[n].t0 = r[n].t1 = 0;
      r[n].rank = n;
      r[n].x = alpha[s] + arc_lkh + beta[dst[i]];
      ++n;
    }
  qsort(r, (size_t)n, sizeof(kor_rec), cmp_key_rank);
  int64_t m = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i + 1;
    double acc = r[i].x;
    while (j < n && r[j].k0 == r[i].k0 && r[j].k1 == r[i].k1 && r[j].k2 == r[i].k2) {
      acc = kor_log_add(acc, r[j].x);
      ++j;
    }
    r[m] = r[i];
    r[m].x = acc - total;
    ++m;
    i = j;
  }
  qsort(r, (size_t)m, sizeof(kor_rec), cmp_segment);
  for (int64_t i = 0; i < m; ++i) {
    out_label[i] = r[i].k0;
    out_t0[i] = r[i].k1;
    out_t1[i] = r[i].k2;
    out_logp[i] = r[i].x;
  }
  free(r);
  return m;
}

/* ---- a11 + a12  position index: kwsbin2/lattice-word-index-position.cc:144-187
 * (key (label,pos=state_len[src]); value (LogAdd sum, best x, t0_best, t1_best), strict '>'
 * keeps the first maximum :178) and :100-127 (pos emitted 1-based :107, sort by
 * (logp desc, label, pos); t0/t1 not in the comparator). */
static int cmp_position(const void *pa, const void *pb) {
  const kor_rec *a = (const kor_rec *)pa, *b = (const kor_rec *)pb;
  if (b->x != a->x) return b->x < a->x ? -1 : 1;
  if (a->k0 != b->k0) return a->k0 < b->k0 ? -1 : 1;
  if (a->k1 != b->k1) return a->k1 < b->k1 ? -1 : 1;
  return 0;
}

int64_t kor_position_index(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                           const int32_t *olabel, const float *graph, const float *acoustic,
                           const int32_t *times, const int32_t *state_len, const double *alpha,
                           const double *beta, double total, const int32_t *inc, int32_t ninc,
                           const int32_t *exc, int32_t nexc, int32_t *out_label,
                           int32_t *out_pos, int32_t *out_t0, int32_t *out_t1,
                           double *out_logp) {
  int64_t A = S > 0 ? arc_begin[S] : 0;
  if (A == 0) return 0;
  kor_rec *r = (kor_rec *)malloc(sizeof(kor_rec) * (size_t)A);
  int64_t n = 0;
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      if (!valid_label(olabel[i], inc, ninc, exc, nexc)) continue;
      double arc_lkh = -((double)graph[i] + (double)acoustic[i]);
      r[n].k0 = olabel[i];
      r[n].k1 = state_len[s];
      r[n].k2 = 0;
      r[n].t0 = times[s];
      r[n].t1 = times[dst[i]];
      r[n].rank = n;
      r[n].x = alpha[s] + arc_lkh + beta[dst[i]];
      ++n;
    }
  qsort(r, (size_t)n, sizeof(kor_rec), cmp_key_rank);
  int64_t m = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i + 1;
    double acc = r[i].x, best = r[i].x;
    int32_t t0 = r[i].t0, t1 = r[i].t1;
    while (j < n && r[j].k0 == r[i].k0 && r[j].k1 == r[i].k1) {
      acc = kor_log_add(acc, r[j].x);
      if (r[j].x > best) {
        best = r[j].x;
        t0 = r[j].t0;
        t1 = r[j].t1;
      }
      ++j;
    }
    r[m] = r[i];
    r[m].k1 = r[i].k1 + 1; /* 1-based position */
    r[m].t0 = t0;
    r[m].t1 = t1;
    r[m].x = acc - total;
    ++m;
    i = j;
  }
  /* std::sort is not stable; ties under the comparator are (logp,label,pos)-equal, which
   * cannot happen for distinct keys, so any order of "equal" elements is the same. */
  qsort(r, (size_t)m, sizeof(kor_rec), cmp_position);
  for (int64_t i = 0; i < m; ++i) {
    out_label[i] = r[i].k0;
    out_pos[i] = r[i].k1;
    out_t0[i] = r[i].t0;
    out_t1[i] = r[i].t1;
    out_logp[i] = r[i].x;
  }
  free(r);
  return m;
}

/* ---- a9 + a12  frame posteriors: latbin/lattice-to-word-frame-post.cc:93-117 (for word
 * arcs, for k in [times[u], times[v]): acc[k][word] = LogAdd(acc, x), with the arc cost
 * formed as a FLOAT sum Value1()+Value2() :102-104) and :121-135 (logp = acc - total
 * narrowed to float :126, per-frame sort by (logp desc, label asc) on the float values).
 * Output: frame_begin[T+1] offsets into (out_label, out_logp).  Returns number of entries
 * or -1 when out_cap is too small.  *misaligned = 1 if a word arc with times[u] >=
 * times[v] was seen (the reference warns once and carries on :105-109). */
typedef struct {
  int32_t label;
  float logp;
} kor_post;
static int cmp_post(const void *pa, const void *pb) {
  const kor_post *a = (const kor_post *)pa, *b = (const kor_post *)pb;
  if (a->logp != b->logp) return b->logp < a->logp ? -1 : 1;
  if (a->label != b->label) return a->label < b->label ? -1 : 1;
  return 0;
}

int64_t kor_frame_post(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                       const int32_t *ilabel, const float *graph, const float *acoustic,
                       const int32_t *times, int32_t total_frames, const double *alpha,
                       const double *beta, double total, int64_t out_cap,
                       int64_t *frame_begin, int32_t *out_label, float *out_logp,
                       int32_t *misaligned) {
  if (misaligned) *misaligned = 0;
  int64_t n = 0;
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i)
      if (ilabel[i] != 0 && times[dst[i]] > times[s]) n += times[dst[i]] - times[s];
  kor_rec *r = (kor_rec *)malloc(sizeof(kor_rec) * (size_t)(n > 0 ? n : 1));
  int64_t c = 0;
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      if (ilabel[i] == 0) continue;
      float fcost = graph[i] + acoustic[i]; /* float sum, as the reference writes it */
      double x = alpha[s] + beta[dst[i]] - fcost;
      if (times[s] >= times[dst[i]] && misaligned) *misaligned = 1;
      for (int32_t k = times[s]; k < times[dst[i]]; ++k) {
        r[c].k0 = k;
        r[c].k1 = ilabel[i];
        r[c].k2 = 0;
        r[c].t0 = r[c].t1 = 0;
        r[c].rank = c;
        r[c].x = x;
        ++c;
      }
    }
  qsort(r, (size_t)c, sizeof(kor_rec), cmp_key_rank);
  int64_t m = 0;
  for (int32_t f = 0; f <= total_frames; ++f) frame_begin[f] = 0;
  kor_post *p = (kor_post *)malloc(sizeof(kor_post) * (size_t)(c > 0 ? c : 1));
  for (int64_t i = 0; i < c;) {
    int64_t j = i + 1;
    double acc = r[i].x;
    while (j < c && r[j].k0 == r[i].k0 && r[j].k1 == r[i].k1) {
      acc = kor_log_add(acc, r[j].x);
      ++j;
    }
    if (r[i].k0 >= 0 && r[i].k0 < total_frames) {
      p[m].label = r[i].k1;
      p[m].logp = (float)(acc - total);
      frame_begin[r[i].k0 + 1] += 1;
      ++m;
    }
    i = j;
  }
  for (int32_t f = 0; f < total_frames; ++f) frame_begin[f + 1] += frame_begin[f];
  free(r);
  if (m > out_cap) {
    free(p);
    return -1;
  }
  for (int32_t f = 0; f < total_frames; ++f)
    qsort(p + frame_begin[f], (size_t)(frame_begin[f + 1] - frame_begin[f]), sizeof(kor_post),
          cmp_post);
  for (int64_t i = 0; i < m; ++i) {
    out_label[i] = p[i].label;
    out_logp[i] = p[i].logp;
  }
  free(p);
  return m;
}

typedef struct {
  int32_t label;
  double logp;
} kor_post2;
static int cmp_post2(const void *pa, const void *pb) {
  const kor_post2 *a = (const kor_post2 *)pa, *b = (const kor_post2 *)pb;
  if (a->logp != b->logp) return b->logp < a->logp ? -1 : 1;
  if (a->label != b->label) return a->label < b->label ? -1 : 1;
  return 0;
}
static int cmp_i32(const void *pa, const void *pb) {
  const int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
  return a < b ? -1 : (a > b ? 1 : 0);
}

/* ---- utterance index, kwsbin2/lattice-word-index-utterance.cc:117-135,143-196 (SURVEY.md 8f f4;
 * pinned by kwsbin2/README.md:25).  For a query word w the reference composes the lattice with the
 * two-state automaton Sigma* w Sigma* (CreateQueryFst :33-56, RhoCompose :171) and takes the betas
 * of the result at its start state (:174-177).  The composed states are (s, q): q = 1 once w has
 * been read.  (s,1) carries every continuation of s, i.e. beta[s] of the lattice itself; (s,0)
 * is not final and follows an arc labelled w to (d,1), any other arc to (d,0):
 *     b0[s] = LogAdd over the arcs of s of  (olabel == w ? beta[d] : b0[d]) - cost.
 * score = b0[start] - total with total = beta[start] (:123, :155).  Words: the distinct non-zero
 * output labels (GetOutputSymbols :124-125), filtered as :126-132.  Records sorted by (logp desc,
 * label asc) :291-300.  beta: the betas of the lattice (kor_betas).  scratch: S doubles. */
int64_t kor_utterance_index(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                            const int32_t *olabel, const float *graph, const float *acoustic,
                            const double *beta, const int32_t *inc, int32_t ninc,
                            const int32_t *exc, int32_t nexc, int64_t out_cap, int32_t *out_label,
                            double *out_logp) {
  if (S <= 0) return 0;
  const int64_t A = arc_begin[S];
  int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A > 0 ? A : 1));
  int64_t nl = 0;
  for (int64_t i = 0; i < A; ++i)
    if (olabel[i] != 0) labels[nl++] = olabel[i];
  qsort(labels, (size_t)nl, sizeof(int32_t), cmp_i32);
  int64_t nu = 0;
  for (int64_t i = 0; i < nl; ++i)
    if (i == 0 || labels[i] != labels[i - 1]) labels[nu++] = labels[i];
  double *b0 = (double *)malloc(sizeof(double) * (size_t)S);
  kor_post2 *p = (kor_post2 *)malloc(sizeof(kor_post2) * (size_t)(nu > 0 ? nu : 1));
  int64_t m = 0;
  const double total = beta[0];
  for (int64_t k = 0; k < nu; ++k) {
    const int32_t w = labels[k];
    const int keep = ninc > 0 ? in_sorted(inc, ninc, w) : !in_sorted(exc, nexc, w);
    if (!keep) continue;
    for (int32_t s = S - 1; s >= 0; --s) {
      double acc = -INFINITY;
      for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
        const double arc_like = -((double)graph[i] + (double)acoustic[i]);
        acc = kor_log_add(acc, (olabel[i] == w ? beta[dst[i]] : b0[dst[i]]) + arc_like);
      }
      b0[s] = acc;
    }
    p[m].label = w;
    p[m].logp = b0[0] - total;
    ++m;
  }
  qsort(p, (size_t)m, sizeof(kor_post2), cmp_post2);
  free(labels);
  free(b0);
  if (m > out_cap) {
    free(p);
    return -1;
  }
  for (int64_t i = 0; i < m; ++i) {
    out_label[i] = p[i].label;
    out_logp[i] = p[i].logp;
  }
  free(p);
  return m;
}

/* ---- position posteriors, latbin/lattice-to-word-position-post.cc:100-135 (the "next" caller
 * that a14 unlocks).  Over the length-disambiguated lattice: for every arc with
 * ilabel != 0, pos = state_len[nextstate] :108 and acc[pos][word] = LogAdd(.., alpha[u] +
 * beta[next] - (Value1 + Value2)) :110-116 (the cost is the float sum the reference writes);
 * entry n-1 of the output holds position n, logp = (float)(acc - total) :122-124, sorted by
 * (logp desc, label asc) on the float values :128-135.  pos_begin[max_len+1]. */
int64_t kor_position_post(int32_t S, const int64_t *arc_begin, const int32_t *dst,
                          const int32_t *ilabel, const float *graph, const float *acoustic,
                          const int32_t *state_len, int32_t max_len, const double *alpha,
                          const double *beta, double total, int64_t out_cap, int64_t *pos_begin,
                          int32_t *out_label, float *out_logp) {
  const int64_t A = S > 0 ? arc_begin[S] : 0;
  kor_rec *r = (kor_rec *)malloc(sizeof(kor_rec) * (size_t)(A > 0 ? A : 1));
  int64_t c = 0;
  for (int32_t s = 0; s < S; ++s)
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      if (ilabel[i] == 0) continue;
      float fcost = graph[i] + acoustic[i];
      r[c].k0 = state_len[dst[i]];
      r[c].k1 = ilabel[i];
      r[c].k2 = 0;
      r[c].t0 = r[c].t1 = 0;
      r[c].rank = c;
      r[c].x = alpha[s] + beta[dst[i]] - fcost;
      ++c;
    }
  qsort(r, (size_t)c, sizeof(kor_rec), cmp_key_rank);
  for (int32_t n = 0; n <= max_len; ++n) pos_begin[n] = 0;
  kor_post *p = (kor_post *)malloc(sizeof(kor_post) * (size_t)(c > 0 ? c : 1));
  int64_t m = 0;
  for (int64_t i = 0; i < c;) {
    int64_t j = i + 1;
    double acc = r[i].x;
    while (j < c && r[j].k0 == r[i].k0 && r[j].k1 == r[i].k1) {
      acc = kor_log_add(acc, r[j].x);
      ++j;
    }
    if (r[i].k0 >= 1 && r[i].k0 <= max_len) { /* acc[0] is not used :101 */
      p[m].label = r[i].k1;
      p[m].logp = (float)(acc - total);
      pos_begin[r[i].k0] += 1;
      ++m;
    }
    i = j;
  }
  for (int32_t n = 0; n < max_len; ++n) pos_begin[n + 1] += pos_begin[n];
  free(r);
  if (m > out_cap) {
    free(p);
    return -1;
  }
  for (int32_t n = 0; n < max_len; ++n)
    qsort(p + pos_begin[n], (size_t)(pos_begin[n + 1] - pos_begin[n]), sizeof(kor_post), cmp_post);
  for (int64_t i = 0; i < m; ++i) {
    out_label[i] = p[i].label;
    out_logp[i] = p[i].logp;
  }
  free(p);
  return m;
}

/* ---- transcript length distribution, latbin/lattice-to-transcript-length-dist.cc:96-124 (second
 * adjacent caller behind a14).  Over the length-disambiguated lattice: for every final state u in
 * state order, acc[state_len[u]] = LogAdd(.., alpha[u] - ConvertToCost(Final(u))) :99-108;
 * logp = (float)(acc - total) :113-115; sorted by (logp desc, length asc) :117-124.
 * Returns the number of pairs (<= out_cap). */
int64_t kor_length_dist(int32_t S, const float *final_graph, const float *final_acoustic,
                        const int32_t *state_len, const double *alpha, double total,
                        int64_t out_cap, int32_t *out_len, float *out_logp) {
  int32_t max_len = -1;
  for (int32_t s = 0; s < S; ++s)
    if (state_len[s] > max_len) max_len = state_len[s];
  if (max_len < 0) return 0;
  double *acc = (double *)malloc(sizeof(double) * (size_t)(max_len + 1));
  char *seen = (char *)calloc((size_t)(max_len + 1), 1);
  for (int32_t s = 0; s < S; ++s) {
    if (isinf(final_graph[s]) && final_graph[s] > 0 && isinf(final_acoustic[s]) && final_acoustic[s] > 0)
      continue; /* Final(u) == Zero */
    const double lkh = alpha[s] - ((double)final_graph[s] + (double)final_acoustic[s]);
    const int32_t L = state_len[s];
    if (!seen[L]) {
      seen[L] = 1;
      acc[L] = lkh;
    } else {
      acc[L] = kor_log_add(acc[L], lkh);
    }
  }
  kor_post *p = (kor_post *)malloc(sizeof(kor_post) * (size_t)(max_len + 1));
  int64_t m = 0;
  for (int32_t L = 0; L <= max_len; ++L)
    if (seen[L]) {
      p[m].label = L;
      p[m].logp = (float)(acc[L] - total);
      ++m;
    }
  qsort(p, (size_t)m, sizeof(kor_post), cmp_post);
  free(acc);
  free(seen);
  if (m > out_cap) {
    free(p);
    return -1;
  }
  for (int64_t i = 0; i < m; ++i) {
    out_len[i] = p[i].label;
    out_logp[i] = p[i].logp;
  }
  free(p);
  return m;
}

/* ---- a14  DisambiguateStateInputSequenceLength, fstext/fstext-utils2.h:109-215.
 * New states are the (len, old_state) pairs reachable from (0,start) :150-168; their ids
 * are the rank of the pair in std::map (lexicographic) order :164-166; state_len[new]=len
 * :170-175; arcs are re-emitted per new state in the old stored order :178-212; the final
 * weight is copied :186.  use_input selects ilabel vs olabel for epsilon :157-158.
 * Outputs are malloc'ed by the callee (free with kor_free).  Returns max_len, or -1.
 * new_arc_old[i] = index of the originating old arc; new_state_old[s] = old state. */
typedef struct {
  int32_t len, st;
} kor_pair;
static int cmp_pair(const void *pa, const void *pb) {
  const kor_pair *a = (const kor_pair *)pa, *b = (const kor_pair *)pb;
  if (a->len != b->len) return a->len < b->len ? -1 : 1;
  if (a->st != b->st) return a->st < b->st ? -1 : 1;
  return 0;
}
static int64_t find_pair(const kor_pair *p, int64_t n, int32_t len, int32_t st) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (p[mid].len < len || (p[mid].len == len && p[mid].st < st))
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

void kor_free(void *p) { free(p); }

int32_t kor_disambiguate_length(int32_t S, int32_t start, const int64_t *arc_begin,
                                const int32_t *dst, const int32_t *ilabel,
                                const int32_t *olabel, int use_input, int32_t *new_S_out,
                                int64_t **new_arc_begin_out, int32_t **new_dst_out,
                                int64_t **new_arc_old_out, int32_t **new_state_old_out,
                                int32_t **state_len_out) {
  *new_S_out = 0;
  *new_arc_begin_out = NULL;
  *new_dst_out = NULL;
  *new_arc_old_out = NULL;
  *new_state_old_out = NULL;
  *state_len_out = NULL;
  if (S == 0 || start < 0) return 0;
  /* reachable (len,state) pairs: since the FST is acyclic and arcs raise len by 0/1,
   * len <= S.  Per state keep a growing sorted set of lengths via BFS over pairs. */
  int64_t cap = 1024, n = 0, head = 0;
  kor_pair *q = (kor_pair *)malloc(sizeof(kor_pair) * (size_t)cap);
  /* visited set: hash on (len,state) */
  int64_t hcap = 4096;
  int64_t *htab = (int64_t *)malloc(sizeof(int64_t) * (size_t)hcap);
  for (int64_t i = 0; i < hcap; ++i) htab[i] = -1;
#define KOR_HASH(l, s) ((((uint64_t)(uint32_t)(l)) * 0x9E3779B97F4A7C15ull) ^ (((uint64_t)(uint32_t)(s)) * 0xC2B2AE3D27D4EB4Full))
  q[n].len = 0;
  q[n].st = start;
  {
    uint64_t h = KOR_HASH(0, start) & (uint64_t)(hcap - 1);
    htab[h] = n;
  }
  ++n;
  int32_t max_len = 0;
  while (head < n) {
    kor_pair cur = q[head++];
    if (cur.len > max_len) max_len = cur.len;
    for (int64_t i = arc_begin[cur.st]; i < arc_begin[cur.st + 1]; ++i) {
      int32_t lab = use_input ? ilabel[i] : olabel[i];
      int32_t nl = lab == 0 ? cur.len : cur.len + 1;
      int32_t ns = dst[i];
      uint64_t h = KOR_HASH(nl, ns) & (uint64_t)(hcap - 1);
      int found = 0;
      while (htab[h] >= 0) {
        if (q[htab[h]].len == nl && q[htab[h]].st == ns) {
          found = 1;
          break;
        }
        h = (h + 1) & (uint64_t)(hcap - 1);
      }
      if (found) continue;
      if (n == cap) {
        cap *= 2;
        q = (kor_pair *)realloc(q, sizeof(kor_pair) * (size_t)cap);
      }
      q[n].len = nl;
      q[n].st = ns;
      htab[h] = n;
      ++n;
      if (n * 2 > hcap) { /* rehash */
        int64_t nh = hcap * 2;
        int64_t *nt = (int64_t *)malloc(sizeof(int64_t) * (size_t)nh);
        for (int64_t k = 0; k < nh; ++k) nt[k] = -1;
        for (int64_t k = 0; k < n; ++k) {
          uint64_t hh = KOR_HASH(q[k].len, q[k].st) & (uint64_t)(nh - 1);
          while (nt[hh] >= 0) hh = (hh + 1) & (uint64_t)(nh - 1);
          nt[hh] = k;
        }
        free(htab);
        htab = nt;
        hcap = nh;
      }
    }
  }
  free(htab);
  /* ids = rank in lexicographic (len,state) order */
  qsort(q, (size_t)n, sizeof(kor_pair), cmp_pair);
  int32_t NS = (int32_t)n;
  int64_t *nab = (int64_t *)malloc(sizeof(int64_t) * (size_t)(NS + 1));
  int32_t *nso = (int32_t *)malloc(sizeof(int32_t) * (size_t)NS);
  int32_t *sl = (int32_t *)malloc(sizeof(int32_t) * (size_t)NS);
  int64_t NA = 0;
  for (int32_t k = 0; k < NS; ++k) {
    nab[k] = NA;
    NA += arc_begin[q[k].st + 1] - arc_begin[q[k].st];
    nso[k] = q[k].st;
    sl[k] = q[k].len;
  }
  nab[NS] = NA;
  int32_t *nd = (int32_t *)malloc(sizeof(int32_t) * (size_t)(NA > 0 ? NA : 1));
  int64_t *nao = (int64_t *)malloc(sizeof(int64_t) * (size_t)(NA > 0 ? NA : 1));
  for (int32_t k = 0; k < NS; ++k) {
    int64_t o = nab[k];
    for (int64_t i = arc_begin[q[k].st]; i < arc_begin[q[k].st + 1]; ++i, ++o) {
      int32_t lab = use_input ? ilabel[i] : olabel[i];
      int32_t nl = lab == 0 ? q[k].len : q[k].len + 1;
      nd[o] = (int32_t)find_pair(q, n, nl, dst[i]);
      nao[o] = i;
    }
  }
  free(q);
  *new_S_out = NS;
  *new_arc_begin_out = nab;
  *new_dst_out = nd;
  *new_arc_old_out = nao;
  *new_state_old_out = nso;
  *state_len_out = sl;
  return max_len;
}

/* ---- A.4  PruneLattice(beam, lat) [UPSTREAM lat/lattice-functions.cc]; call sites
 * kwsbin2/lattice-word-index-segment.cc:49-51, kwsbin2/lattice-word-index-position.cc:49-51;
 * loop structure corroborated by the in-repo latbin/lattice-prune-dyn-beam.cc:35-87.
 * Input costs are the ones the tool holds at that point (already scaled / penalised).
 * Outputs: keep_arc[A] (0 = arc redirected to the dead state), final weights set to Zero in
 * place where pruned, keep_state[S] = accessible and co-accessible after Connect().
 * Returns the number of kept states. */
int32_t kor_prune_lattice(int32_t S, const int64_t *arc_begin, const int32_t *dst, const float *graph,
                          const float *acoustic, float *final_graph, float *final_acoustic,
                          double beam, uint8_t *keep_arc, uint8_t *keep_state) {
  if (S == 0) return 0;
  double *cost = (double *)malloc(sizeof(double) * (size_t)S);
  for (int32_t s = 0; s < S; ++s) cost[s] = INFINITY;
  cost[0] = 0.0;
  double best_final = INFINITY;
  for (int32_t s = 0; s < S; ++s) {
    double fc = cost[s];
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double nc = fc + ((double)graph[i] + (double)acoustic[i]);
      if (cost[dst[i]] > nc) cost[dst[i]] = nc;
    }
    double tf = fc + ((double)final_graph[s] + (double)final_acoustic[s]);
    if (tf < best_final) best_final = tf;
  }
  double cutoff = best_final + beam;
  for (int32_t s = S - 1; s >= 0; --s) {
    double fc = cost[s];
    double bc = (double)final_graph[s] + (double)final_acoustic[s];
    if (bc + fc > cutoff && bc != INFINITY) {
      final_graph[s] = INFINITY;
      final_acoustic[s] = INFINITY;
    }
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      double abc = ((double)graph[i] + (double)acoustic[i]) + cost[dst[i]];
      if (abc < bc) bc = abc;
      keep_arc[i] = (fc + abc > cutoff) ? 0 : 1;
    }
    cost[s] = bc;
  }
  free(cost);
  uint8_t *acc = (uint8_t *)calloc((size_t)S, 1), *co = (uint8_t *)calloc((size_t)S, 1);
  acc[0] = 1;
  for (int32_t s = 0; s < S; ++s)
    if (acc[s])
      for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i)
        if (keep_arc[i]) acc[dst[i]] = 1;
  for (int32_t s = S - 1; s >= 0; --s) {
    int c = !(final_graph[s] == INFINITY && final_acoustic[s] == INFINITY);
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1] && !c; ++i)
      if (keep_arc[i] && co[dst[i]]) c = 1;
    co[s] = (uint8_t)c;
  }
  int32_t n = 0;
  for (int32_t s = 0; s < S; ++s) {
    keep_state[s] = acc[s] && co[s];
    n += keep_state[s];
  }
  free(acc);
  free(co);
  if (!keep_state[0]) {
    memset(keep_state, 0, (size_t)S);
    n = 0;
  }
  return n;
}

/* ---- ComputeLatticeBeam, latbin/lattice-prune-dyn-beam.cc:27-90: Viterbi forward costs :36-58,
 * backward costs sharing the loop with the running cutoff :63-87, beam = cutoff - best :89.
 * fwd / bwd receive the two cost vectors (the reference overwrites one with the other). */
double kor_lattice_beam(int32_t S, const int64_t *arc_begin, const int32_t *dst, const float *graph,
                        const float *acoustic, const float *final_graph, const float *final_acoustic,
                        double *fwd, double *bwd) {
  if (S == 0) return 0.0;
  for (int32_t s = 0; s < S; ++s) fwd[s] = INFINITY;
  fwd[0] = 0.0;
  double best_final_cost = INFINITY;
  for (int32_t s = 0; s < S; ++s) {
    const double this_forward_cost = fwd[s];
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      const double next_forward_cost = this_forward_cost + ((double)graph[i] + (double)acoustic[i]);
      if (fwd[dst[i]] > next_forward_cost) fwd[dst[i]] = next_forward_cost;
    }
    const double this_final_cost = this_forward_cost + ((double)final_graph[s] + (double)final_acoustic[s]);
    if (this_final_cost < best_final_cost) best_final_cost = this_final_cost;
  }
  double cutoff = best_final_cost;
  for (int32_t s = S - 1; s >= 0; --s) {
    const double this_forward_cost = fwd[s];
    double this_backward_cost = (double)final_graph[s] + (double)final_acoustic[s];
    if (this_backward_cost + this_forward_cost > cutoff && this_backward_cost != INFINITY)
      cutoff = this_backward_cost + this_forward_cost;
    for (int64_t i = arc_begin[s]; i < arc_begin[s + 1]; ++i) {
      const double arc_cost = (double)graph[i] + (double)acoustic[i];
      const double arc_backward_cost = arc_cost + bwd[dst[i]];
      const double this_fb_cost = this_forward_cost + arc_backward_cost;
      if (arc_backward_cost < this_backward_cost) this_backward_cost = arc_backward_cost;
      if (this_fb_cost > cutoff) cutoff = this_fb_cost;
    }
    bwd[s] = this_backward_cost;
  }
  return cutoff - best_final_cost;
}

/* ---- timed CPU baseline for bench.py ("port"): fb + arc through-likelihood over a batch
 * of lattices held as one flat CSR (state_off/arc_off per lattice, arc_begin global per
 * state, lattice-local dst).  Scales are applied to private copies first (a2/a3), as the
 * tools do.  Single-threaded by design; bench.py spreads lattices over processes. */
void kor_batch_fb_post(int32_t num_lat, const int64_t *state_off, const int64_t *arc_off,
                       const int64_t *arc_begin_global, const int32_t *dst,
                       const int32_t *ilabel, const float *graph_in, const float *acoustic_in,
                       const float *final_graph_in, const float *final_acoustic_in,
                       float graph_scale, float acoustic_scale, float insertion_penalty,
                       double *alpha, double *beta, double *total, double *through) {
  for (int32_t u = 0; u < num_lat; ++u) {
    int64_t s0 = state_off[u], a0 = arc_off[u];
    int32_t S = (int32_t)(state_off[u + 1] - s0);
    int64_t A = arc_off[u + 1] - a0;
    int64_t *ab = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S + 1));
    float *g = (float *)malloc(sizeof(float) * (size_t)(A > 0 ? A : 1));
    float *a = (float *)malloc(sizeof(float) * (size_t)(A > 0 ? A : 1));
    float *fg = (float *)malloc(sizeof(float) * (size_t)(S > 0 ? S : 1));
    float *fa = (float *)malloc(sizeof(float) * (size_t)(S > 0 ? S : 1));
    for (int32_t s = 0; s <= S; ++s) ab[s] = arc_begin_global[s0 + s] - a0;
    memcpy(g, graph_in + a0, sizeof(float) * (size_t)A);
    memcpy(a, acoustic_in + a0, sizeof(float) * (size_t)A);
    memcpy(fg, final_graph_in + s0, sizeof(float) * (size_t)S);
    memcpy(fa, final_acoustic_in + s0, sizeof(float) * (size_t)S);
    kor_scale_and_penalize(A, S, ilabel + a0, g, a, fg, fa, graph_scale, acoustic_scale,
                           insertion_penalty);
    total[u] = kor_alphas_betas(S, ab, dst + a0, g, a, fg, fa, alpha + s0, beta + s0, NULL, NULL);
    kor_arc_through(S, ab, dst + a0, g, a, alpha + s0, beta + s0, through + a0);
    free(ab);
    free(g);
    free(a);
    free(fg);
    free(fa);
  }
}
