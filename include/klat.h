/*
 * klat.h -- C ABI of the MI355X-native lattice forward-backward / posterior path.
 *
 * The reference (jpuigcerver/kaldi-lattice-utils) has no FFI layer; the hot path sits behind
 * two de-facto interfaces:
 *   B2  double ComputeLatticeAlphasAndBetas(const CompactLattice&, bool viterbi,
 *                                           vector<double>* alpha, vector<double>* beta)
 *       int32 CompactLatticeStateTimes(const CompactLattice&, vector<int32>* times)
 *       as called at latbin/lattice-to-word-frame-post.cc:84-90,
 *       kwsbin2/lattice-word-index-segment.cc:62-63, kwsbin2/lattice-word-index-position.cc:66-67
 *   B1  the command lines of lattice-to-word-frame-post / lattice-word-index-{segment,position}.
 * This header is what a maintainer binds in place of B2; the C++ mirror
 * with the reference's own names lives in kaldi-lattice-utils_amd/host/lattice-functions.h.
 *
 * Conventions: plain pointers and sizes, caller owns every host buffer, device memory is
 * owned by opaque handles, no C++ exceptions cross the boundary.  Every function returns
 * KLAT_OK (0) or an error code; klat_last_error() gives the thread-local message.
 * One klat_ctx per host thread (it owns a HIP stream): calls on different contexts are
 * re-entrant, which is what TaskSequencer-style callers need
 * (kwsbin2/lattice-word-index-segment.cc:260-275).
 */
#ifndef KLAT_H_
#define KLAT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLAT_OK 0
#define KLAT_ERR_INVALID_ARG 1   /* null pointer, negative size, bad option */
#define KLAT_ERR_BAD_TOPOLOGY 2  /* arc with dst <= src or out of range: not top-sorted (KALDI_ERR in the reference) */
#define KLAT_ERR_HIP 3           /* HIP runtime error; message has the hipError string */
#define KLAT_ERR_NO_DEVICE 4     /* HIP extension present but no usable GPU */
#define KLAT_ERR_OOM 5
#define KLAT_ERR_STATE 6         /* call order violated (e.g. results requested before a sweep) */
#define KLAT_ERR_TIMES 7         /* inconsistent state times (KALDI_ASSERT in CompactLatticeStateTimes) */

typedef struct klat_ctx klat_ctx;
typedef struct klat_batch klat_batch;

/* A batch of top-sorted lattices, flattened CSR + SoA (float32 costs, int32 ids).
 * Replaces the per-utterance `CompactLattice` argument of B2 (a1 in SURVEY.md 8a).
 * Lattice u owns states [state_off[u], state_off[u+1]) and arcs [arc_off[u], arc_off[u+1]);
 * src/dst are lattice-local state ids, the start state is local id 0 and every arc has
 * dst > src (what TopSortCompactLatticeIfNeeded establishes, latbin/lattice-to-word-frame-post.cc:80).
 * Arcs of one lattice must be grouped by src in stored order (the order the reference's
 * ArcIterator visits them, needed by the first-maximum rule of
 * kwsbin2/lattice-word-index-position.cc:178).
 * final_graph/final_acoustic: +inf,+inf marks a non-final state (Weight::Zero()).
 * state_off / arc_off are read on the host; the arc and state arrays may be host OR device
 * pointers (lattices produced on the GPU need no round trip): the stream packer runs on the
 * device. */
typedef struct {
  int32_t num_lattices;
  const int64_t* state_off;     /* [num_lattices+1] */
  const int64_t* arc_off;       /* [num_lattices+1] */
  const int32_t* src;           /* [A] */
  const int32_t* dst;           /* [A] */
  const int32_t* ilabel;        /* [A] */
  const int32_t* olabel;        /* [A] */
  const float* graph;           /* [A]  Value1 */
  const float* acoustic;        /* [A]  Value2 */
  const float* final_graph;     /* [S] */
  const float* final_acoustic;  /* [S] */
  const int32_t* state_time;    /* [S]  CompactLatticeStateTimes; may be NULL if no accumulator is used */
} klat_batch_desc;

/* The same batch without what a CompactLattice does not hold: it is an acceptor (ilabel == olabel: one
 * label array), the source of an arc is implied by the arcs' grouping (out_degree per state), and the arc's
 * span dst - src fits a byte on every shape of BASELINE.json.  13 B per arc + 16 B per state cross PCIe
 * instead of 28 + 12; src / dst / olabel are rebuilt on the device.  A batch with a longer arc, or with
 * ilabel != olabel (the padded lattices of lattice-best-path2), takes klat_batch_desc. */
typedef struct {
  int32_t num_lattices;
  const int64_t* state_off;     /* [num_lattices+1] host */
  const int64_t* arc_off;       /* [num_lattices+1] host */
  const int32_t* out_degree;    /* [S] arcs leaving every state; the arcs follow in state order */
  const uint8_t* span;          /* [A] dst - src, 1 .. 255 */
  const int32_t* label;         /* [A] ilabel == olabel */
  const float* graph;           /* [A] */
  const float* acoustic;        /* [A] */
  const float* final_graph;     /* [S] */
  const float* final_acoustic;  /* [S] */
  const int32_t* state_time;    /* [S] or NULL */
} klat_batch_desc_compact;

/* Options of the three tools that reach the sweep. */
typedef struct {
  float graph_scale;        /* --graph-scale       default 1.0 */
  float acoustic_scale;     /* --acoustic-scale    default 1.0 */
  float insertion_penalty;  /* --insertion-penalty default 0.0 */
} klat_scales;

/* Packing options (device layout of the lockstep arc streams, DESIGN.md "Data layout"). */
typedef struct {
  int32_t lanes_per_lattice; /* G in {1,2,4,8,16,32,64} for every lattice; 0 = chosen per lattice from its
                                fanout and length (lattices of equal G form one layout class) */
  int32_t ring_slots;        /* W, power of two in [8,1024] with W * (64/G) <= 4032; 0 = choose from the batch */
  int32_t host_threads;      /* threads used by the host packer; 0 = hardware concurrency */
  /* NULL, or the scales every sweep of this batch will use: the packer then applies ScaleLattice /
   * AddWordInsPen itself (the same float roundings) and stores a term's cost as a float and a 16-bit
   * residual -- 10 bytes per stream slot instead of 12, one sixth less for the sweep to read, the cost exact
   * to 2^-40 of the batch's largest.  klat_forward_backward / klat_viterbi with other scales then return
   * KLAT_ERR_STATE.  The tools know their scales before they read a lattice and set this. */
  const klat_scales* fuse_scales;
  /* non-zero: arc and state arrays of klat_batch_desc that are device pointers (16-byte aligned, readable up to the
   * next multiple of 16 bytes -- any device allocator's blocks are) are used in place instead of copied: the caller keeps
   * them alive and unchanged until klat_batch_destroy.  (A copy of C2's arrays is 2.9 GB of device traffic, 0.6 ms.)
   * Host arrays are uploaded as always. */
  int32_t borrow_device_arrays;
} klat_pack_options;

typedef struct {
  int64_t num_states, num_arcs;       /* totals over the batch */
  int64_t fwd_slots, bwd_slots;       /* arc-stream slots incl. padding (12 B each) */
  int64_t far_terms;                  /* stream terms whose source fell outside the LDS ring */
  int32_t lanes_per_lattice, ring_slots; /* of the layout class that holds most of the stream */
  int32_t num_jobs;                   /* workgroups launched by the sweep kernel */
  int32_t max_rounds;                 /* longest lockstep stream, in rounds */
  int32_t num_layouts;                /* layout classes (distinct lanes_per_lattice) = sweep launches */
  int32_t level_lattices;             /* lattices on the level-synchronous path (wide lattices; no stream job) */
  int32_t chain_lattices;             /* long lattices the log-semiring sweep cuts into segments (klat_chain.hip) */
  int32_t chain_segments;             /* ... and the segments they were cut into */
  int32_t lean_jobs;                  /* sweep jobs whose streams the lean passes of the stream packer wrote (klat_pack_lean.hip:
                                         all their lattices have short arcs); num_jobs - lean_jobs took the general passes */
  int32_t reserved;
} klat_batch_info;

/* Per-kernel device times of the last klat_forward_backward (HIP events on the ctx stream). */
typedef struct {
  float sweep_ms;      /* forward+backward lockstep sweep: the fast sweep kernel alone for a batch of one
                          layout class, all classes' launches (fast + exact) otherwise */
  float totals_ms;     /* exact sweep (its workgroups exit at once unless a job was flagged); the per-lattice
                          totals are formed by the posterior pass and count in posterior_ms */
  float posterior_ms;  /* arc posterior kernel (KLAT_FB_NO_ARC_POSTERIORS: the totals kernel in its place) */
  float total_ms;      /* first event to last event */
} klat_timings;

const char* klat_last_error(void);
const char* klat_version(void);
int klat_device_count(int* count);

int klat_ctx_create(int device, klat_ctx** out);
/* The handle is gone after this call; the stream and the recycled device blocks behind it live on
 * until the last batch created on the context is destroyed (batches may outlive their context). */
void klat_ctx_destroy(klat_ctx* ctx);
void klat_ctx_retain(klat_ctx* ctx);  /* one more owner (libklat_dist's communicators); paired with klat_ctx_destroy */
int klat_ctx_sync(klat_ctx* ctx);
/* the ctx's hipStream_t, for callers that enqueue their own work or events behind it */
void* klat_ctx_stream(klat_ctx* ctx);
/* A context recycles the device blocks of its destroyed batches (hipFree synchronises the whole
 * device); it keeps at most a quarter of the device's memory that way.  klat_ctx_trim waits for the
 * context's stream and hands the cached blocks back to the runtime -- for hosts that share the GPU
 * with another allocator (PyTorch's, RCCL's): call it between phases, or when that allocator reports
 * out-of-memory.  The library itself empties the pools of all its contexts before it reports
 * KLAT_ERR_OOM.  released_bytes may be NULL. */
int klat_ctx_trim(klat_ctx* ctx, int64_t* released_bytes);

/* Page-locked host memory for the arrays of klat_batch_desc and for the result buffers of the
 * fetch calls: copies from / to it run at the full PCIe rate and asynchronously to the host
 * (pageable memory is staged by the runtime: results came back at 17 GB/s instead of ~50).  A
 * caller that keeps two contexts on two host threads -- one creating batch k+1 while the other
 * sweeps batch k and fetches batch k-1 -- overlaps the three (the contexts are re-entrant).
 * Optional: every entry point accepts ordinary memory as well. */
int klat_host_alloc(void** ptr, size_t bytes);
void klat_host_free(void* ptr);

/* Validates (KLAT_ERR_BAD_TOPOLOGY on a non-top-sorted lattice), builds the forward and
 * backward lockstep arc streams on the host and uploads everything the sweeps need.
 * The timed region of the reference starts after this point, too ("Timer does not include
 * preprocessing time", kwsbin2/lattice-word-index-segment.cc:140-141). */
int klat_batch_create(klat_ctx* ctx, const klat_batch_desc* desc, const klat_pack_options* opt,
                      klat_batch** out);
/* The same from the compact descriptor (host or device pointers); KLAT_ERR_INVALID_ARG if the out-degrees
 * of a lattice do not add up to its arcs. */
int klat_batch_create_compact(klat_ctx* ctx, const klat_batch_desc_compact* desc, const klat_pack_options* opt,
                              klat_batch** out);
void klat_batch_destroy(klat_batch* batch);
int klat_batch_get_info(const klat_batch* batch, klat_batch_info* info);

/* Where klat_batch_create spent its time.  The device passes (HIP events on the ctx stream): the count
 * pass (validation, arcs by destination, rounds per lane count), the level planner, the round prefix
 * sums, the stream fill, the posterior pass's one-byte source steps; host_ms is the wall time of the
 * whole call (uploads, the two waits for the device, layout planning on the host).  This is the
 * reference's untimed preprocessing (kwsbin2/lattice-word-index-segment.cc:41-62,140-141); a caller
 * that sweeps every batch once pays it per batch, so bench.py reports it (`one_shot`). */
typedef struct {
  float count_ms, level_plan_ms, rounds_ms, fill_ms, posterior_prep_ms;
  float device_ms;  /* sum of the device passes (the five above and chain_prep_ms) */
  float host_ms;
  float chain_prep_ms;  /* the arc records of the lattices swept in segments */
} klat_pack_timings;
int klat_batch_get_pack_timings(klat_batch* batch, klat_pack_timings* t);

/* Long lattices of a batch that waits for them are swept in segments (klat_batch_info.chain_lattices; DESIGN.md
 * 5).  Values that do not fit that path's arithmetic make the lockstep streams, which those lattices keep,
 * sweep them again inside the same klat_forward_backward.  *raised = 1 if the batch's last sweep did so. */
int klat_batch_chain_fallback(klat_batch* batch, int32_t* raised);

/* Topological depth of every lattice: the number of states on its longest path (the fewest
 * dependent steps ANY forward-backward needs; the lockstep streams take one round per state or
 * more).  Diagnostic: computed on the host from a copy of the arcs, on demand. */
int klat_batch_topological_depth(klat_batch* batch, int32_t* depth /* [num_lattices] */);

/* a2+a3+a7+a8: ScaleLattice / AddWordInsPen fused into the cost load, forward and backward
 * sweeps, per-lattice total = 0.5*(tot_fwd+tot_bwd), per-arc log posterior
 * alpha[src] - cost + beta[dst] - total, alpha/beta in natural log.  Asynchronous on the
 * ctx stream; results stay on the device until fetched. */
int klat_forward_backward(klat_batch* batch, const klat_scales* scales);
/* The same with flags.  KLAT_FB_NO_ARC_POSTERIORS: sweep and totals only -- no per-arc posterior pass
 * (8 B per arc written, 18 read).  The accumulators below do not need it: like the reference's loops
 * (kwsbin2/lattice-word-index-segment.cc:160-163,98-108) they form fw[src] + arc_lkh + bw[dst] from
 * alpha / beta themselves and subtract the total when a record is final.  klat_get_arc_logpost still
 * works afterwards (the pass then runs on demand). */
#define KLAT_FB_NO_ARC_POSTERIORS 1u
int klat_forward_backward_ex(klat_batch* batch, const klat_scales* scales, uint32_t flags);
int klat_get_timings(const klat_batch* batch, klat_timings* t);
/* Kernel times of the last `count` (<= 512) calls of klat_forward_backward on this context, oldest
 * first: a caller may enqueue many steps without waiting and read their times afterwards (waits for
 * the newest).  bench.py times its steps this way. */
int klat_ctx_timing_history(klat_ctx* ctx, int32_t count, klat_timings* out);
/* Number of sweep jobs (waves) of the last klat_forward_backward that left the fast sweep's
 * exponent window and were recomputed by the exact sweep (0 for ordinary lattices; non-zero when
 * the terms entering one state are hundreds of nats apart).  Results are exact either way. */
int klat_get_exact_sweep_jobs(klat_batch* batch, int32_t* count);

/* Synchronising fetches (device -> caller's host buffers); any pointer may be NULL.
 * alpha/beta: [S] doubles, -inf for unreachable / non-coaccessible states.
 * total: [num_lattices]; tot_fwd/tot_bwd as the reference's warning compares them. */
int klat_get_alpha_beta(klat_batch* batch, double* alpha, double* beta);
int klat_get_totals(klat_batch* batch, double* total, double* tot_fwd, double* tot_bwd);
int klat_get_arc_logpost(klat_batch* batch, double* logpost /* [A] */);

/* Interface B2 widened to a batch, host to host in ONE call: what
 *   double ComputeLatticeAlphasAndBetas(const CompactLattice&, bool, vector<double>* alpha, vector<double>* beta)
 * (latbin/lattice-to-word-frame-post.cc:88-90) gives for one lattice, for every lattice of the descriptor, plus the
 * arc log-posteriors alpha[src] - cost + beta[dst] - total (:102-104) when asked for.  The lattices are cut into
 * `pieces` runs of about equal arcs (0: chosen from the batch's size); piece k+1 crosses PCIe while piece k is packed
 * and swept and piece k-1's results come down, on three helper streams the context keeps for this call -- a C2 batch takes
 * 28 ms instead of the 39 ms of create / sweep / fetch one after the other (bench.py `boundary`).  The copies only
 * run beside one another from and into page-locked memory (klat_host_alloc); ordinary memory works, serially.
 * Any of alpha, beta [S], total [num_lattices], arc_logpost [A] may be NULL.  Nothing stays on the device. */
int klat_forward_backward_host(klat_ctx* ctx, const klat_batch_desc_compact* desc, const klat_pack_options* opt,
                               const klat_scales* scales, int32_t pieces, double* alpha, double* beta, double* total,
                               double* arc_logpost);

/* ---- tropical twin of the sweep ----
 * Viterbi forward / backward costs (min,+) over the same batch and, from them, ComputeLatticeBeam
 * (latbin/lattice-prune-dyn-beam.cc:27-90) and the decisions of PruneLattice(beam, &clat)
 * [upstream Kaldi; kwsbin2/lattice-word-index-segment.cc:49-51]: an arc (a final weight) is kept
 * iff the best complete path through it costs at most best + beam.  Costs are the ones the
 * reference holds at that point: scaled and penalised.  The caller applies the masks and
 * Connect()s (host-side graph surgery).  lattice_beam[u] is the beam at which lattice u loses
 * nothing.  Any output pointer may be NULL. */
int klat_viterbi(klat_batch* batch, const klat_scales* scales);
int klat_get_viterbi(klat_batch* batch, double* fwd_cost /* [S] */, double* bwd_cost /* [S] */, double* best /* [U] */);
int klat_prune_mask(klat_batch* batch, float beam, uint8_t* keep_arc /* [A] */, uint8_t* keep_final /* [S] */,
                    double* lattice_beam /* [U] */);

/* ---- second stage: the reference's accumulators over the same batch (a10/a11/a12) ----
 * Label filter of kwsbin2/lattice-word-index-segment.cc:148-154: label != 0 and
 * (include non-empty ? label in include : label not in exclude). */
typedef struct {
  const int32_t* include; int32_t num_include;
  const int32_t* exclude; int32_t num_exclude;
} klat_label_filter;

typedef struct { int32_t label, t0, t1, pad; double logp; } klat_segment_record;            /* 24 B */
typedef struct { int32_t label, pos, t0, t1; double logp; } klat_position_record;           /* 24 B */

/* Segment index (kwsbin2/lattice-word-index-segment.cc:143-174,98-126): one record per
 * distinct (olabel,t0,t1) per lattice, logp = LogAdd over its arcs - total, records of a
 * lattice sorted by (logp desc, label, t0, t1).  record_off[num_lattices+1] receives the
 * per-lattice ranges; returns the record count through *num_records.  Call once with
 * records == NULL to size the output. */
int klat_segment_index(klat_batch* batch, const klat_label_filter* filter,
                       int64_t* record_off, klat_segment_record* records, int64_t capacity,
                       int64_t* num_records);

/* Position index (kwsbin2/lattice-word-index-position.cc:144-187,100-127): state_len[S] is
 * the per-state input length from DisambiguateStateInputSequenceLength
 * (fstext/fstext-utils2.h:170-175); key (olabel, state_len[src]); value LogAdd sum plus the
 * (t0,t1) of the first arc attaining the maximum; pos emitted 1-based; sort (logp desc,
 * label, pos). */
int klat_position_index(klat_batch* batch, const int32_t* state_len,
                        const klat_label_filter* filter, int64_t* record_off,
                        klat_position_record* records, int64_t capacity, int64_t* num_records);

/* The records of the last sizing call (records == NULL) of klat_segment_index / klat_position_index,
 * gathered utterance-major into one device array owned by the batch (24 B each, record_off as that
 * call returned it; *record_off points at a host array of num_lattices+1 entries owned by the batch).
 * This is what the multi-GPU gather of klat_dist.h sends from: the records need not visit the host
 * on the ranks that do not write them (the serial writer of
 * kwsbin2/lattice-word-index-segment.cc:96-128 runs on rank 0 only). */
int klat_index_device_records(klat_batch* batch, const void** dev_records, const int64_t** record_off,
                              int64_t* num_lattices, int64_t* num_records);

/* Records the last sizing call of klat_frame_post stored on the device.  A frame at which no word arc
 * starts or ends has the posteriors of the frame before it (the accumulation of
 * latbin/lattice-to-word-frame-post.cc:93-117 adds the same arcs to both), so such frames share their
 * predecessor's records; the fetching call of klat_frame_post expands them into the dense output.
 * (num_entries of klat_frame_post counts every frame's entries, shared or not.) */
int klat_frame_stored_records(klat_batch* batch, int64_t* num_records);

/* Device time (HIP events on the ctx stream) of the index kernels / the frame-posterior kernels of
 * the last sizing call on this batch; 0 if none ran.  Either pointer may be NULL. */
int klat_get_accum_timings(klat_batch* batch, float* index_ms, float* frame_ms);

/* Utterance index (kwsbin2/lattice-word-index-utterance.cc:143-196; known answer
 * kwsbin2/README.md:25): for lattice u and each of its query words words[word_off[u] ..
 * word_off[u+1]) the log-probability that the word occurs at least once on the path,
 * logp = b0[start] - beta[start] (:155,:177).  The caller chooses the words (the reference takes
 * the lattice's distinct non-zero output labels through its include / exclude filter, :124-132)
 * and sorts the records (:291-300).  Runs after klat_forward_backward with the same scales. */
int klat_utterance_scores(klat_batch* batch, const int64_t* word_off, const int32_t* words, double* logp);

/* Frame posteriors (latbin/lattice-to-word-frame-post.cc:93-135): for each lattice and frame
 * k < num_frames[u] the (label, float32 logp) entries sorted by (logp desc, label asc).
 * frame_off[sum(num_frames)+1] indexes entries; lattice u's frames start at
 * frame_base[u] = sum_{v<u} num_frames[v]. */
int klat_frame_post(klat_batch* batch, const int32_t* num_frames, int64_t* frame_off,
                    int32_t* labels, float* logps, int64_t capacity, int64_t* num_entries);

/* ---- n-best over the batch's graphs in the tropical semiring ----
 * The k lowest-cost complete paths of every lattice over the caller's arc costs: the n-best
 * extension of ShortestPath(new_fst, &nbest_fst, 1) at latbin/lattice-best-path2.cc:185, where the
 * costs are 1 - P(label, position) (:171-173).  Costs are float32 >= 0 (+inf removes an arc) and
 * add left to right along a path (StdArc); final states add nothing (:157-160); paths are distinct
 * arc sequences, ascending by cost (equal costs: by the kernel's back-pointer order).
 * num_paths[u] <= k paths exist for lattice u; path n of lattice u has cost path_cost[u*k+n] and
 * path_len[u*k+n] non-zero output labels at path_labels[(u*k+n)*max_labels ..].  arc_cost may be a
 * host or a device pointer.  Does not need klat_forward_backward. */
#define KLAT_NBEST_MAX 128
int klat_nbest(klat_batch* batch, const float* arc_cost /* [A] */, int32_t k, int32_t max_labels,
               int32_t* num_paths /* [U] */, float* path_cost /* [U*k] */, int32_t* path_len /* [U*k] */,
               int32_t* path_labels /* [U*k*max_labels] */);

/* The streams of a batch as klat_batch_create built them on the device (for tests: they must equal
 * klat_pack_host's, bit for bit).  Arrays of fwd_slots + bwd_slots entries each. */
int klat_batch_get_streams(klat_batch* batch, int32_t* sw, float* sg, float* sa);
/* The same for a batch with fused costs (klat_pack_options.fuse_scales): the third array holds the 16-bit residuals. */
int klat_batch_get_fused_streams(klat_batch* batch, int32_t* sw, float* sg, int16_t* sl);

/* Host-only view of the packed lockstep streams (no device needed): the host packer that defines
 * the stream format klat_batch_create's device packer reproduces.  jobs: 24-byte records {int64 chunk_off; int32 nchunks, dir, lane_tab, tag} (tag: bits 0-1 wave priority, 8-15 G, 16-27 W of the job's layout class);
 * lanes: 16-byte records {int64 state_base; int32 num_states, lattice};
 * items: 16-byte records {int64 arc_begin; int32 count, lattice}.  Stream slot of
 * (chunk c, lane l, round r) is ((chunk_off + c) * 64 + l) * 4 + r. */
typedef struct klat_packed klat_packed;
int klat_pack_host(const klat_batch_desc* desc, const klat_pack_options* opt, klat_packed** out);
void klat_packed_destroy(klat_packed* p);
int klat_packed_info(const klat_packed* p, klat_batch_info* info, int64_t* num_slots,
                     int64_t* num_lane_infos, int64_t* num_items);
int klat_packed_copy(const klat_packed* p, int32_t* sw, float* sg, float* sa, void* jobs,
                     void* lanes, void* items);

/* Synthetic lattice generator of SURVEY.md Appendix D (bench/test plumbing, host only).
 * Fills caller-allocated arrays sized by klat_synth_sizes. */
int klat_synth_sizes(int32_t num_lattices, const int32_t* num_states, const int32_t* fanout,
                     int64_t* total_states, int64_t* total_arcs);
int klat_synth_generate(uint64_t seed, int64_t first_utt_index, int32_t num_lattices,
                        const int32_t* num_states, const int32_t* fanout, int64_t* state_off,
                        int64_t* arc_off, int32_t* src, int32_t* dst, int32_t* ilabel,
                        int32_t* olabel, float* graph, float* acoustic, float* final_graph,
                        float* final_acoustic, int32_t* state_time, int32_t host_threads);

/* A second synthetic shape, for lattices whose topological depth is far below their number of
 * states (real word lattices): levels of `width` states after the start state, every arc leads into
 * the next level, the last level is final; depth = 1 + ceil((S - 1) / width).  See klat_synth.cc for
 * the draw order (tests/test_synth.py restates both generators in Python). */
int klat_synth_wide_sizes(int32_t num_lattices, const int32_t* num_states, const int32_t* fanout, int32_t width,
                          int64_t* total_states, int64_t* total_arcs);
int klat_synth_wide_generate(uint64_t seed, int64_t first_utt_index, int32_t num_lattices,
                             const int32_t* num_states, const int32_t* fanout, int32_t width,
                             int64_t* state_off, int64_t* arc_off, int32_t* src, int32_t* dst,
                             int32_t* ilabel, int32_t* olabel, float* graph, float* acoustic,
                             float* final_graph, float* final_acoustic, int32_t* state_time,
                             int32_t host_threads);

#ifdef __cplusplus
}
#endif
#endif /* KLAT_H_ */
