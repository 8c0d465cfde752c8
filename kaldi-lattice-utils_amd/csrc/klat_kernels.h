// klat_kernels.h -- kernel argument blocks and host-callable launchers (HIP only).
#ifndef KLAT_KERNELS_H_
#define KLAT_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klat_device.h"

namespace klat {

struct ScaleParams {
  float gs_f, as_f;  // graph_scale, acoustic_scale (BaseFloat in the reference)
  float pen;
  int scaled;     // graph_scale != 1 || acoustic_scale != 1 (the reference's own test)
  int penalized;  // insertion_penalty != 0
};

// Scales and penalty in the form the sweep's fast cost evaluation wants (klat_math.h stream_cost).
struct StreamCost {
  float gs, as, half_pen;
  int general;  // graph_scale < 0 or NaN, or pen/2 not exact: the sweep uses arc_cost() instead
  int plain;    // graph scale 1 and no penalty: the graph cost of a term is the stored float itself
};
inline StreamCost make_stream_cost(const ScaleParams& sp) {
  StreamCost c;
  c.gs = sp.scaled ? sp.gs_f : 1.0f;
  c.as = sp.scaled ? sp.as_f : 1.0f;
  c.half_pen = sp.penalized ? 0.5f * sp.pen : 0.0f;
  c.general = (!(c.gs >= 0.0f) || 2.0f * c.half_pen != (sp.penalized ? sp.pen : 0.0f)) ? 1 : 0;
  c.plain = (!c.general && c.gs == 1.0f && c.half_pen == 0.0f) ? 1 : 0;
  return c;
}

// The per-state outputs are kept "padded" on the device: lattice u owns alpha[state_off[u] + u ..
// state_off[u+1] + u], one slot more than its states.  The extra slot is position S of the
// forward sweep, the virtual super-final state whose value is tot_forward, so that the sweep
// stores every position it finishes through one address computation.  beta is padded the same
// way (its extra slot is unused).
__host__ __device__ inline int64_t padded_state_base(int64_t state_off_u, int64_t u) { return state_off_u + u; }

// One entry of the table behind the sweep's natural log: interval i of 16 covers mantissas with
// 2m in [1 + i/16, 1 + (i+1)/16); c is its midpoint.  (16 entries = 256 B of LDS and a degree-10
// series, where 128 entries took 2 KiB and degree 6: with the ring's dummy cells shared per lattice a
// workgroup needs 22 976 B at W = 32 and SEVEN of them fit a CU's 160 KiB -- a batch of 1 564 jobs,
// the 12 500 x 5 000-state shard, then runs as one wave of workgroups instead of 1 536 + 28, which
// cost 0.4 ms of 2.2: tools/tail_scan.py.)
struct __attribute__((aligned(16))) LnTabEntry {
  double inv2;  // 2 / c
  double lnc;   // ln(c / 2)
};
static const int kLnTabSize = 16;

struct SweepArgs {
  const SweepJob* jobs;
  const LaneInfo* lanes;
  const int32_t* sw;  // stream term words
  const float* sg;    // stream graph costs
  const float* sa;    // stream acoustic costs
  double* alpha;      // [S+U] padded natural-log forward values  (vector<double>* alpha of B2)
  double* beta;       // [S+U] padded natural-log backward values (vector<double>* beta of B2)
  int32_t* job_flags;  // [num_jobs] set by the fast sweep when a job needs the exact sweep
  double* sink;        // [num_jobs * 64] where lanes without a state to store park their write
  const LnTabEntry* ln_table;  // [kLnTabSize]
  ScaleParams sp;
  StreamCost sc;  // = make_stream_cost(sp)
  int32_t ring_slots;
  int32_t far_prefetch;  // the jobs hold sources older than the ring: prefetch their values (W >= 32)
  int32_t deep_prefetch = 0;  // the batch's workgroups fit six per CU: the fast sweep may hold 8 chunks per lane in registers
  // fused costs (klat_pack_options.fuse_scales): sg = (float)cost, sl = its residual in units of lo_unit, sa unused
  const int16_t* sl = nullptr;
  double lo_unit = 0.0;
  int32_t fused = 0;
  // null, or a device word: the launch does nothing while it is zero (the classes of the lattices swept in
  // segments run only when klat_chain.hip raised its flag)
  const int32_t* run_if = nullptr;
  // Which of a workgroup's two waves stages and which walks the chain.  The dispatcher puts them on SIMDs s and s + 2 of
  // the workgroup's CU and 256 consecutive workgroups on 256 CUs, so this decides which SIMDs of a CU carry its chain waves.
  // 0: wave 0 stages.  0x100: the waves of a FORWARD job swap roles (klat_capi.cc: kSweepRoleMask); 0x200: those of a
  // backward job; else bit (blockIdx >> 8) & 7 -- the workgroup's turn on its CU -- set: swapped (tools/sweep_roles.py)
  uint32_t role_mask = 0;
};

struct TotalsArgs {
  int32_t num_lattices;
  const int64_t* state_off;
  const double* alpha;  // padded
  const double* beta;   // padded
  double* total; double* tot_fwd; double* tot_bwd;
};

// Copies a padded per-state array into the caller's contiguous [S] layout.
struct UnpadArgs {
  int32_t num_lattices;
  const int64_t* state_off;
  const double* padded;
  double* dense;
};

struct PosteriorArgs {
  const PostItem* items;
  const int64_t* state_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic;
  const double* alpha; const double* beta;  // padded
  double* total; double* tot_fwd; double* tot_bwd;  // [U] written by the first item of every lattice
  double* logpost;
  ScaleParams sp;
  // src as one byte per arc: src[i] - src[i - 1] (0 on the first arc of an item) and the source of
  // every item's first arc -- the posterior pass then reads 21 B per arc instead of 24.  Null when
  // some step is above 255 (a run of 255 states without arcs): the pass reads src itself.
  const uint8_t* src_delta;
  const int32_t* item_src;
  const uint8_t* span;  // dst[i] - src[i] in one byte (same condition: no arc longer than 255 states)
};

struct PosteriorPrepArgs {
  const PostItem* items;
  const int32_t* src;
  const int32_t* dst;
  uint8_t* src_delta;   // [A]
  uint8_t* span;        // [A] dst - src
  int32_t* item_src;    // [items]
  int32_t* overflow;    // [1] set when a step does not fit a byte
};
// klat_batch_desc_compact -> src / dst of every arc (one workgroup per lattice); status[0] = 1 + the first
// lattice whose out-degrees do not add up to its arcs
struct ExpandArgs {
  const int64_t* state_off; const int64_t* arc_off;
  const int32_t* out_degree; const uint8_t* span;
  int32_t* src; int32_t* dst;
  int32_t* status;
};
hipError_t launch_expand_compact(const ExpandArgs& a, int num_lattices, hipStream_t stream);
hipError_t launch_posterior_prep(const PosteriorPrepArgs& a, int num_items, hipStream_t stream);

// Tropical twin: the Viterbi costs come from the sweep kernel's TROPICAL instantiation
// (launch_sweep_tropical: SweepArgs with alpha / beta = the padded forward / backward cost rows,
// slot S of a forward row = the best complete path); klat_viterbi.hip takes the beam decisions.
struct PruneArgs {
  const int64_t* state_off; const int64_t* arc_off;
  const int32_t* src; const int32_t* dst; const int32_t* ilabel;
  const float* graph; const float* acoustic; const float* final_graph; const float* final_acoustic;
  const double* fwd_cost; const double* bwd_cost;
  double beam;
  uint8_t* keep_arc;     // [A] or null
  uint8_t* keep_final;   // [S] or null
  double* lattice_beam;  // [U]
  ScaleParams sp;
};
hipError_t launch_sweep_tropical(const SweepArgs& args, int num_jobs, int G, hipStream_t stream);
hipError_t launch_prune(const PruneArgs& a, int num_lattices, hipStream_t stream);

// fast_begin / fast_end (may be null): events recorded right before and after the fast kernel
// which: the fast sweep, the exact one (for the jobs the fast one flagged), or both one after the other
static const int kSweepFast = 1, kSweepExact = 2;
hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream, hipEvent_t fast_begin = nullptr,
                        hipEvent_t fast_end = nullptr, int which = kSweepFast | kSweepExact);
hipError_t launch_totals(const TotalsArgs& a, hipStream_t stream);
hipError_t launch_unpad(const UnpadArgs& a, hipStream_t stream);
void fill_ln_table(LnTabEntry* host_table);  // kLnTabSize entries
hipError_t launch_posterior(const PosteriorArgs& a, int num_items, hipStream_t stream);

}  // namespace klat
#endif
