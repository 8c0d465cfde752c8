/*
 * klat_dist.h -- C ABI of the multi-GPU part of the path: utterance sharding and the gather of
 * index records to the writing rank (lib/libklat_dist.so; links RCCL).
 *
 * What it replaces in the reference: TaskSequencer<LatticeScorerTask> with ordered emission
 * (kwsbin2/lattice-word-index-segment.cc:260-275: one lattice per worker thread; the task
 * destructors, :96-128, run serially in submission order and write the records, so the output
 * order is the input order).  Here the workers are GPUs, one process each: utterances are dealt
 * to the ranks as contiguous ranges of equal cumulative arc count, every rank runs its range
 * through klat.h on its own GPU with no data-path collective, and the only exchange is the gather
 * of the 24-byte index records to the rank that writes -- all-gather of the per-rank counts, then
 * grouped point-to-point sends straight from the ranks' device buffers into the root's device
 * buffer (RCCL over xGMI; RCCL has no gather-v).  Concatenating the ranks in rank order restores
 * the input order, which is the reference's emission order.
 *
 * Conventions as in klat.h: plain pointers and sizes, KLAT_OK or an error code, klat_last_error().
 */
#ifndef KLAT_DIST_H_
#define KLAT_DIST_H_

#include "klat.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KLAT_ERR_COMM 8 /* the transport (RCCL or the caller's) reported a failure */

typedef struct klat_comm klat_comm;

/* Contiguous utterance ranges with (nearly) equal total arcs: bounds[world+1], rank r owns
 * utterances [bounds[r], bounds[r+1]).  One unit of weight per utterance on top of its arcs, so
 * that empty lattices spread out as well.  Pure host arithmetic. */
int klat_partition_by_arcs(const int64_t* arc_counts, int64_t num_utts, int32_t world, int64_t* bounds);

/* ---- RCCL transport (the product path) ----
 * Rank 0 obtains a 128-byte id and hands it to the other ranks by whatever means the launcher has
 * (a pipe or shared page between forked workers, torch.distributed, a file); every rank then
 * creates its communicator on the stream of its klat_ctx. */
#define KLAT_UNIQUE_ID_BYTES 128
int klat_rccl_unique_id(void* id /* [KLAT_UNIQUE_ID_BYTES] */);
int klat_comm_create_rccl(klat_ctx* ctx, int32_t rank, int32_t world, const void* id, klat_comm** out);

/* ---- caller-supplied transport over HOST memory ----
 * For launchers that bring their own collectives (an MPI or gloo process group) and for the
 * tests, which run the sharding / ordering / reassembly logic of this file as two CPU processes.
 * Every callback returns 0 on success.  send/recv are posted between group_begin and group_end
 * and are complete when group_end returns. */
typedef struct {
  void* user;
  int (*all_gather_i64)(void* user, const int64_t* send, int64_t* recv /* [world*count] */, int32_t count);
  int (*group_begin)(void* user);
  int (*send)(void* user, const void* buf, int64_t bytes, int32_t peer);
  int (*recv)(void* user, void* buf, int64_t bytes, int32_t peer);
  int (*group_end)(void* user);
} klat_transport;
int klat_comm_create_custom(int32_t rank, int32_t world, const klat_transport* transport, klat_comm** out);

/* transport_ranks: the size the transport itself reports (ncclCommCount), = world if it is healthy */
int klat_comm_info(const klat_comm* comm, int32_t* rank, int32_t* world, int32_t* transport_ranks);
void klat_comm_destroy(klat_comm* comm);

/* Gathers every rank's records (utterance-major, record_bytes each, record_off[num_utts+1] on the
 * host) to `root`.  `records` is device memory with the RCCL transport and host memory with a
 * custom one.  On root, *total_utts / *total_records receive the sizes over all ranks (0
 * elsewhere); the gathered arrays stay inside the communicator until the next gather. */
int klat_gather_records(klat_comm* comm, const void* records, int32_t record_bytes, const int64_t* record_off,
                        int64_t num_utts, int32_t root, int64_t* total_utts, int64_t* total_records);
/* The same for the index of a batch: after the sizing call (records == NULL) of
 * klat_segment_index / klat_position_index on every rank. */
int klat_gather_index(klat_comm* comm, klat_batch* batch, int32_t root, int64_t* total_utts, int64_t* total_records);
/* Root only: the gathered records in rank order (= input order) into the caller's host arrays;
 * either pointer may be NULL. */
int klat_gather_fetch(klat_comm* comm, int64_t* record_off /* [total_utts+1] */, void* records, int64_t capacity_records);
/* Seconds the last gather spent between its first collective and the completion of the payload
 * (host clock around a synchronised region), and the payload bytes that arrived at root. */
int klat_gather_stats(const klat_comm* comm, double* seconds, int64_t* payload_bytes);

#ifdef __cplusplus
}
#endif
#endif /* KLAT_DIST_H_ */
