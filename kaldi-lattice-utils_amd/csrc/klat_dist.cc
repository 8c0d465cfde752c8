// klat_dist.cc -- the C ABI of include/klat_dist.h: utterance sharding by cumulative arcs and the
// gather of index records to the writing rank, over RCCL from device buffers (the product) or over
// a caller-supplied host transport (foreign process groups, CPU tests).
//
// One process per GPU.  The gather is not a ring: every peer sends its records straight to the
// root (one xGMI link each), after an all-gather of the per-rank sizes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/klat_dist.h"

namespace klat {
int set_error(int code, const std::string& msg);  // klat_capi.cc (libklat.so)
}
using klat::set_error;

namespace {

int hip_err(hipError_t e, const char* what) {
  return set_error(e == hipErrorOutOfMemory ? KLAT_ERR_OOM : KLAT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
int nccl_err(ncclResult_t r, const char* what) {
  return set_error(KLAT_ERR_COMM, std::string(what) + ": " + ncclGetErrorString(r));
}
#define D_HIP(expr)                                   \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return hip_err(_e, #expr);  \
  } while (0)
#define D_NCCL(expr)                                    \
  do {                                                  \
    ncclResult_t _r = (expr);                           \
    if (_r != ncclSuccess) return nccl_err(_r, #expr);  \
  } while (0)

// A byte buffer in the memory the transport moves: device memory for RCCL, host memory otherwise.
struct Buf {
  bool device = false;
  unsigned char* p = nullptr;
  size_t cap = 0;
  ~Buf() { release(); }
  void release() {
    if (p) {
      if (device) (void)hipFree(p);
      else std::free(p);
    }
    p = nullptr;
    cap = 0;
  }
  int reserve(size_t bytes) {
    if (bytes <= cap && p) return KLAT_OK;
    release();
    const size_t want = bytes ? bytes : 1;
    if (device) {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want);
      if (e != hipSuccess) {
        p = nullptr;
        return hip_err(e, "hipMalloc (gather buffer)");
      }
    } else {
      p = static_cast<unsigned char*>(std::malloc(want));
      if (!p) return set_error(KLAT_ERR_OOM, "host allocation failed (gather buffer)");
    }
    cap = want;
    return KLAT_OK;
  }
};

}  // namespace

struct klat_comm {
  int32_t rank = 0, world = 1;
  bool rccl = false;
  // RCCL
  ncclComm_t nccl = nullptr;
  klat_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  int device = 0;
  Buf meta_dev;  // staging of the size all-gather
  // custom
  klat_transport t = {};
  // result of the last gather (root)
  Buf records, counts;        // gathered records; per-utterance record counts (int64)
  Buf send_counts;            // this rank's per-utterance counts in transport memory
  int32_t record_bytes = 0;
  int64_t total_utts = 0, total_records = 0;
  bool have_result = false;
  double last_seconds = 0.0;
  int64_t last_payload = 0;
};

namespace {

int all_gather_i64(klat_comm* c, const int64_t* send, int64_t* recv, int32_t count) {
  if (!c->rccl) {
    if (c->t.all_gather_i64(c->t.user, send, recv, count) != 0) return set_error(KLAT_ERR_COMM, "transport: all_gather_i64 failed");
    return KLAT_OK;
  }
  const size_t each = sizeof(int64_t) * (size_t)count;
  int rc = c->meta_dev.reserve(each * ((size_t)c->world + 1));
  if (rc != KLAT_OK) return rc;
  int64_t* d_send = reinterpret_cast<int64_t*>(c->meta_dev.p);
  int64_t* d_recv = d_send + count;
  D_HIP(hipMemcpyAsync(d_send, send, each, hipMemcpyHostToDevice, c->stream));
  D_NCCL(ncclAllGather(d_send, d_recv, (size_t)count, ncclInt64, c->nccl, c->stream));
  D_HIP(hipMemcpyAsync(recv, d_recv, each * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  D_HIP(hipStreamSynchronize(c->stream));
  return KLAT_OK;
}

int group_begin(klat_comm* c) {
  if (c->rccl) {
    D_NCCL(ncclGroupStart());
    return KLAT_OK;
  }
  return c->t.group_begin(c->t.user) == 0 ? KLAT_OK : set_error(KLAT_ERR_COMM, "transport: group_begin failed");
}
int group_end(klat_comm* c) {
  if (c->rccl) {
    D_NCCL(ncclGroupEnd());
    return KLAT_OK;
  }
  return c->t.group_end(c->t.user) == 0 ? KLAT_OK : set_error(KLAT_ERR_COMM, "transport: group_end failed");
}
int p2p(klat_comm* c, bool send, void* buf, int64_t bytes, int32_t peer) {
  if (bytes == 0) return KLAT_OK;  // both sides know the sizes: nothing is posted for empty parts
  if (c->rccl) {
    if (send) D_NCCL(ncclSend(buf, (size_t)bytes, ncclUint8, peer, c->nccl, c->stream));
    else D_NCCL(ncclRecv(buf, (size_t)bytes, ncclUint8, peer, c->nccl, c->stream));
    return KLAT_OK;
  }
  const int r = send ? c->t.send(c->t.user, buf, bytes, peer) : c->t.recv(c->t.user, buf, bytes, peer);
  return r == 0 ? KLAT_OK : set_error(KLAT_ERR_COMM, send ? "transport: send failed" : "transport: recv failed");
}
int copy_in(klat_comm* c, void* dst, const void* src, size_t bytes, bool src_is_host) {
  if (bytes == 0) return KLAT_OK;
  if (c->rccl) {
    D_HIP(hipMemcpyAsync(dst, src, bytes, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
  } else {
    std::memcpy(dst, src, bytes);
  }
  return KLAT_OK;
}

}  // namespace

extern "C" {

int klat_partition_by_arcs(const int64_t* arc_counts, int64_t n, int32_t world, int64_t* bounds) {
  if (n < 0 || world < 1 || !bounds || (n > 0 && !arc_counts)) return set_error(KLAT_ERR_INVALID_ARG, "klat_partition_by_arcs: bad argument");
  // prefix sums of (arcs + 1): rank r starts at the utterance boundary nearest r/world of the total,
  // so the ranges are contiguous, ordered and never overlap
  std::vector<int64_t> csum((size_t)n + 1, 0);
  for (int64_t u = 0; u < n; ++u) {
    if (arc_counts[u] < 0) return set_error(KLAT_ERR_INVALID_ARG, "klat_partition_by_arcs: negative arc count");
    csum[(size_t)u + 1] = csum[(size_t)u] + arc_counts[u] + 1;
  }
  const int64_t total = csum[(size_t)n];
  bounds[0] = 0;
  for (int32_t r = 1; r < world; ++r) {
    const int64_t target = (int64_t)((__int128)total * r / world);
    int64_t lo = 0, hi = n + 1;  // first index with csum >= target
    while (lo < hi) {
      const int64_t mid = (lo + hi) / 2;
      if (csum[(size_t)mid] < target) lo = mid + 1;
      else hi = mid;
    }
    int64_t b = lo;
    // the utterance that straddles the target goes to whichever side leaves the boundary nearer the
    // target (always taking the first prefix >= target put it in the lower rank: [50,50,50,50,400] over
    // two ranks gave the second one nothing)
    if (b > 0 && b <= n && target - csum[(size_t)b - 1] < csum[(size_t)b] - target) b -= 1;
    if (b < bounds[r - 1]) b = bounds[r - 1];
    if (b > n) b = n;
    bounds[r] = b;
  }
  bounds[world] = n;
  return KLAT_OK;
}

int klat_rccl_unique_id(void* id) {
  if (!id) return set_error(KLAT_ERR_INVALID_ARG, "id is null");
  static_assert(sizeof(ncclUniqueId) <= KLAT_UNIQUE_ID_BYTES, "unique id does not fit");
  ncclUniqueId u;
  D_NCCL(ncclGetUniqueId(&u));
  std::memset(id, 0, KLAT_UNIQUE_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return KLAT_OK;
}

int klat_comm_create_rccl(klat_ctx* ctx, int32_t rank, int32_t world, const void* id, klat_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return set_error(KLAT_ERR_INVALID_ARG, "klat_comm_create_rccl: bad argument");
  *out = nullptr;
  klat_comm* c = new (std::nothrow) klat_comm();
  if (!c) return set_error(KLAT_ERR_OOM, "host allocation failed");
  c->rank = rank;
  c->world = world;
  c->rccl = true;
  c->ctx = ctx;
  klat_ctx_retain(ctx);  // the communicator works on the context's stream
  c->stream = static_cast<hipStream_t>(klat_ctx_stream(ctx));
  c->meta_dev.device = c->records.device = c->counts.device = c->send_counts.device = true;
  hipError_t e = hipGetDevice(&c->device);  // klat_ctx_create left the calling thread on the context's device
  if (e != hipSuccess) {
    delete c;
    klat_ctx_destroy(ctx);
    return hip_err(e, "hipGetDevice");
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    klat_ctx_destroy(ctx);
    return nccl_err(r, "ncclCommInitRank");
  }
  *out = c;
  return KLAT_OK;
}

int klat_comm_create_custom(int32_t rank, int32_t world, const klat_transport* t, klat_comm** out) {
  if (!t || !out || world < 1 || rank < 0 || rank >= world || !t->all_gather_i64 || !t->group_begin || !t->send || !t->recv || !t->group_end)
    return set_error(KLAT_ERR_INVALID_ARG, "klat_comm_create_custom: bad argument");
  *out = nullptr;
  klat_comm* c = new (std::nothrow) klat_comm();
  if (!c) return set_error(KLAT_ERR_OOM, "host allocation failed");
  c->rank = rank;
  c->world = world;
  c->t = *t;
  *out = c;
  return KLAT_OK;
}

int klat_comm_info(const klat_comm* c, int32_t* rank, int32_t* world, int32_t* transport_ranks) {
  if (!c) return set_error(KLAT_ERR_INVALID_ARG, "comm is null");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (transport_ranks) {
    *transport_ranks = c->world;
    if (c->rccl) {
      if (!c->nccl) return set_error(KLAT_ERR_COMM, "klat_comm_info: the communicator was aborted after an earlier failure");
      int n = 0;
      D_NCCL(ncclCommCount(c->nccl, &n));
      *transport_ranks = n;
    }
  }
  return KLAT_OK;
}

void klat_comm_destroy(klat_comm* c) {
  if (!c) return;
  if (c->rccl) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
  }
  klat_ctx* ctx = c->rccl ? c->ctx : nullptr;
  delete c;  // (its device buffers first)
  if (ctx) klat_ctx_destroy(ctx);
}

// A failure that only this rank sees must not leave the others inside the exchange: everything a
// rank can check or allocate on its own happens BEFORE a size all-gather that carries a status word,
// the root's allocation (which needs the sizes) is confirmed by a second one-word all-gather, so all
// ranks leave together; an error between group_begin and group_end still closes the group, and an
// RCCL communicator that failed inside the exchange is aborted (its peers then fail instead of
// waiting) and refuses further use.
static int abort_comm(klat_comm* c, int rc) {
  if (c->rccl && c->nccl) {
    const std::string msg = klat_last_error();  // (keep the first error's text)
    (void)ncclCommAbort(c->nccl);
    c->nccl = nullptr;
    return set_error(rc, msg + " (communicator aborted)");
  }
  return rc;
}

int klat_gather_records(klat_comm* c, const void* records, int32_t record_bytes, const int64_t* record_off,
                        int64_t num_utts, int32_t root, int64_t* total_utts, int64_t* total_records) {
  if (!c || record_bytes <= 0 || num_utts < 0 || (num_utts > 0 && !record_off) || root < 0 || root >= c->world)
    return set_error(KLAT_ERR_INVALID_ARG, "klat_gather_records: bad argument");
  if (c->rccl && !c->nccl) return set_error(KLAT_ERR_COMM, "klat_gather_records: the communicator was aborted after an earlier failure");
  if (c->rccl) D_HIP(hipSetDevice(c->device));
  c->have_result = false;
  const bool is_root = c->rank == root;
  const auto t0 = std::chrono::steady_clock::now();

  // 1. everything local: validation, the per-utterance record counts in transport memory
  int local_rc = KLAT_OK;
  std::string local_msg;
  const int64_t num_records = num_utts > 0 ? record_off[num_utts] - record_off[0] : 0;
  if (num_records < 0 || (num_records > 0 && !records)) local_rc = set_error(KLAT_ERR_INVALID_ARG, "klat_gather_records: bad records");
  std::vector<int64_t> cnt((size_t)num_utts);
  for (int64_t u = 0; u < num_utts && local_rc == KLAT_OK; ++u) {
    cnt[(size_t)u] = record_off[u + 1] - record_off[u];
    if (cnt[(size_t)u] < 0) local_rc = set_error(KLAT_ERR_INVALID_ARG, "klat_gather_records: record_off decreases");
  }
  const void* cnt_src = cnt.data();
  if (local_rc == KLAT_OK && c->rccl && !is_root) {
    if ((local_rc = c->send_counts.reserve(sizeof(int64_t) * (size_t)num_utts)) == KLAT_OK &&
        (local_rc = copy_in(c, c->send_counts.p, cnt.data(), sizeof(int64_t) * (size_t)num_utts, true)) == KLAT_OK) {
      const hipError_t e = hipStreamSynchronize(c->stream);  // `cnt` is pageable host memory
      if (e != hipSuccess) local_rc = hip_err(e, "hipStreamSynchronize");
    }
    cnt_src = c->send_counts.p;
  }
  if (local_rc != KLAT_OK) local_msg = klat_last_error();

  // 2. sizes and status of every rank
  const int64_t mine[4] = {local_rc == KLAT_OK ? num_records : 0, local_rc == KLAT_OK ? num_utts : 0, (int64_t)record_bytes, (int64_t)local_rc};
  std::vector<int64_t> sizes((size_t)c->world * 4);
  int rc = all_gather_i64(c, mine, sizes.data(), 4);
  if (rc != KLAT_OK) return abort_comm(c, rc);
  if (local_rc != KLAT_OK) return set_error(local_rc, local_msg);
  int64_t tot_rec = 0, tot_utt = 0;
  for (int32_t r = 0; r < c->world; ++r) {  // (every rank sees the same table: they all return here or none does)
    if (sizes[(size_t)r * 4 + 3] != 0)
      return set_error(KLAT_ERR_COMM, "klat_gather_records: rank " + std::to_string(r) + " failed before the exchange (code " + std::to_string(sizes[(size_t)r * 4 + 3]) + ")");
    if (sizes[(size_t)r * 4 + 2] != record_bytes) return set_error(KLAT_ERR_INVALID_ARG, "klat_gather_records: ranks disagree on the record size");
    if (sizes[(size_t)r * 4] < 0 || sizes[(size_t)r * 4 + 1] < 0) return set_error(KLAT_ERR_COMM, "klat_gather_records: negative size from a rank");
    tot_rec += sizes[(size_t)r * 4];
    tot_utt += sizes[(size_t)r * 4 + 1];
  }

  // 3. the root's buffers need the sizes: a second status word before anything is posted
  int64_t ready = 0;
  std::string ready_msg;
  if (is_root) {
    int arc = c->records.reserve((size_t)tot_rec * (size_t)record_bytes);
    if (arc == KLAT_OK) arc = c->counts.reserve(sizeof(int64_t) * (size_t)tot_utt);
    if (arc != KLAT_OK) ready = arc, ready_msg = klat_last_error();
  }
  std::vector<int64_t> readies((size_t)c->world);
  rc = all_gather_i64(c, &ready, readies.data(), 1);
  if (rc != KLAT_OK) return abort_comm(c, rc);
  if (readies[(size_t)root] != 0)
    return is_root ? set_error((int)ready, ready_msg) : set_error(KLAT_ERR_COMM, "klat_gather_records: the root could not allocate the gathered records");

  // 4. payload: every peer straight to the root, at its offset in rank order
  if ((rc = group_begin(c))) return abort_comm(c, rc);
  if (is_root) {
    int64_t ro = 0, uo = 0;
    for (int32_t r = 0; r < c->world && rc == KLAT_OK; ++r) {
      const int64_t nr = sizes[(size_t)r * 4], nu = sizes[(size_t)r * 4 + 1];
      unsigned char* rdst = c->records.p + (size_t)ro * (size_t)record_bytes;
      unsigned char* cdst = c->counts.p + sizeof(int64_t) * (size_t)uo;
      if (r == root) {
        if ((rc = copy_in(c, rdst, records, (size_t)nr * (size_t)record_bytes, false)) == KLAT_OK)
          rc = copy_in(c, cdst, cnt.data(), sizeof(int64_t) * (size_t)nu, true);
      } else {
        if ((rc = p2p(c, false, rdst, nr * record_bytes, r)) == KLAT_OK) rc = p2p(c, false, cdst, (int64_t)sizeof(int64_t) * nu, r);
      }
      ro += nr;
      uo += nu;
    }
  } else {
    if ((rc = p2p(c, true, const_cast<void*>(records), num_records * record_bytes, root)) == KLAT_OK)
      rc = p2p(c, true, const_cast<void*>(cnt_src), (int64_t)sizeof(int64_t) * num_utts, root);
  }
  {
    std::string first = rc != KLAT_OK ? klat_last_error() : "";
    const int erc = group_end(c);  // (always paired with group_begin)
    if (rc != KLAT_OK) return abort_comm(c, set_error(rc, first));
    if (erc != KLAT_OK) return abort_comm(c, erc);
  }
  if (c->rccl) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return abort_comm(c, hip_err(e, "hipStreamSynchronize"));
  }
  c->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  c->last_payload = is_root ? (tot_rec - num_records) * record_bytes + (int64_t)sizeof(int64_t) * (tot_utt - num_utts) : 0;
  c->record_bytes = record_bytes;
  c->total_utts = is_root ? tot_utt : 0;
  c->total_records = is_root ? tot_rec : 0;
  c->have_result = is_root;
  if (total_utts) *total_utts = c->total_utts;
  if (total_records) *total_records = c->total_records;
  return KLAT_OK;
}

int klat_gather_index(klat_comm* c, klat_batch* b, int32_t root, int64_t* total_utts, int64_t* total_records) {
  if (!c || !b) return set_error(KLAT_ERR_INVALID_ARG, "klat_gather_index: null argument");
  if (!c->rccl) return set_error(KLAT_ERR_INVALID_ARG, "klat_gather_index needs the RCCL transport: the batch's records are device memory");
  const void* dev = nullptr;
  const int64_t* off = nullptr;
  int64_t n = 0, U = 0;
  const int rc = klat_index_device_records(b, &dev, &off, &U, &n);
  if (rc != KLAT_OK) return rc;
  return klat_gather_records(c, dev, (int32_t)sizeof(klat_segment_record), off, U, root, total_utts, total_records);
}

int klat_gather_fetch(klat_comm* c, int64_t* record_off, void* records, int64_t capacity) {
  if (!c) return set_error(KLAT_ERR_INVALID_ARG, "comm is null");
  if (!c->have_result) return set_error(KLAT_ERR_STATE, "klat_gather_fetch: no gathered result on this rank (root only, after a gather)");
  if (records && capacity < c->total_records) return set_error(KLAT_ERR_INVALID_ARG, "klat_gather_fetch: capacity too small");
  std::vector<int64_t> cnt((size_t)c->total_utts);
  if (c->rccl) {
    D_HIP(hipSetDevice(c->device));
    if (record_off && c->total_utts > 0)
      D_HIP(hipMemcpyAsync(cnt.data(), c->counts.p, sizeof(int64_t) * cnt.size(), hipMemcpyDeviceToHost, c->stream));
    if (records && c->total_records > 0)
      D_HIP(hipMemcpyAsync(records, c->records.p, (size_t)c->total_records * (size_t)c->record_bytes, hipMemcpyDeviceToHost, c->stream));
    D_HIP(hipStreamSynchronize(c->stream));
  } else {
    if (record_off && c->total_utts > 0) std::memcpy(cnt.data(), c->counts.p, sizeof(int64_t) * cnt.size());
    if (records && c->total_records > 0) std::memcpy(records, c->records.p, (size_t)c->total_records * (size_t)c->record_bytes);
  }
  if (record_off) {
    record_off[0] = 0;
    for (int64_t u = 0; u < c->total_utts; ++u) record_off[u + 1] = record_off[u] + cnt[(size_t)u];
    if (record_off[c->total_utts] != c->total_records) return set_error(KLAT_ERR_COMM, "klat_gather_fetch: gathered counts do not add up to the gathered records");
  }
  return KLAT_OK;
}

int klat_gather_stats(const klat_comm* c, double* seconds, int64_t* payload_bytes) {
  if (!c) return set_error(KLAT_ERR_INVALID_ARG, "comm is null");
  if (seconds) *seconds = c->last_seconds;
  if (payload_bytes) *payload_bytes = c->last_payload;
  return KLAT_OK;
}

}  // extern "C"
