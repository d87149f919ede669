// comm.hip -- the data-parallel gradient exchange from the C ABI: RCCL reached directly instead of through torch.distributed.
//
// What it replaces: the reference trains under DistributedDataParallel (engine/defaults.py:256, scripts/train_VOC.py:67-77), whose reducer all-reduces
// gradient buckets on NCCL's stream; SURVEY section 8(b) lists `unit_comm_init(rank, world, id)`, `unit_allreduce_bucket_async`, `unit_comm_wait` as the
// comm exports of the drop-in library for a host that does not carry torch.distributed (unit_amd/parallel.py, the Python host of this repository, keeps
// using torch.distributed: same RCCL underneath, plus gloo for the CPU rehearsals).
//
// RCCL is NOT a link-time dependency of libunit_hip.so: the first comm call resolves the five entry points with dlopen -- the copy already in the
// process if there is one (PyTorch loads its own librccl), else the system one -- so a host that never calls these functions never loads it.
// One process per GPU, the bucket is summed IN PLACE on the stream given (the caller's collective stream); `unit_comm_wait` is the event hand-off
// between that stream and the compute stream (the same primitive as unit_stream_wait_stream). No hidden allocation: RCCL's own buffers are its own.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

extern "C" int unit_stream_wait_stream(void* waiter, void* signaller);          // multi.hip

namespace {
struct NcclId { char internal[128]; };                 // ncclUniqueId (nccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void* ncclComm_t;
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(ncclComm_t*, int, NcclId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*fn_destroy)(ncclComm_t);
typedef const char* (*fn_err)(int);
typedef int (*fn_version)(int*);

struct Rccl {
  void* h = nullptr;
  fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_all_reduce all_reduce = nullptr; fn_destroy destroy = nullptr;
  fn_err err = nullptr; fn_version version = nullptr;
};
Rccl g_rccl;

int rccl_load_once() {
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }          // the copy the host process already runs on
  if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { unit_set_error("unit_comm: librccl.so not found (dlopen)"); return UNIT_ERR_UNSUPPORTED; }
  Rccl r;
  r.h = h;
  r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  r.err = (fn_err)dlsym(h, "ncclGetErrorString");
  r.version = (fn_version)dlsym(h, "ncclGetVersion");
  if (!r.get_id || !r.init_rank || !r.all_reduce || !r.destroy) { unit_set_error("unit_comm: librccl.so lacks the NCCL entry points"); return UNIT_ERR_UNSUPPORTED; }
  g_rccl = r;
  return UNIT_OK;
}

// two host threads may make their first comm call at the same time: the table is filled exactly once, later callers read a complete
// table (a failed load is remembered with its message re-set for every caller) -- ADVICE r05
int rccl_load() {
  static std::once_flag once;
  static int status = UNIT_ERR_UNSUPPORTED;
  std::call_once(once, [] { status = rccl_load_once(); });
  if (status != UNIT_OK) unit_set_error("unit_comm: librccl.so could not be loaded (dlopen) or lacks the NCCL entry points");
  return status;
}

int rccl_fail(int rc, const char* what) {
  thread_local char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, g_rccl.err ? g_rccl.err(rc) : "RCCL error");
  unit_set_error(buf);
  return UNIT_ERR_LAUNCH;
}
}  // namespace

// rank 0 fills `id` (UNIT_COMM_ID_BYTES = 128 bytes) and hands it to the other ranks by whatever channel the host has (a file, MPI, a socket)
extern "C" int unit_comm_unique_id(void* id, int id_bytes) {
  UNIT_CHECK_ARG(id != nullptr && id_bytes >= (int)sizeof(NcclId), "unit_comm_unique_id: a buffer of at least 128 bytes");
  int rc = rccl_load();
  if (rc != UNIT_OK) return rc;
  NcclId u;
  int e = g_rccl.get_id(&u);
  if (e != 0) return rccl_fail(e, "ncclGetUniqueId");
  memcpy(id, &u, sizeof(u));
  return UNIT_OK;
}

// every rank, on its device (hipSetDevice before the call): *comm = an opaque handle for the calls below
extern "C" int unit_comm_init(int rank, int world, const void* id, int id_bytes, void** comm) {
  UNIT_CHECK_ARG(comm != nullptr && id != nullptr && id_bytes >= (int)sizeof(NcclId) && world >= 1 && rank >= 0 && rank < world,
                 "unit_comm_init: 0 <= rank < world, the 128-byte id of unit_comm_unique_id");
  int rc = rccl_load();
  if (rc != UNIT_OK) return rc;
  NcclId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  int e = g_rccl.init_rank(&c, world, u, rank);
  if (e != 0) return rccl_fail(e, "ncclCommInitRank");
  *comm = c;
  return UNIT_OK;
}

// buf[0 .. count) <- sum over the ranks, in place, enqueued on `stream` (asynchronous: returns when the collective is queued). dtype UNIT_F32 or UNIT_BF16
// (bf16 buckets: unit_amd/parallel.py bf16_buckets). The 1 / world of the mean is folded into the optimizer's gradient scale, as in the Python host.
extern "C" int unit_allreduce_bucket_async(void* comm, void* buf, long count, int dtype, void* stream) {
  UNIT_CHECK_ARG(comm != nullptr && (buf != nullptr || count == 0) && count >= 0, "unit_allreduce_bucket_async: a communicator and a bucket");
  UNIT_CHECK_ARG(dtype == UNIT_F32 || dtype == UNIT_BF16, "unit_allreduce_bucket_async: fp32 or bf16 buckets");
  if (count == 0) return UNIT_OK;
  if (!g_rccl.h) { unit_set_error("unit_allreduce_bucket_async: unit_comm_init first"); return UNIT_ERR_ARG; }
  const int nccl_dtype = dtype == UNIT_F32 ? 7 : 9;           // ncclFloat32 / ncclBfloat16
  int e = g_rccl.all_reduce(buf, buf, (size_t)count, nccl_dtype, 0 /* ncclSum */, (ncclComm_t)comm, (hipStream_t)stream);
  if (e != 0) return rccl_fail(e, "ncclAllReduce");
  return UNIT_OK;
}

// everything enqueued on `compute_stream` after this call waits for the collectives enqueued on `comm_stream` before it (event hand-off; no host wait)
extern "C" int unit_comm_wait(void* compute_stream, void* comm_stream) { return unit_stream_wait_stream(compute_stream, comm_stream); }

extern "C" int unit_comm_destroy(void* comm) {
  if (comm == nullptr) return UNIT_OK;
  if (!g_rccl.h) { unit_set_error("unit_comm_destroy: no communicator was created"); return UNIT_ERR_ARG; }
  int e = g_rccl.destroy((ncclComm_t)comm);
  if (e != 0) return rccl_fail(e, "ncclCommDestroy");
  return UNIT_OK;
}

// RCCL's version code (major * 10000 + minor * 100 + patch) of the library the calls above resolved to; 0 = not loadable
extern "C" int unit_comm_rccl_version(void) {
  if (rccl_load() != UNIT_OK || !g_rccl.version) return 0;
  int v = 0;
  return g_rccl.version(&v) == 0 ? v : 0;
}
