// comm.hip — the ONE exchange step of the path behind the C ABI: RCCL over xGMI, one process per GPU.
//
// Every kernel of this library is independent per grid cell, so a multi-GPU run shards the cell axis and only the
// reduced outputs — (P, C/N) counts / statistics, (nq, C/N) quantile nodes — are exchanged, once, with an all-gather.
// The reference has no collectives: there is no reference line to match here, only north_star's
// "RCCL over xGMI only for the final gather".
//
// librccl.so is dlopen'ed at the first xh_comm_* call: single-GPU users never load it and libxclimhip.so carries no link
// dependency on it.  Rendezvous is the caller's business: rank 0 calls xh_comm_unique_id and hands the 128 bytes to the
// other ranks (xclim_amd/shard.py does it through a file in a node-local directory); xh_comm_init is collective.
//
// Overlap: xh_comm_allgather(..., slot >= 0) runs on the communicator's own stream after everything queued on the
// context's stream so far and records the slot's completion event; the kernels of the next step keep running on the
// context's stream.  xh_comm_fence(slot) makes the context's stream wait for that slot's collective (before the send
// buffer of the slot is overwritten), xh_comm_sync waits on the host.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <unistd.h>

#include "common.h"

#define XH_COMM_SLOTS 4

struct xh_comm {
  xh_ctx* ctx;
  ncclComm_t comm;
  int nranks, rank;
  hipStream_t stream;               // collectives that overlap the context's kernels
  hipEvent_t ev_in;                 // "inputs ready" (recorded on the context's stream)
  hipEvent_t ev_done[XH_COMM_SLOTS];
  double* d_word;                   // one device word for barriers / scalar reductions
};

namespace {

struct RcclApi {
  void* handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
};
RcclApi g_rccl = {};

int load_rccl() {
  if (g_rccl.handle) return XH_OK;
  const char* names[] = {getenv("XH_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  XH_REQUIRE(h != nullptr, XH_ERR_NODEVICE, "xh_comm: librccl.so not found (%s); set XH_RCCL_LIBRARY", dlerror());
  RcclApi a = {};
  a.handle = h;
#define XH_SYM(field, name)                                                                             \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                        \
  XH_REQUIRE(a.field != nullptr, XH_ERR_NODEVICE, "xh_comm: symbol %s missing from librccl", name)
  XH_SYM(GetUniqueId, "ncclGetUniqueId");
  XH_SYM(CommInitRank, "ncclCommInitRank");
  XH_SYM(CommDestroy, "ncclCommDestroy");
  XH_SYM(AllGather, "ncclAllGather");
  XH_SYM(AllReduce, "ncclAllReduce");
  XH_SYM(GetErrorString, "ncclGetErrorString");
#undef XH_SYM
  g_rccl = a;
  return XH_OK;
}

#define XH_CHECK_RCCL(expr)                                                                             \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      xh_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__);      \
      return XH_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

}  // namespace

extern "C" {

int xh_comm_unique_id(void* id) {
  XH_REQUIRE(id != nullptr, XH_ERR_ARG, "xh_comm_unique_id: id is NULL");
  static_assert(sizeof(ncclUniqueId) == XH_COMM_ID_BYTES, "XH_COMM_ID_BYTES must match ncclUniqueId");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId u;
  XH_CHECK_RCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return XH_OK;
}

int xh_comm_init(xh_ctx* ctx, int nranks, int rank, const void* id, xh_comm** out) {
  XH_REQUIRE(ctx && id && out, XH_ERR_ARG, "xh_comm_init: NULL argument");
  XH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, XH_ERR_ARG, "xh_comm_init: rank %d outside [0, %d)", rank, nranks);
  int rc = load_rccl();
  if (rc) return rc;
  XH_CHECK_HIP(hipSetDevice(ctx->device));
  xh_comm* c = new xh_comm();
  memset(c, 0, sizeof(*c));
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  // RCCL prints a version banner on STDOUT when the first communicator comes up; a caller that writes its result there
  // (bench.py: one JSON line, parsed by whoever launched it) would find the banner after it once the C buffer is flushed
  // at exit.  File descriptor 1 points at stderr for the duration of the call.
  fflush(stdout);
  const int saved_out = dup(1);
  if (saved_out >= 0) (void)dup2(2, 1);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
  fflush(stdout);
  if (saved_out >= 0) {
    (void)dup2(saved_out, 1);
    (void)close(saved_out);
  }
  if (r != ncclSuccess) {
    xh_set_error("ncclCommInitRank(nranks=%d, rank=%d) failed: %s", nranks, rank, g_rccl.GetErrorString(r));
    delete c;
    return XH_ERR_HIP;
  }
  // resources of the communicator; on any failure everything created so far — the RCCL communicator included — is
  // released again (xh_comm_destroy tolerates the members that are still zero)
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  for (int i = 0; i < XH_COMM_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_word, 2 * sizeof(double));
  if (e != hipSuccess) {
    xh_set_error("xh_comm_init: %s while creating the stream / events / scratch of the communicator", hipGetErrorString(e));
    (void)hipGetLastError();
    (void)xh_comm_destroy(c);
    return XH_ERR_HIP;
  }
  *out = c;
  return XH_OK;
}

int xh_comm_destroy(xh_comm* c) {
  if (!c) return XH_OK;
  (void)hipSetDevice(c->ctx->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  for (int i = 0; i < XH_COMM_SLOTS; ++i)
    if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_word) (void)hipFree(c->d_word);
  delete c;
  return XH_OK;
}

int xh_comm_size(xh_comm* c, int* nranks, int* rank) {
  XH_REQUIRE(c && nranks && rank, XH_ERR_ARG, "xh_comm_size: NULL argument");
  *nranks = c->nranks;
  *rank = c->rank;
  return XH_OK;
}

int xh_comm_allgather(xh_comm* c, const void* send, void* recv, size_t bytes_per_rank, int slot) {
  XH_REQUIRE(c && send && recv, XH_ERR_ARG, "xh_comm_allgather: NULL argument");
  XH_REQUIRE(slot < XH_COMM_SLOTS, XH_ERR_ARG, "xh_comm_allgather: slot %d outside [-1, %d)", slot, XH_COMM_SLOTS);
  if (bytes_per_rank == 0) return XH_OK;
  if (slot < 0) {  // in line with the kernels, on the context's stream
    XH_CHECK_RCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, c->ctx->stream));
    return XH_OK;
  }
  XH_CHECK_HIP(hipEventRecord(c->ev_in, c->ctx->stream));
  XH_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  XH_CHECK_RCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, c->stream));
  XH_CHECK_HIP(hipEventRecord(c->ev_done[slot], c->stream));
  return XH_OK;
}

int xh_comm_fence(xh_comm* c, int slot) {
  XH_REQUIRE(c && slot >= 0 && slot < XH_COMM_SLOTS, XH_ERR_ARG, "xh_comm_fence: bad arguments");
  XH_CHECK_HIP(hipStreamWaitEvent(c->ctx->stream, c->ev_done[slot], 0));
  return XH_OK;
}

int xh_comm_sync(xh_comm* c) {
  XH_REQUIRE(c != nullptr, XH_ERR_ARG, "xh_comm_sync: comm is NULL");
  XH_CHECK_HIP(hipStreamSynchronize(c->stream));
  XH_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
  return XH_OK;
}

int xh_comm_allreduce_f64(xh_comm* c, double* host_values, int count, int op) {
  XH_REQUIRE(c && host_values, XH_ERR_ARG, "xh_comm_allreduce_f64: NULL argument");
  XH_REQUIRE(count >= 1 && count <= 2, XH_ERR_ARG, "xh_comm_allreduce_f64: count must be 1 or 2 (scalars: timings, checksums)");
  XH_REQUIRE(op == XH_COMM_SUM || op == XH_COMM_MAX || op == XH_COMM_MIN, XH_ERR_OP, "xh_comm_allreduce_f64: op %d not recognized", op);
  const ncclRedOp_t rop = op == XH_COMM_SUM ? ncclSum : (op == XH_COMM_MAX ? ncclMax : ncclMin);
  hipStream_t s = c->ctx->stream;
  XH_CHECK_HIP(hipMemcpyAsync(c->d_word, host_values, sizeof(double) * count, hipMemcpyHostToDevice, s));
  XH_CHECK_RCCL(g_rccl.AllReduce(c->d_word, c->d_word, (size_t)count, ncclFloat64, rop, c->comm, s));
  XH_CHECK_HIP(hipMemcpyAsync(host_values, c->d_word, sizeof(double) * count, hipMemcpyDeviceToHost, s));
  XH_CHECK_HIP(hipStreamSynchronize(s));
  return XH_OK;
}

int xh_comm_barrier(xh_comm* c) {
  XH_REQUIRE(c != nullptr, XH_ERR_ARG, "xh_comm_barrier: comm is NULL");
  XH_CHECK_HIP(hipStreamSynchronize(c->stream));
  double one = 1.0;
  return xh_comm_allreduce_f64(c, &one, 1, XH_COMM_SUM);
}

}  // extern "C"
