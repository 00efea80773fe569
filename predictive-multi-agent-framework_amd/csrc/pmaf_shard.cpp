// pmaf_shard.cpp -- communicators of the multi-GPU path (include/pmaf.h "multi-GPU"): one process per GPU,
// populations sharded over ranks, the fixed-size winner records exchanged with ONE all-gather per tick
// (SURVEY.md 8e). Two transports behind one handle:
//   * RCCL: ncclAllGather over xGMI on device buffers, enqueued on a HIP stream (the product path between GPUs);
//   * host callback: the caller's CPU all-gather (MPI, gloo, tests), staged through pinned host memory.
// The exchange itself (what is packed, when it is enqueued) lives in pmaf_host.cpp.
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "pmaf_comm.hpp"

extern void pmaf_set_last_error(const std::string &msg);  // pmaf_host.cpp (thread-local message of pmaf_last_error)

static_assert(PMAF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pmaf.h: PMAF_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");

namespace {
int fail(int code, const std::string &msg) {
  pmaf_set_last_error(msg);
  return code;
}
std::string nccl_text(const char *what, ncclResult_t r) {
  char buf[384];
  snprintf(buf, sizeof(buf), "%s: RCCL error %d (%s)", what, (int)r, ncclGetErrorString(r));
  return buf;
}
std::string hip_text(const char *what, hipError_t e) {
  char buf[384];
  snprintf(buf, sizeof(buf), "%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return buf;
}
}  // namespace

std::string pmaf_comm_enqueue_allgather(pmaf_comm *c, const double *send_dev, double *recv_dev, size_t n_per_rank,
                                        hipStream_t s) {
  if (!c || !c->rccl) return "pmaf_comm_enqueue_allgather: not an RCCL communicator";
  ncclResult_t r = ncclAllGather(send_dev, recv_dev, n_per_rank, ncclDouble, (ncclComm_t)c->nccl_comm, s);
  if (r != ncclSuccess) return nccl_text("ncclAllGather", r);
  return std::string();
}

extern "C" {

int pmaf_comm_unique_id(void *id_out) {
  if (!id_out) return fail(PMAF_ERR_INVALID, "pmaf_comm_unique_id: NULL argument");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(PMAF_ERR_DEVICE, nccl_text("ncclGetUniqueId", r));
  std::memcpy(id_out, &id, sizeof(id));
  return PMAF_OK;
}

static int finish_rccl_comm(pmaf_comm *c, pmaf_comm **out) {
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    if (c->owns_nccl && c->nccl_comm) ncclCommDestroy((ncclComm_t)c->nccl_comm);
    delete c;
    return fail(PMAF_ERR_DEVICE, hip_text("pmaf_comm (stream)", e));
  }
  *out = c;
  return PMAF_OK;
}

int pmaf_comm_init_rccl(int32_t world, int32_t rank, const void *id, int32_t device, pmaf_comm **out) {
  if (!id || !out) return fail(PMAF_ERR_INVALID, "pmaf_comm_init_rccl: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(PMAF_ERR_INVALID, "pmaf_comm_init_rccl: need 0 <= rank < world");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PMAF_ERR_DEVICE, "pmaf_comm_init_rccl: no HIP device available");
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(PMAF_ERR_DEVICE, "pmaf_comm_init_rccl: hipGetDevice failed");
  if (dev >= ndev) return fail(PMAF_ERR_INVALID, "pmaf_comm_init_rccl: device ordinal out of range");
  hipError_t e = hipSetDevice(dev);
  if (e != hipSuccess) return fail(PMAF_ERR_DEVICE, hip_text("pmaf_comm_init_rccl (hipSetDevice)", e));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, world, uid, rank);
  if (r != ncclSuccess) return fail(PMAF_ERR_DEVICE, nccl_text("ncclCommInitRank", r));
  // what RCCL itself says it joined (reported by pmaf_comm_world, bench.py's config.collective_world): a communicator
  // that came up with another rank count than asked for must not pass as an N-rank run
  int n_rccl = 0, r_rccl = -1;
  r = ncclCommCount(comm, &n_rccl);
  if (r == ncclSuccess) r = ncclCommUserRank(comm, &r_rccl);
  if (r != ncclSuccess || n_rccl != world || r_rccl != rank) {
    ncclCommDestroy(comm);
    if (r != ncclSuccess) return fail(PMAF_ERR_DEVICE, nccl_text("ncclCommCount / ncclCommUserRank", r));
    return fail(PMAF_ERR_DEVICE, "pmaf_comm_init_rccl: RCCL reports " + std::to_string(n_rccl) + " rank(s), this one as rank " +
                                     std::to_string(r_rccl) + "; asked for rank " + std::to_string(rank) + " of " + std::to_string(world));
  }
  pmaf_comm *c = new pmaf_comm();
  c->world = n_rccl; c->rank = r_rccl; c->device = dev;
  c->rccl = true; c->nccl_comm = comm; c->owns_nccl = true;
  return finish_rccl_comm(c, out);
}

int pmaf_comm_from_rccl(void *nccl_comm, int32_t device, pmaf_comm **out) {
  if (!nccl_comm || !out) return fail(PMAF_ERR_INVALID, "pmaf_comm_from_rccl: NULL argument");
  int world = 0, rank = 0, dev = device;
  ncclResult_t r = ncclCommCount((ncclComm_t)nccl_comm, &world);
  if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)nccl_comm, &rank);
  if (r == ncclSuccess && dev < 0) r = ncclCommCuDevice((ncclComm_t)nccl_comm, &dev);
  if (r != ncclSuccess) return fail(PMAF_ERR_DEVICE, nccl_text("pmaf_comm_from_rccl", r));
  pmaf_comm *c = new pmaf_comm();
  c->world = world; c->rank = rank; c->device = dev;
  c->rccl = true; c->nccl_comm = nccl_comm; c->owns_nccl = false;
  return finish_rccl_comm(c, out);
}

int pmaf_comm_init_host(int32_t world, int32_t rank, pmaf_host_allgather_fn fn, void *ctx, pmaf_comm **out) {
  if (!fn || !out) return fail(PMAF_ERR_INVALID, "pmaf_comm_init_host: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(PMAF_ERR_INVALID, "pmaf_comm_init_host: need 0 <= rank < world");
  pmaf_comm *c = new pmaf_comm();
  c->world = world; c->rank = rank; c->device = -1;
  c->rccl = false; c->fn = fn; c->ctx = ctx;
  *out = c;
  return PMAF_OK;
}

int pmaf_comm_destroy(pmaf_comm *c) {
  if (!c) return PMAF_OK;
  if (c->rccl) {
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->owns_nccl && c->nccl_comm) ncclCommDestroy((ncclComm_t)c->nccl_comm);
  }
  delete c;
  return PMAF_OK;
}

int pmaf_comm_world(const pmaf_comm *c) { return c ? c->world : 0; }
int pmaf_comm_rank(const pmaf_comm *c) { return c ? c->rank : -1; }

int pmaf_comm_allgather(pmaf_comm *c, const double *send, double *recv, size_t n_per_rank) {
  if (!c || !send || !recv) return fail(PMAF_ERR_INVALID, "pmaf_comm_allgather: NULL argument");
  if (n_per_rank == 0) return PMAF_OK;
  if (!c->rccl) {
    if (c->fn(c->ctx, send, recv, n_per_rank * sizeof(double)) != 0)
      return fail(PMAF_ERR_DEVICE, "pmaf_comm_allgather: the host all-gather callback failed");
    return PMAF_OK;
  }
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(PMAF_ERR_DEVICE, hip_text("pmaf_comm_allgather (hipSetDevice)", e));
  const size_t need = n_per_rank * (size_t)(c->world + 1);
  if (need > c->stage_doubles) {
    if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; }
    if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; }
    c->stage_doubles = 0;
    const size_t cap = need < 1024 ? 1024 : need * 2;
    e = hipMalloc((void **)&c->d_stage, cap * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_stage, cap * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) return fail(PMAF_ERR_NOMEM, hip_text("pmaf_comm_allgather (staging)", e));
    c->stage_doubles = cap;
  }
  std::memcpy(c->h_stage, send, n_per_rank * sizeof(double));
  double *d_send = c->d_stage, *d_recv = c->d_stage + n_per_rank;
  e = hipMemcpyAsync(d_send, c->h_stage, n_per_rank * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail(PMAF_ERR_DEVICE, hip_text("pmaf_comm_allgather (H2D)", e));
  const std::string err = pmaf_comm_enqueue_allgather(c, d_send, d_recv, n_per_rank, c->stream);
  if (!err.empty()) return fail(PMAF_ERR_DEVICE, err);
  e = hipMemcpyAsync(c->h_stage + n_per_rank, d_recv, n_per_rank * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(PMAF_ERR_DEVICE, hip_text("pmaf_comm_allgather (D2H)", e));
  std::memcpy(recv, c->h_stage + n_per_rank, n_per_rank * c->world * sizeof(double));
  return PMAF_OK;
}

// CfManager::evaluateAgents, B/src/cf_manager.cpp:336-353: first minimum over the costs (strict <, lowest index wins
// ties), then switch only if min < 0.9 * cost[previous best]; the same rule k_manager applies on the device
int32_t pmaf_select_best(const double *costs, int32_t n, int32_t prev_best) {
  if (!costs || n <= 0) return -1;
  double min_cost = 1.7976931348623157e308;
  int32_t min_idx = 0;
  for (int32_t i = 0; i < n; i++)
    if (costs[i] < min_cost) { min_cost = costs[i]; min_idx = i; }
  if (prev_best >= 0 && prev_best < n) {
    if (costs[min_idx] < 0.9 * costs[prev_best]) return min_idx;
    return prev_best;
  }
  return min_idx;
}

}  // extern "C"
