// pmaf_comm.hpp -- internal view of a communicator (include/pmaf.h: pmaf_comm) shared by pmaf_shard.cpp (which owns
// it) and pmaf_host.cpp (the winner-record exchange).
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>

#include "../../include/pmaf.h"

struct pmaf_comm {
  int world = 1, rank = 0;
  int device = 0;
  bool rccl = false;            // true: nccl_comm is an ncclComm_t; false: host callback
  void *nccl_comm = nullptr;
  bool owns_nccl = false;
  pmaf_host_allgather_fn fn = nullptr;
  void *ctx = nullptr;
  // staging for pmaf_comm_allgather of host data over RCCL
  hipStream_t stream = nullptr;
  double *d_stage = nullptr, *h_stage = nullptr;
  size_t stage_doubles = 0;
};

// RCCL only: enqueue ncclAllGather of n_per_rank doubles (device memory) on stream s. Returns an empty string or the
// error text.
std::string pmaf_comm_enqueue_allgather(pmaf_comm *c, const double *send_dev, double *recv_dev, size_t n_per_rank,
                                        hipStream_t s);
