// Collective layer of the LS-SPA engine: RCCL (all-reduce / all-gather over xGMI) behind a handful of calls.
// librccl is resolved at run time (dlopen) the first time a communicator is made, so a single-GPU process
// never loads it and the library carries no link-time dependency on it.
//
// What it replaces in the reference: nothing -- cvxgrp/ls-spa is single-process.  The exchange it implements is
// the multi-device form of merge_sample_mean / merge_sample_cov (ls_spa/ls_spa.py:103-119, :212-216): one SUM
// all-reduce of the packed batch moments per chunk of orderings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace lsspa {

struct Comm;   // one RCCL communicator bound to one device

constexpr int COMM_ID_BYTES = 128;   // == NCCL_UNIQUE_ID_BYTES

// status: 0 ok, otherwise a message in err
int comm_unique_id(uint8_t* out128, std::string& err);
int comm_create(const uint8_t* id128, int rank, int world, int device, Comm** out, std::string& err);
void comm_destroy(Comm* c);
int comm_rank(const Comm* c);
int comm_world(const Comm* c);
// in-place SUM all-reduce of `count` elements resident on the communicator's device, enqueued on st
int comm_allreduce_f64(Comm* c, double* buf, size_t count, hipStream_t st, std::string& err);
int comm_allreduce_i64(Comm* c, int64_t* buf, size_t count, hipStream_t st, std::string& err);
// recv [world][count] <- every rank's send [count]
int comm_allgather_f64(Comm* c, const double* send, double* recv, size_t count, hipStream_t st, std::string& err);

}  // namespace lsspa
