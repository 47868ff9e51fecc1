// RCCL binding of the collective layer (comm.h).  The six RCCL entry points used are looked up with dlsym in
// whatever librccl the process already holds (e.g. the one a PyTorch-ROCm wheel brought along) or, failing that,
// the system one: mixing two copies of RCCL -- or RCCL built against another HIP runtime than the one in the
// process -- is what the lookup order avoids.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace lsspa {

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;        // optional: what RCCL itself reports
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  std::string load_error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  // RTLD_NOLOAD first: a copy already in the process wins (same HIP runtime as the rest of the process)
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
  for (const char* n : names)
    if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) g_rccl.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) {
    const char* e = dlerror();
    g_rccl.load_error = std::string("cannot load librccl: ") + (e ? e : "unknown error");
    return;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(g_rccl.handle, name);
    if (!p && g_rccl.load_error.empty()) g_rccl.load_error = std::string("librccl lacks ") + name;
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(g_rccl.handle, "ncclCommCount"));
  g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(g_rccl.handle, "ncclCommUserRank"));
}

bool rccl_ready(std::string& err) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.load_error.empty()) {
    err = g_rccl.load_error;
    return false;
  }
  return true;
}

int check(ncclResult_t r, const char* what, std::string& err) {
  if (r == ncclSuccess) return 0;
  err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return 1;
}

}  // namespace

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

int comm_unique_id(uint8_t* out128, std::string& err) {
  static_assert(sizeof(ncclUniqueId) == COMM_ID_BYTES, "unique id size");
  if (!out128) {
    err = "NULL id buffer";
    return 1;
  }
  if (!rccl_ready(err)) return 1;
  ncclUniqueId id;
  if (check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId", err)) return 1;
  std::memcpy(out128, id.internal, COMM_ID_BYTES);
  return 0;
}

int comm_create(const uint8_t* id128, int rank, int world, int device, Comm** out, std::string& err) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
    err = "communicator arguments: need 0 <= rank < world and an id";
    return 1;
  }
  if (!rccl_ready(err)) return 1;
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) {
    err = std::string("hipSetDevice: ") + hipGetErrorString(he);
    return 1;
  }
  ncclUniqueId id;
  std::memcpy(id.internal, id128, COMM_ID_BYTES);
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  if (check(g_rccl.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank", err)) {
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

// rank / size as the RCCL communicator reports them (the numbers it was made with are the fallback for a
// librccl without the two queries): "did RCCL see N ranks" is answered by RCCL, not by our own bookkeeping
int comm_rank(const Comm* c) {
  if (!c) return 0;
  int r = c->rank;
  if (c->comm && g_rccl.CommUserRank && g_rccl.CommUserRank(c->comm, &r) != ncclSuccess) r = c->rank;
  return r;
}
int comm_world(const Comm* c) {
  if (!c) return 1;
  int n = c->world;
  if (c->comm && g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) != ncclSuccess) n = c->world;
  return n;
}

int comm_allreduce_f64(Comm* c, double* buf, size_t count, hipStream_t st, std::string& err) {
  if (!c || !buf) {
    err = "all-reduce without a communicator or a buffer";
    return 1;
  }
  if (count == 0) return 0;
  return check(g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, st), "ncclAllReduce(f64)", err);
}

int comm_allreduce_i64(Comm* c, int64_t* buf, size_t count, hipStream_t st, std::string& err) {
  if (!c || !buf) {
    err = "all-reduce without a communicator or a buffer";
    return 1;
  }
  if (count == 0) return 0;
  return check(g_rccl.AllReduce(buf, buf, count, ncclInt64, ncclSum, c->comm, st), "ncclAllReduce(i64)", err);
}

int comm_allgather_f64(Comm* c, const double* send, double* recv, size_t count, hipStream_t st, std::string& err) {
  if (!c || !send || !recv) {
    err = "all-gather without a communicator or a buffer";
    return 1;
  }
  if (count == 0) return 0;
  return check(g_rccl.AllGather(send, recv, count, ncclDouble, c->comm, st), "ncclAllGather(f64)", err);
}

}  // namespace lsspa
