// pk_comm.hip -- the ONE collective of the sharded drivers behind the C ABI: an RCCL all-gather over xGMI for embedders that have
// no torch.distributed (SURVEY 8b proposed `pk_allgather_f64`; VERDICT r2 missing #7).  One process per GPU; every rank integrates its rows
// with the batch entry points and gathers the per-row results -- candidates / replicas never move.
//
// RCCL is bound at run time (dlopen), not at link time: the library stays loadable on a box without librccl, and inside a Python process it
// picks up the copy PyTorch already mapped (two RCCL copies in one process would each own their own xGMI state).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "../../include/phoskin.h"

struct pk_ctx;
extern "C" int pk_ctx_device(pk_ctx*);
extern "C" void* pk_ctx_stream(pk_ctx*);
extern "C" int pk_ctx_fail(pk_ctx*, int code, const char* msg);

namespace {

struct rccl_id { char internal[128]; };                  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;
struct rccl_api {
  void* lib = nullptr;
  int (*GetUniqueId)(rccl_id*) = nullptr;
  int (*CommInitRank)(rccl_comm*, int, rccl_id, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
  int (*CommDestroy)(rccl_comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};
constexpr int kRcclFloat64 = 8;                          // ncclFloat64 / ncclDouble

rccl_api& api() {
  static rccl_api a;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) { a.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
    a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) { a.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy"; a.lib = nullptr; }
  });
  return a;
}

struct comm_entry { rccl_comm comm; int rank, world; };
std::mutex g_mu;
std::map<pk_ctx*, comm_entry> g_comms;                   // a context owns at most one communicator

int rccl_fail(pk_ctx* c, const char* what, int rc) {
  const rccl_api& a = api();
  std::string m = std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
  return pk_ctx_fail(c, PK_ERR_HIP, m.c_str());
}

}  // namespace

extern "C" {

int pk_comm_unique_id(pk_ctx* c, char* id_out) {
  if (!c || !id_out) return PK_ERR_ARG;
  rccl_api& a = api();
  if (!a.lib) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, a.why.c_str());
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  rccl_id id;
  const int rc = a.GetUniqueId(&id);
  if (rc != 0) return rccl_fail(c, "ncclGetUniqueId", rc);
  std::memcpy(id_out, id.internal, PK_COMM_ID_BYTES);
  return PK_OK;
}

int pk_comm_init(pk_ctx* c, const char* id_in, int rank, int world) {
  if (!c || !id_in) return PK_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world) return pk_ctx_fail(c, PK_ERR_ARG, "0 <= rank < world required");
  rccl_api& a = api();
  if (!a.lib) return pk_ctx_fail(c, PK_ERR_UNSUPPORTED, a.why.c_str());
  {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_comms.count(c)) return pk_ctx_fail(c, PK_ERR_ARG, "this context already owns a communicator (pk_comm_destroy first)");
  }
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  rccl_id id;
  std::memcpy(id.internal, id_in, PK_COMM_ID_BYTES);
  rccl_comm comm = nullptr;
  const int rc = a.CommInitRank(&comm, world, id, rank);
  if (rc != 0) return rccl_fail(c, "ncclCommInitRank", rc);
  std::lock_guard<std::mutex> g(g_mu);
  g_comms[c] = comm_entry{comm, rank, world};
  return PK_OK;
}

int pk_comm_rank(pk_ctx* c) { std::lock_guard<std::mutex> g(g_mu); auto it = g_comms.find(c); return it == g_comms.end() ? PK_ERR_ARG : it->second.rank; }
int pk_comm_world(pk_ctx* c) { std::lock_guard<std::mutex> g(g_mu); auto it = g_comms.find(c); return it == g_comms.end() ? PK_ERR_ARG : it->second.world; }

int pk_allgather_f64(pk_ctx* c, const double* send, int64_t count, double* recv) {
  if (!c) return PK_ERR_ARG;
  if (count < 0) return pk_ctx_fail(c, PK_ERR_ARG, "count must be >= 0");
  comm_entry e;
  {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_comms.find(c);
    if (it == g_comms.end()) return pk_ctx_fail(c, PK_ERR_ARG, "no communicator on this context (pk_comm_init)");
    e = it->second;
  }
  if (count == 0) return PK_OK;
  if (!send || !recv) return pk_ctx_fail(c, PK_ERR_ARG, "null pointer");
  if (hipSetDevice(pk_ctx_device(c)) != hipSuccess) return pk_ctx_fail(c, PK_ERR_HIP, "hipSetDevice");
  const int rc = api().AllGather(send, recv, (size_t)count, kRcclFloat64, e.comm, (hipStream_t)pk_ctx_stream(c));
  if (rc != 0) return rccl_fail(c, "ncclAllGather", rc);
  return PK_OK;
}

int pk_comm_destroy(pk_ctx* c) {
  if (!c) return PK_ERR_ARG;
  comm_entry e;
  {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_comms.find(c);
    if (it == g_comms.end()) return PK_OK;
    e = it->second;
    g_comms.erase(it);
  }
  (void)hipSetDevice(pk_ctx_device(c));
  (void)hipStreamSynchronize((hipStream_t)pk_ctx_stream(c));
  const int rc = api().CommDestroy(e.comm);
  return rc == 0 ? PK_OK : rccl_fail(c, "ncclCommDestroy", rc);
}

}  // extern "C"
