// comm.hip -- the one collective of the multi-GPU path: the sum of the mapping statistics over the ranks.
//
// The reference has no distributed mode; its only cross-read state is the statistics block that
// ProcessSingledEndReads / ProcessPairedEndReads accumulate over a run (StatSingleReads, mapping.hpp:94-100;
// StatPairedReads incl. the fragment-length histogram, paired.hpp:96-105).  With one process per GPU, each
// holding an index replica and a contiguous shard of the reads, that block is the only thing
// the ranks exchange: one ncclAllReduce(sum, uint64) over RCCL at the end of a run -- about 8 KB, i.e.
// latency-bound, xGMI bandwidth is irrelevant.
//
// librccl is loaded with dlopen on first use (no link-time dependency: single-GPU users never touch it, and
// a process that also runs PyTorch must not end up with two copies of it -- the SONAME lookup returns the
// copy that is already loaded).
#include <dlfcn.h>
#include <link.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "device_common.h"

namespace walt {

struct RcclApi {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;
static std::string g_rccl_error;

static void load_rccl() {
  const char* names[] = {getenv("WALT_AMD_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  std::string tried;
  // a copy the process holds already (a torch process has loaded its own librccl, under whatever path and name) is
  // taken first, so that two copies of RCCL never live in one process: the loaded objects are searched for "librccl",
  // and RTLD_NOLOAD turns the resident one's path into a handle
  {
    std::string resident;
    dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) -> int {
      if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) { *static_cast<std::string*>(data) = info->dlpi_name; return 1; }
      return 0;
    }, &resident);
    if (!resident.empty()) g_rccl.so = dlopen(resident.c_str(), RTLD_NOW | RTLD_NOLOAD);
  }
  for (const char* nm : names) {
    if (!nm || !*nm || g_rccl.so) continue;
    g_rccl.so = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  }
  for (const char* nm : names) {
    if (!nm || !*nm || g_rccl.so) continue;
    g_rccl.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.so) break;
    const char* why = dlerror();  // may be null
    tried += std::string(tried.empty() ? "" : "; ") + nm + ": " + (why ? why : "unknown error");
  }
  if (!g_rccl.so) {
    g_rccl_error = "cannot load librccl (" + tried + ")";
    return;
  }
  auto sym = [&](const char* s) {
    void* p = dlsym(g_rccl.so, s);
    if (!p && g_rccl_error.empty()) g_rccl_error = std::string("librccl lacks ") + s;
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}

static int rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl_error.empty()) return fail(WALT_EHIP, g_rccl_error);
  return WALT_OK;
}

#define WALT_RCCL(expr)                                                                                   \
  do {                                                                                                    \
    ncclResult_t r_ = (expr);                                                                             \
    if (r_ != ncclSuccess) return walt::fail(WALT_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

}  // namespace walt

using namespace walt;

struct walt_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
  hipStream_t stream = nullptr;
  unsigned long long* d_buf = nullptr;
  size_t cap = 0;
};

extern "C" {

int walt_comm_available(void) { return rccl_ready(); }

int walt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(WALT_EINVAL, "walt_comm_unique_id: null buffer");
  int rc = rccl_ready();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == WALT_COMM_ID_BYTES, "WALT_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  WALT_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return WALT_OK;
}

int walt_comm_init(int device, int rank, int world, const void* id, walt_comm** out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(WALT_EINVAL, "walt_comm_init: bad argument");
  *out = nullptr;
  int rc = rccl_ready();
  if (rc) return rc;
  // everything that can fail LOCALLY comes before the collective call: a rank that returns an error from here has not
  // entered ncclCommInitRank, so its peers are not left blocked inside it by a failure only this rank sees (the caller
  // agrees on the ranks' status before and after: walt_amd/dist.py c_abi_cross_check)
  WALT_HIP(hipSetDevice(device));
  walt_comm* c = new walt_comm();
  c->device = device; c->rank = rank; c->world = world;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    e = hipMalloc(reinterpret_cast<void**>(&c->d_buf), 4096 * sizeof(uint64_t));
    if (e == hipSuccess) c->cap = 4096;
  }
  if (e != hipSuccess) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(WALT_EHIP, std::string("walt_comm_init (local set-up): ") + hipGetErrorString(e));
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    (void)hipFree(c->d_buf);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(WALT_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  *out = c;
  return WALT_OK;
}

int walt_stats_allreduce(walt_comm* c, uint64_t* v, size_t n) {
  if (!v && n) return fail(WALT_EINVAL, "walt_stats_allreduce: null vector");
  if (!c || n == 0) return WALT_OK;  // a single process (walt_comm_init was never needed): the vector already is the total
  WALT_HIP(hipSetDevice(c->device));
  if (c->cap < n) {
    if (c->d_buf) WALT_HIP(hipFree(c->d_buf));
    c->d_buf = nullptr;
    c->cap = 0;
    WALT_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_buf), n * sizeof(uint64_t)));
    c->cap = n;
  }
  WALT_HIP(hipMemcpyAsync(c->d_buf, v, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  WALT_RCCL(g_rccl.AllReduce(c->d_buf, c->d_buf, n, ncclUint64, ncclSum, c->comm, c->stream));
  WALT_HIP(hipMemcpyAsync(v, c->d_buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  WALT_HIP(hipStreamSynchronize(c->stream));
  return WALT_OK;
}

int walt_comm_rank(const walt_comm* c) { return c ? c->rank : 0; }
int walt_comm_world(const walt_comm* c) { return c ? c->world : 1; }

void walt_comm_close(walt_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->d_buf) hipFree(c->d_buf);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
}

}  // extern "C"
