// RCCL all-reduce of the fused synthetic-set gradient buffer, owned by this library (SURVEY 8b
// `allreduce_syn_grads`).  The reference has no counterpart: its only multi-GPU code is nn.DataParallel
// (distill.py:443-445); north_star asks for "RCCL all-reduce of the matching-loss gradient over xGMI".
//
// RCCL is bound at RUN time (dlopen + dlsym), not at link time: the process already holds one RCCL --
// the one libtorch_hip.so was linked against -- and a second copy in the same process would interpose its
// symbols with the first.  RTLD_NOLOAD picks up whichever `librccl.so.1` is already mapped; only a process
// without one (a plain C host) loads /opt/rocm's.  <rccl/rccl.h> is used for the types alone.
// One communicator = one rank = one GPU; the caller moves the 128-byte unique id between ranks (any
// channel: torch.distributed store, MPI, a file), as with ncclGetUniqueId / ncclCommInitRank themselves.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

#include <algorithm>
#include <cstring>

#include "kernels.h"
#include "mdd_hip.h"

#define CHECK_ARG(cond, msg) \
  do { if (!(cond)) return mdd_set_error_msg(2, "mdd: invalid argument: " msg); } while (0)

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
std::once_flag g_once;
Rccl g_rccl;

void bind_rccl() {
  Rccl& r = g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has
  for (const char* n : names)
    if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.handle) { const char* e = dlerror(); r.why = std::string("librccl.so.1 not loadable: ") + (e ? e : "?"); return; }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.handle, name);
    if (!p && r.why.empty()) r.why = std::string("RCCL symbol missing: ") + name;
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
}
// 0 when RCCL is bound; otherwise sets the library error (no fallback: the caller asked for RCCL)
int need_rccl() {
  std::call_once(g_once, bind_rccl);
  if (!g_rccl.why.empty()) return mdd_set_error_msg(5, ("mdd: RCCL unavailable: " + g_rccl.why).c_str());
  return 0;
}
int rccl_fail(ncclResult_t rc, const char* what) {
  std::string m = std::string("mdd: RCCL error in ") + what + ": " +
                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return mdd_set_error_msg(5, m.c_str());
}
#define RCCL_CHECK(expr, what)                       \
  do {                                               \
    ncclResult_t _r = (expr);                        \
    if (_r != ncclSuccess) return rccl_fail(_r, what); \
  } while (0)

struct DeviceScope {      // the calling thread's current device is restored on every exit path
  int prev = -1; bool ok = false;
  explicit DeviceScope(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

__global__ void k_div_inplace(float* __restrict__ x, float d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = x[i] / d;           // a true division: bit-identical to the host path's `flat.div_(world)`
}

}  // namespace

struct mdd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

int mdd_comm_unique_id(void* id_out) {
  CHECK_ARG(id_out, "null pointer");
  if (int rc = need_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == MDD_COMM_ID_BYTES, "MDD_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  RCCL_CHECK(g_rccl.GetUniqueId((ncclUniqueId*)id_out), "ncclGetUniqueId");
  return 0;
}

int mdd_comm_create(const void* id, int rank, int world, int device_id, mdd_comm** out) {
  CHECK_ARG(id && out, "null pointer");
  CHECK_ARG(world >= 1 && rank >= 0 && rank < world && device_id >= 0, "rank / world / device");
  *out = nullptr;
  if (int rc = need_rccl()) return rc;
  DeviceScope ds(device_id);
  if (!ds.ok) return mdd_set_error_msg(2, "mdd: invalid argument: device_id");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  RCCL_CHECK(g_rccl.CommInitRank(&c, world, uid, rank), "ncclCommInitRank");
  int count = 0;
  ncclResult_t rc = g_rccl.CommCount(c, &count);
  if (rc != ncclSuccess || count != world) {
    g_rccl.CommDestroy(c);
    return rc != ncclSuccess ? rccl_fail(rc, "ncclCommCount")
                             : mdd_set_error_msg(5, "mdd: RCCL communicator has a different rank count than asked for");
  }
  auto* m = new mdd_comm();
  m->comm = c; m->rank = rank; m->world = world; m->device = device_id;
  *out = m;
  return 0;
}

void mdd_comm_destroy(mdd_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) {
    DeviceScope ds(c->device);
    g_rccl.CommDestroy(c->comm);
  }
  delete c;
}

int mdd_comm_world(const mdd_comm* c) { return c ? c->world : -1; }

int mdd_allreduce_syn_grads(mdd_comm* c, float* buf_dev, int64_t n, int average, void* stream) {
  CHECK_ARG(c && c->comm && buf_dev && n >= 0, "null pointer");
  if (n == 0) return 0;
  DeviceScope ds(c->device);
  if (!ds.ok) return mdd_set_error_msg(2, "mdd: invalid argument: communicator device");
  hipStream_t st = (hipStream_t)stream;
  RCCL_CHECK(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat32, ncclSum, c->comm, st), "ncclAllReduce");
  if (average && c->world > 1) {
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    k_div_inplace<<<blocks, 256, 0, st>>>(buf_dev, (float)c->world, n);
    HIP_CHECK_RET(hipGetLastError());
  }
  return 0;
}

}  // extern "C"
