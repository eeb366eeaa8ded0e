// dexr_comm.hip -- the one exchange step of the path: reassembling the qpos tensor across the GPUs of a node
// (BASELINE.json north_star: "RCCL all-gather over xGMI only to reassemble the qpos tensor"; SURVEY.md section 8b
// `dexr_allgather`, 8e).  The reference has no distributed mode, so nothing on its side corresponds.
//
// RCCL is bound at run time (dlopen) -- libdexr.so has no link-time dependency on it, a single-GPU deployment never loads
// it, and a process that already carries an RCCL (torch-ROCm bundles one) keeps using THAT copy: two RCCL / HIP runtimes
// in one process do not share streams.  Collectives are enqueued on the caller's stream like every other "_dev" entry
// point: solve -> all-gather is stream-ordered with no host round trip and can be captured into one hipGraph.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; the symbols are resolved with dlsym below

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "dexr.h"

int dexr_set_error(int code, const char* fmt, ...);  // dexr_api.hip

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  std::string where;
};

Rccl* g_rccl = nullptr;
std::mutex g_mu;

Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl) return g_rccl;
  Rccl r;
  // 1. a copy already mapped into the process (torch's), 2. the loader's search path, 3. the ROCm install
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (r.so) { r.where = std::string(n) + " (already loaded)"; break; }
  }
  for (int i = 0; i < 3 && !r.so; ++i) {
    r.so = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (r.so) r.where = names[i];
  }
  if (!r.so) {
    dexr_set_error(DEXR_ERR_UNSUPPORTED, "librccl not found: %s", dlerror());
    return nullptr;
  }
#define DEXR_SYM(name)                                                                            \
  r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.so, "nccl" #name));                         \
  if (!r.name) {                                                                                  \
    dexr_set_error(DEXR_ERR_UNSUPPORTED, "%s lacks nccl" #name, r.where.c_str());                 \
    return nullptr;                                                                               \
  }
  DEXR_SYM(GetVersion)
  DEXR_SYM(GetUniqueId)
  DEXR_SYM(CommInitRank)
  DEXR_SYM(CommDestroy)
  DEXR_SYM(GetErrorString)
  DEXR_SYM(AllGather)
  DEXR_SYM(AllReduce)
#undef DEXR_SYM
  g_rccl = new Rccl(r);
  return g_rccl;
}

}  // namespace

struct dexr_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  double* d_scratch = nullptr;  // 8 doubles for the control-plane reductions (barrier, max)
  double* h_scratch = nullptr;  // pinned mirror
};

#define RCCL_TRY(r, expr)                                                                                          \
  do {                                                                                                             \
    ncclResult_t e_ = (expr);                                                                                      \
    if (e_ != ncclSuccess) return dexr_set_error(DEXR_ERR_HIP, "%s failed: %s", #expr, (r)->GetErrorString(e_));   \
  } while (0)
#define HIPC_TRY(expr)                                                                                             \
  do {                                                                                                             \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess) return dexr_set_error(DEXR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
  } while (0)

extern "C" {

int dexr_comm_unique_id(void* id_out) {
  if (!id_out) return dexr_set_error(DEXR_ERR_INVALID, "null argument");
  Rccl* r = rccl();
  if (!r) return DEXR_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == DEXR_UNIQUE_ID_BYTES, "dexr.h: DEXR_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  RCCL_TRY(r, r->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return DEXR_OK;
}

int dexr_comm_create(const void* unique_id, int32_t rank, int32_t world, dexr_comm** out) {
  if (!unique_id || !out) return dexr_set_error(DEXR_ERR_INVALID, "null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return dexr_set_error(DEXR_ERR_INVALID, "rank %d outside world of %d", rank, world);
  Rccl* r = rccl();
  if (!r) return DEXR_ERR_UNSUPPORTED;
  dexr_comm* c = new (std::nothrow) dexr_comm();
  if (!c) return dexr_set_error(DEXR_ERR_INVALID, "out of host memory");
  c->rank = rank;
  c->world = world;
  hipError_t he = hipGetDevice(&c->device);
  if (he == hipSuccess) he = hipMalloc((void**)&c->d_scratch, 8 * sizeof(double));
  if (he == hipSuccess) he = hipHostMalloc((void**)&c->h_scratch, 8 * sizeof(double), hipHostMallocDefault);
  if (he != hipSuccess) {
    dexr_comm_destroy(c);
    return dexr_set_error(DEXR_ERR_HIP, "communicator scratch allocation failed: %s", hipGetErrorString(he));
  }
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    c->comm = nullptr;
    dexr_comm_destroy(c);
    return dexr_set_error(DEXR_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r->GetErrorString(e));
  }
  *out = c;
  return DEXR_OK;
}

void dexr_comm_destroy(dexr_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl) (void)g_rccl->CommDestroy(c->comm);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->h_scratch) (void)hipHostFree(c->h_scratch);
  delete c;
}

int dexr_comm_info(const dexr_comm* c, int32_t* rank, int32_t* world, int32_t* rccl_version) {
  if (!c) return dexr_set_error(DEXR_ERR_INVALID, "null argument");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) {
    int v = 0;
    if (g_rccl) (void)g_rccl->GetVersion(&v);
    *rccl_version = v;
  }
  return DEXR_OK;
}

int dexr_allgather(dexr_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!c || !send || !recv) return dexr_set_error(DEXR_ERR_INVALID, "null argument");
  if (bytes_per_rank == 0) return DEXR_OK;
  Rccl* r = g_rccl;
  // 4-byte elements when the shard allows it (every qpos shard does): RCCL's copy kernels vectorise on the element type
  if (bytes_per_rank % 4 == 0)
    RCCL_TRY(r, r->AllGather(send, recv, bytes_per_rank / 4, ncclFloat32, c->comm, static_cast<hipStream_t>(stream)));
  else
    RCCL_TRY(r, r->AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, static_cast<hipStream_t>(stream)));
  return DEXR_OK;
}

int dexr_comm_max_f64(dexr_comm* c, double* values_inout, int32_t n, void* stream) {
  if (!c || !values_inout) return dexr_set_error(DEXR_ERR_INVALID, "null argument");
  if (n < 1 || n > 8) return dexr_set_error(DEXR_ERR_INVALID, "n=%d outside 1..8", n);
  Rccl* r = g_rccl;
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::memcpy(c->h_scratch, values_inout, (size_t)n * sizeof(double));
  HIPC_TRY(hipMemcpyAsync(c->d_scratch, c->h_scratch, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  RCCL_TRY(r, r->AllReduce(c->d_scratch, c->d_scratch, (size_t)n, ncclFloat64, ncclMax, c->comm, st));
  HIPC_TRY(hipMemcpyAsync(c->h_scratch, c->d_scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPC_TRY(hipStreamSynchronize(st));
  std::memcpy(values_inout, c->h_scratch, (size_t)n * sizeof(double));
  return DEXR_OK;
}

int dexr_comm_barrier(dexr_comm* c, void* stream) {
  double v = 0.0;
  return dexr_comm_max_f64(c, &v, 1, stream);
}

}  // extern "C"
