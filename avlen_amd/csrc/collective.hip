// grad_allreduce: the one collective of the path behind the C ABI.
//
// The reference averages pi_q's gradients over the ranks once per optimiser step through DistributedDataParallel's reducer
// (ss_baselines/savi/ddppo/algo/ddppo.py:75-96; GLOO / TCP in the yaml).  Here the trained parameters' gradient is ONE contiguous
// fp32 range (engine.FlatParams), so the exchange is a single in-place ncclAllReduce(avg) on the stream the backward ran on --
// RCCL, i.e. xGMI inside a node.  RCCL is bound at RUN time (dlopen of the librccl the process already holds -- PyTorch ships one
// -- else the ROCm one), so libavlen_hip.so keeps linking against nothing but the HIP runtime, and a single-GPU user never loads it.
#include <dlfcn.h>
#include <string.h>
#include "common.h"
#include "../../include/avlen_hip.h"

namespace {
struct UniqueId { char internal[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
struct Rccl { void* h; GetUniqueIdFn get_id; CommInitRankFn init; AllReduceFn allreduce; CommDestroyFn destroy; };
Rccl g_rccl = {};

const Rccl* rccl() {
  if (g_rccl.h) return g_rccl.allreduce ? &g_rccl : nullptr;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);      // the instance the process already loaded (torch's)
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return nullptr;
  g_rccl.h = h;
  g_rccl.get_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  g_rccl.init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  if (!g_rccl.get_id || !g_rccl.init || !g_rccl.destroy) g_rccl.allreduce = nullptr;
  return g_rccl.allreduce ? &g_rccl : nullptr;
}
}  // namespace

extern "C" int avlen_comm_unique_id(void* out, size_t bytes) {
  const Rccl* r = rccl();
  if (!r || !out || bytes < sizeof(UniqueId)) return AVLEN_ERR_ARG;
  UniqueId id;
  if (r->get_id(&id) != 0) return AVLEN_ERR_LAUNCH;
  memcpy(out, &id, sizeof(id));
  return AVLEN_OK;
}

extern "C" int avlen_comm_init_rank(void** comm, int nranks, const void* unique_id, int rank) {
  const Rccl* r = rccl();
  if (!r || !comm || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return AVLEN_ERR_ARG;
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  return r->init(comm, nranks, id, rank) == 0 ? AVLEN_OK : AVLEN_ERR_LAUNCH;
}

extern "C" int avlen_comm_destroy(void* comm) {
  const Rccl* r = rccl();
  if (!r || !comm) return AVLEN_ERR_ARG;
  return r->destroy(comm) == 0 ? AVLEN_OK : AVLEN_ERR_LAUNCH;
}

// bucket[0 .. count) <- mean over the ranks of `comm`, in place, on `stream`.  dtype: AVLEN_PREC_FP32 (fp32) | AVLEN_PREC_BF16 (bf16).
extern "C" int avlen_grad_allreduce(void* bucket, size_t count, int dtype, void* comm, hipStream_t stream) {
  const Rccl* r = rccl();
  if (!r || !comm || (count && !bucket) || (dtype != AVLEN_PREC_FP32 && dtype != AVLEN_PREC_BF16)) return AVLEN_ERR_ARG;
  if (count == 0) return AVLEN_OK;
  const int nccl_dtype = dtype == AVLEN_PREC_FP32 ? 7 : 9;     // ncclFloat32 | ncclBfloat16
  return r->allreduce(bucket, bucket, count, nccl_dtype, 4 /* ncclAvg */, comm, stream) == 0 ? AVLEN_OK : AVLEN_ERR_LAUNCH;
}
