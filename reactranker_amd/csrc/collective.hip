// Gradient all-reduce of a data-parallel job behind the C-ABI (rr_allreduce_f32).
//
// The reference has no distributed code; the north star shards whole queries over the GPUs of a node and
// sums gradients with ONE RCCL all-reduce of a flat fp32 bucket per step.  The Python mirror does that through
// torch.distributed (reactranker_amd/dp.py: the communicator belongs to the launcher).  These entry points give a
// non-Python host the same step: forward, loss kernel, backward, rr_allreduce_f32 on the gradient buffers it handed to
// rr_reaction_backward.  RCCL is resolved at run time - first among the symbols the process already has (a host that
// links RCCL, or torch's bundled copy), then by dlopen("librccl.so.1") - so the library carries no link-time dependency
// and never brings a second RCCL into a process that has one.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "rr_common.h"

namespace {

struct UniqueId { char internal[RR_COMM_ID_BYTES]; };     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_id_fn)(UniqueId*);
typedef int (*init_rank_fn)(void**, int, UniqueId, int);
typedef int (*destroy_fn)(void*);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*reduce_scatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*comm_int_fn)(const void*, int*);

struct Rccl {
  bool tried = false, ok = false;
  int how = 0;                                            // 1: symbols already in the process, 2: dlopen'ed here
  get_id_fn get_id = nullptr;
  init_rank_fn init_rank = nullptr;
  destroy_fn destroy = nullptr;
  allreduce_fn allreduce = nullptr;
  reduce_scatter_fn reduce_scatter = nullptr;             // optional: the two-phase form of rr_allreduce_rsag_f32
  all_gather_fn all_gather = nullptr;
  comm_int_fn comm_count = nullptr, comm_rank = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

const Rccl& rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (!g_rccl.tried) {
    g_rccl.tried = true;
    // RTLD_DEFAULT is a null handle on glibc: "found in the process" must be tracked apart from the handle's value
    void* h = RTLD_DEFAULT;
    bool have = dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
    if (!have) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      have = h != nullptr;
      g_rccl.how = 2;
    } else {
      g_rccl.how = 1;
    }
    if (have) {
      g_rccl.get_id = reinterpret_cast<get_id_fn>(dlsym(h, "ncclGetUniqueId"));
      g_rccl.init_rank = reinterpret_cast<init_rank_fn>(dlsym(h, "ncclCommInitRank"));
      g_rccl.destroy = reinterpret_cast<destroy_fn>(dlsym(h, "ncclCommDestroy"));
      g_rccl.allreduce = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
      g_rccl.reduce_scatter = reinterpret_cast<reduce_scatter_fn>(dlsym(h, "ncclReduceScatter"));
      g_rccl.all_gather = reinterpret_cast<all_gather_fn>(dlsym(h, "ncclAllGather"));
      g_rccl.comm_count = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommCount"));
      g_rccl.comm_rank = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommUserRank"));
      g_rccl.ok = g_rccl.get_id && g_rccl.init_rank && g_rccl.destroy && g_rccl.allreduce;
    }
    if (!g_rccl.ok) g_rccl.how = 0;
  }
  return g_rccl;
}

constexpr int kNcclFloat32 = 7, kNcclSum = 0;             // rccl.h: ncclFloat32, ncclSum

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ x, int64_t n4, int64_t n, float s) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<f32x4*>(x)[i];
    reinterpret_cast<f32x4*>(x)[i] = v * s;
  }
  for (int64_t i = n4 * 4 + t0; i < n; i += stride) x[i] *= s;
}

}  // namespace

extern "C" {

int rr_comm_backend(void) { return rccl().how; }

int rr_comm_unique_id(void* id) {
  RR_CHECK_ARG(id);
  const Rccl& r = rccl();
  if (!r.ok) return RR_ERR_UNSUPPORTED;
  return r.get_id(static_cast<UniqueId*>(id)) == 0 ? RR_OK : RR_ERR_LAUNCH;
}

int rr_comm_init_rank(rr_comm_t* comm, int n_ranks, const void* id, int rank) {
  RR_CHECK_ARG(comm && id && n_ranks >= 1 && rank >= 0 && rank < n_ranks);
  const Rccl& r = rccl();
  if (!r.ok) return RR_ERR_UNSUPPORTED;
  UniqueId u;
  memcpy(&u, id, sizeof(u));
  void* c = nullptr;
  if (r.init_rank(&c, n_ranks, u, rank) != 0 || c == nullptr) return RR_ERR_LAUNCH;
  *comm = c;
  return RR_OK;
}

int rr_comm_destroy(rr_comm_t comm) {
  RR_CHECK_ARG(comm);
  const Rccl& r = rccl();
  if (!r.ok) return RR_ERR_UNSUPPORTED;
  return r.destroy(comm) == 0 ? RR_OK : RR_ERR_LAUNCH;
}

int rr_allreduce_f32(float* buf, int64_t n, float scale, rr_comm_t comm, rr_stream_t stream) {
  RR_CHECK_ARG(buf && comm && n >= 0);
  if (n == 0) return RR_OK;
  const Rccl& r = rccl();
  if (!r.ok) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (r.allreduce(buf, buf, static_cast<size_t>(n), kNcclFloat32, kNcclSum, comm, s) != 0) return RR_ERR_LAUNCH;
  if (scale != 1.0f) {
    const bool vec = rr_aligned16(buf);
    scale_kernel<<<rr_grid_for(vec ? (n + 3) / 4 : n, 256), 256, 0, s>>>(buf, vec ? n / 4 : 0, n, scale);
    return rr_launch_status();
  }
  return RR_OK;
}

// The same sum as TWO collectives - reduce-scatter, then all-gather, in place - over the largest prefix of the bucket that
// divides by the rank count; the < n_ranks elements behind it go through a (tiny) all-reduce.  SURVEY.md section 5: xGMI is
// point-to-point (7 links per GPU), so for the 3-12 MB gradient bucket a direct reduce-scatter + all-gather lets every rank
// own 1/R of the reduction and use all its links at once, where a ring all-reduce is bound by one link.  Which of the two
// RCCL runs faster at these sizes is a measurement for an 8-GPU node (none was available to any round so far): this entry
// point exists so that the measurement is one flag away, rr_allreduce_f32 stays the default.  Summation order: element i of
// chunk c is reduced by RCCL's reduce-scatter for the rank that owns c - with 2 ranks the same bits as the all-reduce (a + b),
// with more ranks the order inside RCCL may differ from its all-reduce algorithm's (both are fixed per communicator).
int rr_allreduce_rsag_f32(float* buf, int64_t n, float scale, rr_comm_t comm, rr_stream_t stream) {
  RR_CHECK_ARG(buf && comm && n >= 0);
  if (n == 0) return RR_OK;
  const Rccl& r = rccl();
  if (!r.ok) return RR_ERR_UNSUPPORTED;
  if (!r.reduce_scatter || !r.all_gather || !r.comm_count || !r.comm_rank) return RR_ERR_UNSUPPORTED;
  int R = 0, me = -1;
  if (r.comm_count(comm, &R) != 0 || r.comm_rank(comm, &me) != 0 || R < 1 || me < 0 || me >= R) return RR_ERR_LAUNCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t chunk = n / R, body = chunk * R;
  if (chunk > 0) {
    float* mine = buf + static_cast<int64_t>(me) * chunk;              // in-place forms: recv = send + rank * count
    if (r.reduce_scatter(buf, mine, static_cast<size_t>(chunk), kNcclFloat32, kNcclSum, comm, s) != 0) return RR_ERR_LAUNCH;
    if (r.all_gather(mine, buf, static_cast<size_t>(chunk), kNcclFloat32, comm, s) != 0) return RR_ERR_LAUNCH;
  }
  if (body < n && r.allreduce(buf + body, buf + body, static_cast<size_t>(n - body), kNcclFloat32, kNcclSum, comm, s) != 0)
    return RR_ERR_LAUNCH;
  if (scale != 1.0f) {
    const bool vec = rr_aligned16(buf);
    scale_kernel<<<rr_grid_for(vec ? (n + 3) / 4 : n, 256), 256, 0, s>>>(buf, vec ? n / 4 : 0, n, scale);
    return rr_launch_status();
  }
  return RR_OK;
}

}  // extern "C"
