// Elementwise kernels: ReLU/dropout backward, axpby, FFN output heads; plus the ABI's
// status/version helpers.  All HBM-bound streaming kernels with 16-byte lanes.
#include "rr_common.h"
#include <math.h>
#include <string.h>

namespace {

__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float scale, float* __restrict__ dz, float* __restrict__ acc,
                                                       int64_t n4, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n4; i += stride) {
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 r;
    r.x = v.x > 0.f ? g.x * scale : 0.f;
    r.y = v.y > 0.f ? g.y * scale : 0.f;
    r.z = v.z > 0.f ? g.z * scale : 0.f;
    r.w = v.w > 0.f ? g.w * scale : 0.f;
    if (dz) reinterpret_cast<f32x4*>(dz)[i] = r;
    if (acc) {
      f32x4 a = reinterpret_cast<f32x4*>(acc)[i];
      reinterpret_cast<f32x4*>(acc)[i] = a + r;
    }
  }
  for (int64_t i = n4 * 4 + t0; i < n; i += stride) {   // tail
    const float r = y[i] > 0.f ? dy[i] * scale : 0.f;
    if (dz) dz[i] = r;
    if (acc) acc[i] += r;
  }
}

// out = sum_k adds[k] + (y > 0 ? dy * scale : 0): the gradient of a residual that several layers read
// (d_inp = sum over the message-passing iterations of dZ_it, plus the ReLU-backward of the first layer) in
// ONE pass over memory instead of a read-modify-write per iteration.
constexpr int RR_MAX_ADDS = 15;   // message-passing depth of a step plan (MAXD = 16) minus one: every per-iteration dZ in ONE pass
struct AddPtrs {
  const float* p[RR_MAX_ADDS];
  int n;
};

__global__ void __launch_bounds__(256) relu_bwd_sum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float scale, const AddPtrs adds, float* __restrict__ out,
                                                           int64_t n4, int64_t n, int vec) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (vec) {
    for (int64_t i = t0; i < n4; i += stride) {
      const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
      const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
      f32x4 s = f32x4(0.f);
      for (int k = 0; k < adds.n; ++k) s = s + reinterpret_cast<const f32x4*>(adds.p[k])[i];   // fixed order
      f32x4 r;
      r.x = v.x > 0.f ? g.x * scale : 0.f;
      r.y = v.y > 0.f ? g.y * scale : 0.f;
      r.z = v.z > 0.f ? g.z * scale : 0.f;
      r.w = v.w > 0.f ? g.w * scale : 0.f;
      reinterpret_cast<f32x4*>(out)[i] = s + r;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) {
      float s = 0.f;
      for (int k = 0; k < adds.n; ++k) s += adds.p[k][i];
      out[i] = s + (y[i] > 0.f ? dy[i] * scale : 0.f);
    }
  } else {
    for (int64_t i = t0; i < n; i += stride) {
      float s = 0.f;
      for (int k = 0; k < adds.n; ++k) s += adds.p[k][i];
      out[i] = s + (y[i] > 0.f ? dy[i] * scale : 0.f);
    }
  }
}

__global__ void __launch_bounds__(256) relu_bwd_scalar_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float scale, float* __restrict__ dz,
                                                              float* __restrict__ acc, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float r = y[i] > 0.f ? dy[i] * scale : 0.f;
    if (dz) dz[i] = r;
    if (acc) acc[i] += r;
  }
}

__global__ void __launch_bounds__(256) axpby_kernel(float alpha, const float* __restrict__ a, float beta,
                                                    const float* __restrict__ b, float* __restrict__ out, int64_t n4,
                                                    int64_t n, int vec) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (vec) {
    for (int64_t i = t0; i < n4; i += stride) {
      f32x4 r = reinterpret_cast<const f32x4*>(a)[i] * alpha;
      if (b) r = r + reinterpret_cast<const f32x4*>(b)[i] * beta;
      reinterpret_cast<f32x4*>(out)[i] = r;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
  } else {
    for (int64_t i = t0; i < n; i += stride) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
  }
}

__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, int64_t n, uint32_t thr,
                                                      float keep_scale, uint64_t seed, float* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = rr_keep(seed, static_cast<uint64_t>(i), thr) ? x[i] * keep_scale : 0.f;
}

// Head layout (models/base_model.py:61-106).  raw is [M, N]; for the multi-parameter heads
// raw is split into G equal column chunks and the activated chunks are interleaved:
//   out[m, j*G + g] = act_g(raw[m, g*(N/G) + j])      (torch.stack(..., dim=2).view(size))
__device__ inline void head_map(int head, int N, int col_out, int* col_in, int* act, float* addc) {
  *col_in = col_out;
  *act = 0;
  *addc = 0.f;
  int G = 1;
  switch (head) {
    case RR_HEAD_SOFTPLUS: *act = 1; return;
    case RR_HEAD_SOFTPLUS_PLUS1: *act = 1; *addc = 1.0f; return;
    case RR_HEAD_EVIDENTIAL_RANKING: G = 2; break;
    case RR_HEAD_GAUSSIAN_SOFTPLUS: G = 2; break;
    case RR_HEAD_LOGNORM_SOFTPLUS: G = 2; break;
    case RR_HEAD_EVIDENTIAL4_SOFTPLUS: G = 4; break;
    default: return;
  }
  const int per = N / G;
  const int j = col_out / G, g = col_out % G;
  *col_in = g * per + j;
  const float mv = 1e-6f;
  if (head == RR_HEAD_EVIDENTIAL_RANKING) {
    if (g == 1) { *act = 1; *addc = mv; }
  } else if (head == RR_HEAD_GAUSSIAN_SOFTPLUS) {
    if (g == 1) { *act = 1; }
  } else if (head == RR_HEAD_LOGNORM_SOFTPLUS) {
    *act = 1; *addc = mv;
  } else {  // EVIDENTIAL4: mu, softplus+mv, softplus+mv+1, softplus+mv
    if (g >= 1) { *act = 1; *addc = (g == 2) ? (mv + 1.0f) : mv; }
  }
}

__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ raw, int64_t M, int N, int head,
                                                       float* __restrict__ out) {
  const int64_t total = M * N;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t m = e / N;
    const int c = static_cast<int>(e - m * N);
    int ci, act;
    float addc;
    head_map(head, N, c, &ci, &act, &addc);
    const float x = raw[m * N + ci];
    out[e] = act ? (rr_softplus(x) + addc) : x;
  }
}

__global__ void __launch_bounds__(256) head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ raw,
                                                       int64_t M, int N, int head, float* __restrict__ draw) {
  const int64_t total = M * N;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t m = e / N;
    const int c = static_cast<int>(e - m * N);
    int ci, act;
    float addc;
    head_map(head, N, c, &ci, &act, &addc);     // out column c reads raw column ci (a bijection)
    const float x = raw[m * N + ci];
    draw[m * N + ci] = act ? dout[e] * rr_softplus_grad(x) : dout[e];
  }
}

// ---------------------------------------------------------------- Adam over many tensors, one launch
struct AdamMany {
  rr_adam_tensor t[RR_MAX_ADAM];
  int32_t first_block[RR_MAX_ADAM + 1];     // prefix sum of the tensors' 1024-element blocks
  int n;
  double step_size, bc2_sqrt, beta2, one_m_b1, one_m_b2, eps, wd;
};

// torch's fused Adam keeps its scalars in double, so each element's update is double arithmetic rounded to f32 once
// (fused_adam_utils.cuh); the same here - 0.8 M elements, the kernel stays launch-latency-sized.
__global__ void __launch_bounds__(256) adam_many_kernel(const AdamMany A) {
  const int b = blockIdx.x;
  int ti = 0;
  while (ti + 1 < A.n && b >= A.first_block[ti + 1]) ++ti;          // (block-uniform: a scalar loop over <= 64 entries)
  const rr_adam_tensor T = A.t[ti];
  const int64_t base = static_cast<int64_t>(b - A.first_block[ti]) * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    if (i < T.n) {
      const float pf = T.p[i];
      double g = static_cast<double>(T.g[i]);
      if (A.wd != 0.0) g = g + A.wd * static_cast<double>(pf);
      const float gf = static_cast<float>(g);                          // (torch: grad += weight_decay * param, an f32 value)
      const float m = static_cast<float>(static_cast<double>(T.m[i]) + A.one_m_b1 * (static_cast<double>(gf) - static_cast<double>(T.m[i])));
      const float v = static_cast<float>(A.beta2 * static_cast<double>(T.v[i]) + A.one_m_b2 * static_cast<double>(gf) * static_cast<double>(gf));
      const double denom = sqrt(static_cast<double>(v)) / A.bc2_sqrt + A.eps;
      T.m[i] = m;
      T.v[i] = v;
      T.p[i] = static_cast<float>(static_cast<double>(pf) - A.step_size * static_cast<double>(m) / denom);
    }
  }
}

}  // namespace

extern "C" {

int rr_adam_step_f32(const rr_adam_tensor* descs, int n_tensors, int64_t step, double lr, double beta1, double beta2, double eps,
                     double weight_decay, rr_stream_t stream) {
  RR_CHECK_ARG(descs && n_tensors >= 0 && n_tensors <= RR_MAX_ADAM && step >= 1);
  RR_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0);
  AdamMany A;
  memset(&A, 0, sizeof(A));
  int64_t blocks = 0;
  for (int i = 0; i < n_tensors; ++i) {
    const rr_adam_tensor& t = descs[i];
    if (t.g == nullptr || t.n == 0) continue;
    RR_CHECK_ARG(t.p && t.m && t.v && t.n > 0);
    A.t[A.n] = t;
    A.first_block[A.n] = static_cast<int32_t>(blocks);
    blocks += (t.n + 1023) / 1024;
    if (blocks > (int64_t(1) << 30)) return RR_ERR_UNSUPPORTED;
    ++A.n;
  }
  if (A.n == 0) return RR_OK;
  A.first_block[A.n] = static_cast<int32_t>(blocks);
  const double bc1 = 1.0 - pow(beta1, static_cast<double>(step));
  const double bc2 = 1.0 - pow(beta2, static_cast<double>(step));
  A.step_size = lr / bc1;
  A.bc2_sqrt = sqrt(bc2);
  A.beta2 = beta2; A.eps = eps; A.wd = weight_decay;
  A.one_m_b1 = 1.0 - beta1;
  A.one_m_b2 = 1.0 - beta2;
  adam_many_kernel<<<static_cast<unsigned>(blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(A);
  return rr_launch_status();
}

const char* rr_strerror(int status) {
  switch (status) {
    case RR_OK: return "ok";
    case RR_ERR_ARG: return "invalid argument (null pointer, negative size or inconsistent shapes)";
    case RR_ERR_ALIGN: return "pointer or leading dimension not aligned as required";
    case RR_ERR_LAUNCH: return "HIP kernel launch failed";
    case RR_ERR_UNSUPPORTED: return "size outside the supported range";
    case RR_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

int rr_version(void) { return RR_ABI_VERSION; }

size_t rr_abi_gather_epi_size(void) { return sizeof(rr_gather_epi); }

void rr_abi_struct_sizes(size_t* linear_args, size_t* wgrad_args) {
  if (linear_args) *linear_args = sizeof(rr_linear_args);
  if (wgrad_args) *wgrad_args = sizeof(rr_wgrad_args);
}

int rr_dropout_keep_host(uint64_t seed, uint64_t index, float p) {
  return rr_keep(seed, index, rr_drop_threshold(p)) ? 1 : 0;
}

int rr_dropout_f32(const float* x, int64_t n, float p, uint64_t seed, float* out, rr_stream_t stream) {
  RR_CHECK_ARG(x && out && n >= 0 && p >= 0.f && p < 1.f);
  if (n == 0) return RR_OK;
  dropout_kernel<<<rr_grid_for(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(x, n, rr_drop_threshold(p),
                                                                                    1.0f / (1.0f - p), seed, out);
  return rr_launch_status();
}

int rr_relu_bwd_f32(const float* dy, const float* y, float scale, float* dz, float* acc, int64_t n,
                    rr_stream_t stream) {
  RR_CHECK_ARG(dy && y && (dz || acc) && n >= 0);
  if (n == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = rr_aligned16(dy) && rr_aligned16(y) && (!dz || rr_aligned16(dz)) && (!acc || rr_aligned16(acc));
  if (vec) {
    relu_bwd_kernel<<<rr_grid_for((n + 3) / 4, 256), 256, 0, s>>>(dy, y, scale, dz, acc, n / 4, n);
  } else {
    relu_bwd_scalar_kernel<<<rr_grid_for(n, 256), 256, 0, s>>>(dy, y, scale, dz, acc, n);
  }
  return rr_launch_status();
}

int rr_relu_bwd_sum_f32(const float* dy, const float* y, float scale, const float* const* adds, int n_adds,
                        float* out, int64_t n, rr_stream_t stream) {
  RR_CHECK_ARG(dy && y && out && n >= 0 && n_adds >= 0 && n_adds <= RR_MAX_ADDS && (n_adds == 0 || adds));
  if (n == 0) return RR_OK;
  AddPtrs A;
  A.n = n_adds;
  bool vec = rr_aligned16(dy) && rr_aligned16(y) && rr_aligned16(out);
  for (int k = 0; k < RR_MAX_ADDS; ++k) {
    A.p[k] = k < n_adds ? adds[k] : nullptr;
    if (k < n_adds) {
      RR_CHECK_ARG(adds[k]);
      vec = vec && rr_aligned16(adds[k]);
    }
  }
  relu_bwd_sum_kernel<<<rr_grid_for(vec ? (n + 3) / 4 : n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      dy, y, scale, A, out, n / 4, n, vec ? 1 : 0);
  return rr_launch_status();
}

int rr_axpby_f32(float alpha, const float* a, float beta, const float* b, float* out, int64_t n,
                 rr_stream_t stream) {
  RR_CHECK_ARG(a && out && n >= 0);
  if (n == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int vec = rr_aligned16(a) && rr_aligned16(out) && (!b || rr_aligned16(b));
  axpby_kernel<<<rr_grid_for((n + 3) / 4, 256), 256, 0, s>>>(alpha, a, beta, b, out, n / 4, n, vec);
  return rr_launch_status();
}

static bool head_ok(int head, int N) {
  switch (head) {
    case RR_HEAD_IDENTITY:
    case RR_HEAD_SOFTPLUS:
    case RR_HEAD_SOFTPLUS_PLUS1: return true;
    case RR_HEAD_EVIDENTIAL_RANKING:
    case RR_HEAD_GAUSSIAN_SOFTPLUS:
    case RR_HEAD_LOGNORM_SOFTPLUS: return N % 2 == 0;
    case RR_HEAD_EVIDENTIAL4_SOFTPLUS: return N % 4 == 0;
    default: return false;
  }
}

int rr_head_fwd_f32(const float* raw, int64_t M, int N, int head, float* out, rr_stream_t stream) {
  RR_CHECK_ARG(raw && out && M >= 0 && N >= 1 && head_ok(head, N));
  if (M == 0) return RR_OK;
  head_fwd_kernel<<<rr_grid_for(M * N, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(raw, M, N, head, out);
  return rr_launch_status();
}

int rr_head_bwd_f32(const float* dout, const float* raw, int64_t M, int N, int head, float* draw,
                    rr_stream_t stream) {
  RR_CHECK_ARG(dout && raw && draw && M >= 0 && N >= 1 && head_ok(head, N));
  if (M == 0) return RR_OK;
  head_bwd_kernel<<<rr_grid_for(M * N, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(dout, raw, M, N, head, draw);
  return rr_launch_status();
}

}  // extern "C"
