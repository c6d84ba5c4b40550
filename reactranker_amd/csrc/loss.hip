// Per-query ranking losses: ListMLE, ListNet, evidential UC-Listwise, RankNet, plus the
// pointwise MSE / Gaussian NLL and the standalone LogCumsumExp op.
//
// One 64-lane wavefront owns one list (a query's candidates).  The list is staged in LDS
// (5 * max_len floats), ranked by target with an O(C^2/64) counting pass, and reduced /
// scanned with wave shuffles; lists longer than 64 give each lane a contiguous chunk.
// Only wave-level synchronisation is used (no workgroup barrier), so waves of different
// list lengths never wait on each other.  Per-query partials are finished by a fixed-order
// second kernel: no float atomics, results are run-to-run identical.
//
// The reference evaluates these losses as a Python loop of ~10 tiny ATen ops per query
// (train/loss.py:86-97, 338-347, 504-554; train/train_pairwise.py:99-137).
#include "rr_common.h"

namespace {

constexpr int kMaxLen = 8192;
static_assert(kMaxLen <= 65536, "ranking_metrics_kernel keeps list positions in 16 bits");

__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct ListView {
  float* s;       // scores (as given)
  float* t;       // targets
  float* ss;      // scores sorted by target, descending
  int32_t* perm;  // perm[j] = original position of sorted element j
  float* aux;     // scratch (fd values)
};

__device__ inline ListView carve(float* sm, int L) {
  ListView v;
  v.s = sm;
  v.t = sm + L;
  v.ss = sm + 2 * L;
  v.perm = reinterpret_cast<int32_t*>(sm + 3 * L);
  v.aux = sm + 4 * L;
  return v;
}

// rank by target, descending, ties by original index (stable); scatters s into ss / perm.
__device__ inline void rank_sort(const ListView& v, int C, int lane) {
  for (int i = lane; i < C; i += RR_WAVE) {
    const float ti = v.t[i];
    int r = 0;
    for (int j = 0; j < C; ++j) {
      const float tj = v.t[j];
      r += (tj > ti || (tj == ti && j < i)) ? 1 : 0;
    }
    v.ss[r] = v.s[i];
    v.perm[r] = i;
  }
  wave_sync();
}

__device__ inline float list_max(const float* a, int C, int lane) {
  float m = -INFINITY;
  for (int i = lane; i < C; i += RR_WAVE) m = fmaxf(m, a[i]);
  return rr_wave_max(m);
}

// fd[j] = log(sum_{i>=j} exp(x[i] - m)) + m for the C values in x (LogCumsumExp.forward,
// train/loss.py:28-34).  Each lane owns the contiguous chunk [lo, hi).
__device__ inline void logcumsumexp_rev(const float* x, float* fd, int C, int lane, float m) {
  const int E = (C + RR_WAVE - 1) / RR_WAVE;
  const int lo = min(lane * E, C), hi = min(lo + E, C);
  float local = 0.f;
  for (int j = hi - 1; j >= lo; --j) local += expf(x[j] - m);
  // suffix sums over lanes without subtraction: scan the lane-reversed values
  const float rev = __shfl(local, RR_WAVE - 1 - lane, RR_WAVE);
  const float incl_rev = rr_wave_incl_scan(rev, lane);
  const float suffix_incl = __shfl(incl_rev, RR_WAVE - 1 - lane, RR_WAVE);   // sum over lanes >= lane
  const float nxt = __shfl_down(suffix_incl, 1, RR_WAVE);
  float run = (lane == RR_WAVE - 1) ? 0.f : nxt;                             // sum over lanes > lane, no subtraction
  for (int j = hi - 1; j >= lo; --j) {
    run += expf(x[j] - m);
    fd[j] = logf(run) + m;
  }
  wave_sync();
}

// cs[j] = sum_{i<=j} v(i), v(i) = exp(-fd[i]); returned through out[] (may alias nothing else).
__device__ inline void cumsum_exp_neg(const float* fd, float* out, int C, int lane) {
  const int E = (C + RR_WAVE - 1) / RR_WAVE;
  const int lo = min(lane * E, C), hi = min(lo + E, C);
  float local = 0.f;
  for (int j = lo; j < hi; ++j) local += expf(-fd[j]);
  const float incl = rr_wave_incl_scan(local, lane);
  const float prev = __shfl_up(incl, 1, RR_WAVE);
  float run = lane == 0 ? 0.f : prev;
  for (int j = lo; j < hi; ++j) {
    run += expf(-fd[j]);
    out[j] = run;
  }
  wave_sync();
}

// ---------------------------------------------------------------- fused loss + gradient launches ("step" entry points)
// One launch writes the loss AND d loss / d score for an upstream gradient of one (what `loss.backward()` feeds a loss that
// is the root of the graph): the reference's trainer step (train_listwise.py:287-288) then needs no second loss kernel, no
// separate reduction launch and no host-created gradient.  After its partial is written every workgroup draws a ticket; the
// one that draws the last sums all partials in reduce_scale_kernel's order - 256 strided accumulators, then its halving
// tree; here on one wave, lane l playing threads l, l + 64, l + 128, l + 192 - so the loss has the bits of the two-kernel
// path, and re-arms the counter (one zero-initialised device word the caller keeps) for the next launch.
__device__ inline void finish_last(const float* partial, int n, float scale, float* out, unsigned int* counter, int lane) {
  __threadfence();                                                   // release: this workgroup's partial
  unsigned int ticket = 0u;
  if (lane == 0) ticket = atomicAdd(counter, 1u);
  ticket = __shfl(ticket, 0, RR_WAVE);
  if (ticket != static_cast<unsigned int>(n) - 1u) return;
  __threadfence();                                                   // acquire: every other workgroup's partial
  float a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float acc = 0.f;
    for (int i = lane + 64 * u; i < n; i += 256) acc += partial[i];
    a[u] = acc;
  }
  float r = (a[0] + a[2]) + (a[1] + a[3]);                           // tree steps o = 128 and o = 64
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, RR_WAVE);  // red[t] += red[t + o] for t < o
  if (lane == 0) {
    out[0] = r * scale;
    *counter = 0u;
  }
}

// ---------------------------------------------------------------- ListMLE
__global__ void __launch_bounds__(RR_WAVE) listmle_fwd_kernel(const float* __restrict__ score, int64_t sstride,
                                                              const float* __restrict__ targets,
                                                              const int32_t* __restrict__ seg_off, int L,
                                                              float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) {
    if (lane == 0) partial[q] = 0.f;
    return;
  }
  ListView v = carve(sm, L);
  for (int i = lane; i < C; i += RR_WAVE) {
    v.s[i] = score[static_cast<int64_t>(off + i) * sstride];
    v.t[i] = targets[off + i];
  }
  wave_sync();
  rank_sort(v, C, lane);
  const float m = list_max(v.ss, C, lane);
  logcumsumexp_rev(v.ss, v.aux, C, lane, m);
  float acc = 0.f;
  for (int i = lane; i < C; i += RR_WAVE) acc += v.aux[i] - v.ss[i];
  acc = rr_wave_sum(acc);
  if (lane == 0) partial[q] = acc / static_cast<float>(C);          // torch.mean, loss.py:94
}

__global__ void __launch_bounds__(RR_WAVE) listmle_bwd_kernel(const float* __restrict__ score, int64_t sstride,
                                                              const float* __restrict__ targets,
                                                              const int32_t* __restrict__ seg_off, int L, int Q,
                                                              const float* __restrict__ gloss,
                                                              float* __restrict__ dscore, int64_t dstride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) return;
  ListView v = carve(sm, L);
  for (int i = lane; i < C; i += RR_WAVE) {
    v.s[i] = score[static_cast<int64_t>(off + i) * sstride];
    v.t[i] = targets[off + i];
  }
  wave_sync();
  rank_sort(v, C, lane);
  const float m = list_max(v.ss, C, lane);
  logcumsumexp_rev(v.ss, v.aux, C, lane, m);
  cumsum_exp_neg(v.aux, v.t, C, lane);                              // v.t now holds cumsum(exp(-fd))
  const float g = gloss[0] / (static_cast<float>(C) * static_cast<float>(Q));
  for (int j = lane; j < C; j += RR_WAVE) {
    // LogCumsumExp.backward keeps the un-shifted exp(x) (loss.py:59); "- 1" is d(-sorted_item)
    const float d = g * (expf(v.ss[j]) * v.t[j]) - g;
    dscore[static_cast<int64_t>(off + v.perm[j]) * dstride] = d;
  }
}

// forward + backward (upstream gradient one) of a list in one pass over its staged copy; same operations in the same order
// as the two kernels above, so partial[q] and dscore have their bits
__global__ void __launch_bounds__(RR_WAVE) listmle_step_kernel(const float* __restrict__ score, int64_t sstride,
                                                               const float* __restrict__ targets,
                                                               const int32_t* __restrict__ seg_off, int L, int Q,
                                                               float* __restrict__ partial, float* __restrict__ dscore,
                                                               int64_t dstride, float scale, float* __restrict__ loss,
                                                               unsigned int* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) {
    if (lane == 0) partial[q] = 0.f;
  } else {
    ListView v = carve(sm, L);
    for (int i = lane; i < C; i += RR_WAVE) {
      v.s[i] = score[static_cast<int64_t>(off + i) * sstride];
      v.t[i] = targets[off + i];
    }
    wave_sync();
    rank_sort(v, C, lane);
    const float m = list_max(v.ss, C, lane);
    logcumsumexp_rev(v.ss, v.aux, C, lane, m);
    float acc = 0.f;
    for (int i = lane; i < C; i += RR_WAVE) acc += v.aux[i] - v.ss[i];
    acc = rr_wave_sum(acc);
    if (lane == 0) partial[q] = acc / static_cast<float>(C);
    cumsum_exp_neg(v.aux, v.t, C, lane);
    const float g = 1.0f / (static_cast<float>(C) * static_cast<float>(Q));
    for (int j = lane; j < C; j += RR_WAVE) {
      const float d = g * (expf(v.ss[j]) * v.t[j]) - g;
      dscore[static_cast<int64_t>(off + v.perm[j]) * dstride] = d;
    }
  }
  finish_last(partial, Q, scale, loss, counter, lane);
}

// ---------------------------------------------------------------- softmax helpers
__device__ inline void softmax_stats(const float* a, int C, int lane, float* mx, float* sum) {
  const float m = list_max(a, C, lane);
  float s = 0.f;
  for (int i = lane; i < C; i += RR_WAVE) s += expf(a[i] - m);
  *mx = m;
  *sum = rr_wave_sum(s);
}

// ---------------------------------------------------------------- ListNet
__global__ void __launch_bounds__(RR_WAVE) listnet_kernel(const float* __restrict__ score, int64_t sstride,
                                                          const float* __restrict__ targets,
                                                          const int32_t* __restrict__ seg_off, int L, int bwd,
                                                          float* __restrict__ partial, const float* __restrict__ gloss,
                                                          float inv_total, float* __restrict__ dscore,
                                                          int64_t dstride, float* __restrict__ loss = nullptr,
                                                          unsigned int* __restrict__ counter = nullptr) {
  // bwd: 0 = forward (partial), 1 = backward (dscore), 2 = both in one pass + the last-arriver reduction (step entry point:
  // gloss == nullptr means an upstream gradient of one)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) {
    if (bwd != 1 && lane == 0) partial[q] = 0.f;
    if (bwd == 2) finish_last(partial, gridDim.x, inv_total, loss, counter, lane);
    return;
  }
  float* s = sm;
  float* t = sm + L;
  for (int i = lane; i < C; i += RR_WAVE) {
    s[i] = score[static_cast<int64_t>(off + i) * sstride];
    t[i] = targets[off + i];
  }
  wave_sync();
  float ms, zs, mt, zt;
  softmax_stats(s, C, lane, &ms, &zs);
  softmax_stats(t, C, lane, &mt, &zt);
  if (bwd != 1) {
    float acc = 0.f;
    for (int i = lane; i < C; i += RR_WAVE) {
      const float pred = logf(expf(s[i] - ms) / zs);                // torch.log(F.softmax(item)), loss.py:339
      const float targ = expf(t[i] - mt) / zt;                      // loss.py:341
      acc += -targ * pred;                                          // loss.py:343
    }
    acc = rr_wave_sum(acc);
    if (lane == 0) partial[q] = acc;
  }
  if (bwd != 0) {
    float tsum = 0.f;
    for (int i = lane; i < C; i += RR_WAVE) tsum += expf(t[i] - mt) / zt;
    tsum = rr_wave_sum(tsum);
    const float g = (gloss ? gloss[0] : 1.0f) * inv_total;
    for (int i = lane; i < C; i += RR_WAVE) {
      const float p = expf(s[i] - ms) / zs;
      const float targ = expf(t[i] - mt) / zt;
      dscore[static_cast<int64_t>(off + i) * dstride] = g * (p * tsum - targ);
    }
  }
  if (bwd == 2) finish_last(partial, gridDim.x, inv_total, loss, counter, lane);
}

// ---------------------------------------------------------------- evidential UC-Listwise
__global__ void __launch_bounds__(RR_WAVE) evidential_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                             int64_t stride, const float* __restrict__ targets,
                                                             const int32_t* __restrict__ seg_off, int L, int Q, int bwd,
                                                             float* __restrict__ partial,
                                                             const float* __restrict__ gloss, float* __restrict__ dmu,
                                                             float* __restrict__ dvar, int64_t dstride,
                                                             float* __restrict__ loss = nullptr,
                                                             unsigned int* __restrict__ counter = nullptr) {
  // bwd: 0 = forward, 1 = backward, 2 = both + the last-arriver reduction (see listnet_kernel)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) {
    if (bwd != 1 && lane == 0) partial[q] = 0.f;
    if (bwd == 2) finish_last(partial, gridDim.x, 1.0f / static_cast<float>(Q), loss, counter, lane);
    return;
  }
  float* s = sm;
  float* t = sm + L;
  float* vv = sm + 2 * L;
  for (int i = lane; i < C; i += RR_WAVE) {
    s[i] = mu[static_cast<int64_t>(off + i) * stride];
    vv[i] = var[static_cast<int64_t>(off + i) * stride];
    t[i] = targets[off + i];
  }
  wave_sync();
  float ms, zs, mt, zt;
  softmax_stats(s, C, lane, &ms, &zs);
  softmax_stats(t, C, lane, &mt, &zt);
  const float two_pi = 2.0f * 3.141592653f;                         // loss.py:543 (truncated pi)
  if (bwd != 1) {
    float acc = 0.f;
    for (int i = lane; i < C; i += RR_WAVE) {
      const float lp = logf(expf(s[i] - ms) / zs);
      const float lt = logf(expf(t[i] - mt) / zt);
      const float d = lt - lp;
      const float unc = 0.5f * (d * d) / vv[i] + 0.5f * logf(two_pi * vv[i]);   // loss.py:541-543
      const float pen = fabsf(s[i] - t[i]);                                       // loss.py:545
      acc += -lt + unc + pen;                                                      // loss.py:549
    }
    acc = rr_wave_sum(acc);
    if (lane == 0) partial[q] = acc / static_cast<float>(C);
  }
  if (bwd != 0) {
    float csum = 0.f;
    for (int i = lane; i < C; i += RR_WAVE) {
      const float lp = logf(expf(s[i] - ms) / zs);
      const float lt = logf(expf(t[i] - mt) / zt);
      csum += -(lt - lp) / vv[i];
    }
    csum = rr_wave_sum(csum);
    const float g = (gloss ? gloss[0] : 1.0f) / (static_cast<float>(C) * static_cast<float>(Q));
    for (int i = lane; i < C; i += RR_WAVE) {
      const float p = expf(s[i] - ms) / zs;
      const float lp = logf(p);
      const float lt = logf(expf(t[i] - mt) / zt);
      const float d = lt - lp;
      const float c = -d / vv[i];
      const float diff = s[i] - t[i];
      const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      dmu[static_cast<int64_t>(off + i) * dstride] = g * (c - p * csum + sgn);
      dvar[static_cast<int64_t>(off + i) * dstride] = g * (-0.5f * d * d / (vv[i] * vv[i]) + 0.5f / vv[i]);
    }
  }
  if (bwd == 2) finish_last(partial, gridDim.x, 1.0f / static_cast<float>(Q), loss, counter, lane);
}

// ---------------------------------------------------------------- RankNet
__global__ void __launch_bounds__(RR_WAVE) ranknet_fwd_kernel(const float* __restrict__ score, int64_t sstride,
                                                              const float* __restrict__ targets,
                                                              const int32_t* __restrict__ seg_off, int L, float sigma,
                                                              float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) {
    if (lane == 0) {
      partial[2 * q] = 0.f;
      reinterpret_cast<int32_t*>(partial)[2 * q + 1] = 0;
    }
    return;
  }
  float* s = sm;
  float* t = sm + L;
  for (int i = lane; i < C; i += RR_WAVE) {
    s[i] = score[static_cast<int64_t>(off + i) * sstride];
    t[i] = targets[off + i];
  }
  wave_sync();
  float acc = 0.f;
  int npos = 0;
  for (int i = lane; i < C; i += RR_WAVE) {
    const float ti = t[i], si = s[i];
    for (int j = 0; j < C; ++j) {
      const float rel = ti - t[j];
      const float x = sigma * (si - s[j]);
      if (rel > 0.f) {
        ++npos;
        acc += logf(1.0f + expf(-x));                              // C_pos, train_pairwise.py:119 (naive, may be inf)
      } else if (rel < 0.f) {
        acc += logf(1.0f + expf(x));                               // C_neg, train_pairwise.py:120
      }
    }
  }
  int tot = npos;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, RR_WAVE);
  acc = rr_wave_sum(acc);
  if (lane == 0) {
    partial[2 * q] = tot > 0 ? acc : 0.f;                          // pair-less queries are skipped, :103-104
    reinterpret_cast<int32_t*>(partial)[2 * q + 1] = 2 * tot;      // num_pairs, train_pairwise.py:106
  }
}

// backward kept separate so lambdas never alias the staged scores
__global__ void __launch_bounds__(RR_WAVE) ranknet_bwd_kernel(const float* __restrict__ score, int64_t sstride,
                                                              const float* __restrict__ targets,
                                                              const int32_t* __restrict__ seg_off, int L, float sigma,
                                                              int mode, const float* __restrict__ gloss,
                                                              float* __restrict__ dscore, int64_t dstride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  if (C <= 0) return;
  float* s = sm;
  float* t = sm + L;
  float* lamv = sm + 2 * L;
  for (int i = lane; i < C; i += RR_WAVE) {
    s[i] = score[static_cast<int64_t>(off + i) * sstride];
    t[i] = targets[off + i];
  }
  wave_sync();
  int npos = 0;
  for (int i = lane; i < C; i += RR_WAVE) {
    const float ti = t[i], si = s[i];
    float lam = 0.f;
    for (int j = 0; j < C; ++j) {
      const float rel = ti - t[j];
      const float x = sigma * (si - s[j]);
      if (rel > 0.f) {
        ++npos;
        lam += -sigma / (1.0f + expf(x));                          // train_pairwise.py:126,128
      } else if (rel < 0.f) {
        lam += sigma / (1.0f + expf(-x));                          // train_pairwise.py:127,128
      }
    }
    lamv[i] = lam;
  }
  int tot = npos;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, RR_WAVE);
  // C_ij is symmetric, so d(sum_ij C_ij)/ds_i = 2 * lambda_i (mode 0); mode 1 = accelerate_grad's row sum
  const float g = tot > 0 ? gloss[0] * (mode == 0 ? 2.0f : 1.0f) : 0.f;
  for (int i = lane; i < C; i += RR_WAVE) dscore[static_cast<int64_t>(off + i) * dstride] = g * lamv[i];
}

// ---------------------------------------------------------------- final fixed-order reductions
__global__ void __launch_bounds__(256) reduce_scale_kernel(const float* __restrict__ partial, int64_t n, int64_t step,
                                                           float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += partial[i * step];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ void __launch_bounds__(256) reduce_pairs_kernel(const float* __restrict__ partial, int64_t n,
                                                           int64_t* __restrict__ out) {
  __shared__ long long red[256];
  long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += reinterpret_cast<const int32_t*>(partial)[2 * i + 1];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ---------------------------------------------------------------- pointwise losses
__global__ void __launch_bounds__(256) pointwise_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                            int64_t stride, const float* __restrict__ targets,
                                                            int64_t n, int gauss, float* __restrict__ partial) {
  __shared__ float red[256];
  const float half_log_2pi = 0.5f * logf(2.0f * 3.14159274101257324f);   // float32(np.pi), loss.py:152,159
  float acc = 0.f;
  const int64_t gs = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += gs) {
    const float d = mean[i * stride] - targets[i];
    if (gauss) {
      const float v = var[i * stride];
      acc += half_log_2pi + 0.5f * logf(v) + (d * d) / (2.0f * v);
    } else {
      acc += d * d;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) pointwise_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                            int64_t stride, const float* __restrict__ targets,
                                                            int64_t n, int gauss, const float* __restrict__ gloss,
                                                            float* __restrict__ dmean, float* __restrict__ dvar,
                                                            int64_t dstride) {
  const float g = gloss[0] / static_cast<float>(n);
  const int64_t gs = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += gs) {
    const float d = mean[i * stride] - targets[i];
    if (gauss) {
      const float v = var[i * stride];
      dmean[i * dstride] = g * d / v;
      dvar[i * dstride] = g * (0.5f / v - (d * d) / (2.0f * v * v));
    } else {
      dmean[i * dstride] = g * 2.0f * d;
    }
  }
}

// ---------------------------------------------------------------- ranking metrics (eval.py:475-555)
__device__ inline double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(RR_WAVE) ranking_metrics_kernel(const float* __restrict__ score, int64_t sstride,
                                                                  const float* __restrict__ targets,
                                                                  const int32_t* __restrict__ seg_off, int L,
                                                                  double ratio, double ndcg_cut,
                                                                  int32_t* __restrict__ order,
                                                                  double* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int off = seg_off[q], C = seg_off[q + 1] - off;
  double* st = stats + static_cast<int64_t>(q) * RR_RANKING_NSTATS;
  if (C <= 0) {
    if (lane < RR_RANKING_NSTATS) st[lane] = 0.0;
    return;
  }
  float* s = sm;
  float* t = sm + L;
  uint16_t* po = reinterpret_cast<uint16_t*>(sm + 2 * L);    // predicted order (list positions < 8192 fit 16 bits)
  uint16_t* to = po + L;                                     // target order
  uint16_t* pr = to + L;                                     // predicted rank of every candidate (inverse of po)
  uint16_t* tr = pr + L;                                     // target rank of every candidate (inverse of to)
  for (int i = lane; i < C; i += RR_WAVE) {
    s[i] = score[static_cast<int64_t>(off + i) * sstride];
    t[i] = targets[off + i];
  }
  wave_sync();
  for (int i = lane; i < C; i += RR_WAVE) {                   // stable descending ranks of both keys
    const float si = s[i], ti = t[i];
    int rp = 0, rt = 0;
    for (int j = 0; j < C; ++j) {
      const float sj = s[j], tj = t[j];
      rp += (sj > si || (sj == si && j < i)) ? 1 : 0;
      rt += (tj > ti || (tj == ti && j < i)) ? 1 : 0;
    }
    po[rp] = static_cast<uint16_t>(i);
    to[rt] = static_cast<uint16_t>(i);
    pr[i] = static_cast<uint16_t>(rp);
    tr[i] = static_cast<uint16_t>(rt);
  }
  wave_sync();
  for (int r = lane; r < C; r += RR_WAVE) order[off + r] = po[r];
  int len25 = static_cast<int>(rint(static_cast<double>(C) * 0.25));   // python round(): half to even (:522)
  if (len25 < 1) len25 = 1;
  // recall@25%: predicted top-len25 that are in the target top-len25
  int hits = 0, hit0 = 0;
  for (int i = lane; i < len25; i += RR_WAVE) {
    const int pi = po[i];
    int in = 0;
    for (int j = 0; j < len25; ++j) in |= (to[j] == pi) ? 1 : 0;
    hits += in;
    if (i == 0) hit0 = in;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { hits += __shfl_xor(hits, o, RR_WAVE); hit0 += __shfl_xor(hit0, o, RR_WAVE); }
  // DCG sums with exp gains (compute_NDCG) over the first len25 / all positions, and exp2 gains over the first 10
  double d25 = 0, i25 = 0, dall = 0, iall = 0, d10 = 0, i10 = 0;
  for (int i = lane; i < C; i += RR_WAVE) {
    const double disc = log2(static_cast<double>(i) + 2.0);
    const double gp = exp(static_cast<double>(t[po[i]])), gt = exp(static_cast<double>(t[to[i]]));
    dall += gp / disc;
    iall += gt / disc;
    if (i < len25) { d25 += gp / disc; i25 += gt / disc; }
    if (i < 10) {
      d10 += (exp2(static_cast<double>(t[po[i]])) - 1.0) / disc;
      i10 += (exp2(static_cast<double>(t[to[i]])) - 1.0) / disc;
    }
  }
  d25 = wave_sum_f64(d25); i25 = wave_sum_f64(i25);
  dall = wave_sum_f64(dall); iall = wave_sum_f64(iall);
  d10 = wave_sum_f64(d10); i10 = wave_sum_f64(i10);
  // evaluate_top_scores (eval.py:76-177) at its `ratio`: cut = python round(C * ratio), at least 1 (:144-146);
  // recall of the predicted top-cut in the target top-cut (:147-151) and the TARGET's first maximum looked up in the
  // predicted top-cut (:156-159).  The first maximum (list.index(max)) is rank 0 of the stable descending order.
  int lenr = static_cast<int>(rint(static_cast<double>(C) * ratio));
  if (lenr < 1) lenr = 1;
  if (lenr > C) lenr = C;
  int hitr = 0;
  for (int i = lane; i < lenr; i += RR_WAVE) {
    const int pi = po[i];
    int in = 0;
    for (int j = 0; j < lenr; ++j) in |= (to[j] == pi) ? 1 : 0;
    hitr += in;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hitr += __shfl_xor(hitr, o, RR_WAVE);
  // calculate_ndcg (eval.py:329-457): KL(softmax(targets) || softmax(scores)) with un-shifted f32 exponentials
  // (:401-404: an overflowing exp gives inf / inf = NaN there and here), and NDCG over the first ceil(C * cut)
  // positions of the TARGET order with rank-derived gains C + 1 - predicted rank (:406-425, cal_NDCG :309-325).
  double se_t = 0, se_s = 0;
  for (int i = lane; i < C; i += RR_WAVE) {
    se_t += static_cast<double>(expf(t[i]));
    se_s += static_cast<double>(expf(s[i]));
  }
  se_t = wave_sum_f64(se_t); se_s = wave_sum_f64(se_s);
  int ncut = static_cast<int>(ceil(static_cast<double>(C) * ndcg_cut));
  if (ncut > C) ncut = C;
  double kl = 0, dcut = 0, icut = 0;
  for (int i = lane; i < C; i += RR_WAVE) {
    const double P = static_cast<double>(expf(t[i])) / se_t, Qd = static_cast<double>(expf(s[i])) / se_s;
    kl += P * log(P / Qd);
    if (i < ncut) {
      // the reference ranks the predictions AFTER putting them in target order (:406-411), so a stable sort breaks
      // tied predictions by target rank, not by list position
      const int cand = to[i];
      const float sc = s[cand];
      int rank = 0;
      for (int j = 0; j < C; ++j) rank += (s[j] > sc || (s[j] == sc && tr[j] < i)) ? 1 : 0;
      const double disc = log2(static_cast<double>(i) + 2.0);
      dcut += static_cast<double>(C - rank) / disc;
      icut += static_cast<double>(C - i) / disc;
    }
  }
  kl = wave_sum_f64(kl); dcut = wave_sum_f64(dcut); icut = wave_sum_f64(icut);
  if (lane == 0) {
    st[8] = (pr[to[0]] < lenr) ? 1.0 : 0.0;
    st[9] = dcut / icut;
    st[10] = kl;
    st[11] = static_cast<double>(hitr) / static_cast<double>(lenr);
    const double p0 = exp(static_cast<double>(t[po[0]])), t0 = exp(static_cast<double>(t[to[0]]));
    double p2 = p0, t2 = t0;                                  // NDCG2: nested lists -> no discount (:543)
    if (C > 1) { p2 += exp(static_cast<double>(t[po[1]])); t2 += exp(static_cast<double>(t[to[1]])); }
    st[0] = (po[0] == to[0]) ? 1.0 : 0.0;
    st[1] = hit0 ? 1.0 : 0.0;
    st[2] = static_cast<double>(hits) / static_cast<double>(len25);
    st[3] = p0 / t0;
    st[4] = p2 / t2;
    st[5] = d25 / i25;
    st[6] = dall / iall;
    st[7] = d10 / i10;
  }
}

// ---------------------------------------------------------------- standalone LogCumsumExp
__global__ void __launch_bounds__(RR_WAVE) lce_fwd_kernel(const float* __restrict__ x, int n, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x;
  float* xs = sm;
  float* fd = sm + n;
  for (int i = lane; i < n; i += RR_WAVE) xs[i] = x[i];
  wave_sync();
  const float m = list_max(xs, n, lane);
  logcumsumexp_rev(xs, fd, n, lane, m);
  for (int i = lane; i < n; i += RR_WAVE) y[i] = fd[i];
}

__global__ void __launch_bounds__(RR_WAVE) lce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ gy, int n, float* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x;
  float* fd = sm;
  float* cs = sm + n;
  for (int i = lane; i < n; i += RR_WAVE) fd[i] = y[i];
  wave_sync();
  cumsum_exp_neg(fd, cs, n, lane);
  for (int i = lane; i < n; i += RR_WAVE) gx[i] = gy[i] * (expf(x[i]) * cs[i]);      // loss.py:59
}

template <typename Kern>
int set_lds(Kern k, size_t bytes) {
  if (bytes > 65536) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
      return RR_ERR_LAUNCH;
  }
  return RR_OK;
}

inline bool list_args_ok(const void* a, const void* t, const int32_t* seg, int Q, int max_len) {
  return a && t && seg && Q >= 0 && max_len >= 0;
}

}  // namespace

extern "C" {

int rr_listmle_fwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, float* loss, float* partial, rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && loss && partial && score_stride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 5u * L * sizeof(float);
  if (Q > 0) {
    if (set_lds(listmle_fwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
    listmle_fwd_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, partial);
  }
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, Q, 1, Q > 0 ? 1.0f / static_cast<float>(Q) : 0.f, loss);
  return rr_launch_status();
}

int rr_listmle_bwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, const float* gloss, float* dscore, int64_t dscore_stride, rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && gloss && dscore && score_stride >= 1 &&
               dscore_stride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (Q == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 5u * L * sizeof(float);
  if (set_lds(listmle_bwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  listmle_bwd_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, Q, gloss, dscore,
                                             dscore_stride);
  return rr_launch_status();
}

int rr_listmle_step_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q, int max_len,
                        float* loss, float* partial, unsigned int* counter, float* dscore, int64_t dscore_stride,
                        rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && loss && partial && counter && dscore && score_stride >= 1 &&
               dscore_stride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (Q == 0) {                                          // nothing to rank: the loss is the empty mean the forward entry point writes
    reduce_scale_kernel<<<1, 256, 0, s>>>(partial, 0, 1, 0.f, loss);
    return rr_launch_status();
  }
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 5u * L * sizeof(float);
  if (set_lds(listmle_step_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  listmle_step_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, Q, partial, dscore, dscore_stride,
                                              1.0f / static_cast<float>(Q), loss, counter);
  return rr_launch_status();
}

int rr_listnet_fwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, int64_t total, float* loss, float* partial, rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && loss && partial && score_stride >= 1 && total >= 0);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 2u * L * sizeof(float);
  if (Q > 0) {
    if (set_lds(listnet_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
    listnet_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, 0, partial, nullptr, 0.f, nullptr,
                                           1);
  }
  // ONE global mean over all candidates (loss.py:347)
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, Q, 1, total > 0 ? 1.0f / static_cast<float>(total) : 0.f, loss);
  return rr_launch_status();
}

int rr_listnet_bwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, int64_t total, const float* gloss, float* dscore, int64_t dscore_stride,
                       rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && gloss && dscore && score_stride >= 1 &&
               dscore_stride >= 1 && total >= 0);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (Q == 0 || total == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 2u * L * sizeof(float);
  if (set_lds(listnet_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  listnet_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, 1, nullptr, gloss,
                                         1.0f / static_cast<float>(total), dscore, dscore_stride);
  return rr_launch_status();
}

int rr_listnet_step_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q, int max_len,
                        int64_t total, float* loss, float* partial, unsigned int* counter, float* dscore, int64_t dscore_stride,
                        rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && loss && partial && counter && dscore && score_stride >= 1 &&
               dscore_stride >= 1 && total >= 0);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (Q == 0 || total == 0) {
    reduce_scale_kernel<<<1, 256, 0, s>>>(partial, 0, 1, 0.f, loss);
    return rr_launch_status();
  }
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 2u * L * sizeof(float);
  if (set_lds(listnet_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  listnet_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, 2, partial, nullptr,
                                         1.0f / static_cast<float>(total), dscore, dscore_stride, loss, counter);
  return rr_launch_status();
}

int rr_evidential_ranking_fwd_f32(const float* mu, const float* var, int64_t stride, const float* targets,
                                  const int32_t* seg_off, int Q, int max_len, float* loss, float* partial,
                                  rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(mu, targets, seg_off, Q, max_len) && var && loss && partial && stride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 3u * L * sizeof(float);
  if (Q > 0) {
    if (set_lds(evidential_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
    evidential_kernel<<<Q, RR_WAVE, lds, s>>>(mu, var, stride, targets, seg_off, L, Q, 0, partial, nullptr, nullptr,
                                              nullptr, 1);
  }
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, Q, 1, Q > 0 ? 1.0f / static_cast<float>(Q) : 0.f, loss);
  return rr_launch_status();
}

int rr_evidential_ranking_bwd_f32(const float* mu, const float* var, int64_t stride, const float* targets,
                                  const int32_t* seg_off, int Q, int max_len, const float* gloss, float* dmu,
                                  float* dvar, int64_t dstride, rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(mu, targets, seg_off, Q, max_len) && var && gloss && dmu && dvar && stride >= 1 &&
               dstride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (Q == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 3u * L * sizeof(float);
  if (set_lds(evidential_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  evidential_kernel<<<Q, RR_WAVE, lds, s>>>(mu, var, stride, targets, seg_off, L, Q, 1, nullptr, gloss, dmu, dvar,
                                            dstride);
  return rr_launch_status();
}

int rr_evidential_ranking_step_f32(const float* mu, const float* var, int64_t stride, const float* targets, const int32_t* seg_off,
                                   int Q, int max_len, float* loss, float* partial, unsigned int* counter, float* dmu, float* dvar,
                                   int64_t dstride, rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(mu, targets, seg_off, Q, max_len) && var && loss && partial && counter && dmu && dvar && stride >= 1 &&
               dstride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (Q == 0) {
    reduce_scale_kernel<<<1, 256, 0, s>>>(partial, 0, 1, 0.f, loss);
    return rr_launch_status();
  }
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 3u * L * sizeof(float);
  if (set_lds(evidential_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  evidential_kernel<<<Q, RR_WAVE, lds, s>>>(mu, var, stride, targets, seg_off, L, Q, 2, partial, nullptr, dmu, dvar, dstride, loss,
                                            counter);
  return rr_launch_status();
}

int rr_ranknet_fwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, float sigma, float* loss_sum, int64_t* pairs, float* partial,
                       rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && loss_sum && pairs && partial && score_stride >= 1);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 2u * L * sizeof(float);
  if (Q > 0) {
    if (set_lds(ranknet_fwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
    ranknet_fwd_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, sigma, partial);
  }
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, Q, 2, 1.0f, loss_sum);
  reduce_pairs_kernel<<<1, 256, 0, s>>>(partial, Q, pairs);
  return rr_launch_status();
}

int rr_ranknet_bwd_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off, int Q,
                       int max_len, float sigma, int mode, const float* gloss, float* dscore, int64_t dscore_stride,
                       rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && gloss && dscore && score_stride >= 1 &&
               dscore_stride >= 1 && (mode == 0 || mode == 1));
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (Q == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 3u * L * sizeof(float);
  if (set_lds(ranknet_bwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  ranknet_bwd_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, sigma, mode, gloss, dscore,
                                             dscore_stride);
  return rr_launch_status();
}

static int pointwise_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return static_cast<int>(b);
}

int64_t rr_pointwise_partial_count(int64_t n) { return pointwise_blocks(n); }

int rr_mse_fwd_f32(const float* pred, int64_t stride, const float* targets, int64_t n, float* loss, float* partial,
                   rr_stream_t stream) {
  RR_CHECK_ARG(pred && targets && loss && partial && n >= 0 && stride >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = pointwise_blocks(n);
  pointwise_fwd_kernel<<<nb, 256, 0, s>>>(pred, nullptr, stride, targets, n, 0, partial);
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, nb, 1, n > 0 ? 1.0f / static_cast<float>(n) : NAN, loss);
  return rr_launch_status();
}

int rr_mse_bwd_f32(const float* pred, int64_t stride, const float* targets, int64_t n, const float* gloss,
                   float* dpred, int64_t dstride, rr_stream_t stream) {
  RR_CHECK_ARG(pred && targets && gloss && dpred && n >= 0 && stride >= 1 && dstride >= 1);
  if (n == 0) return RR_OK;
  pointwise_bwd_kernel<<<pointwise_blocks(n), 256, 0, static_cast<hipStream_t>(stream)>>>(
      pred, nullptr, stride, targets, n, 0, gloss, dpred, nullptr, dstride);
  return rr_launch_status();
}

int rr_gauss_nll_fwd_f32(const float* mean, const float* var, int64_t stride, const float* targets, int64_t n,
                         float* loss, float* partial, rr_stream_t stream) {
  RR_CHECK_ARG(mean && var && targets && loss && partial && n >= 0 && stride >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = pointwise_blocks(n);
  pointwise_fwd_kernel<<<nb, 256, 0, s>>>(mean, var, stride, targets, n, 1, partial);
  reduce_scale_kernel<<<1, 256, 0, s>>>(partial, nb, 1, n > 0 ? 1.0f / static_cast<float>(n) : NAN, loss);
  return rr_launch_status();
}

int rr_gauss_nll_bwd_f32(const float* mean, const float* var, int64_t stride, const float* targets, int64_t n,
                         const float* gloss, float* dmean, float* dvar, int64_t dstride, rr_stream_t stream) {
  RR_CHECK_ARG(mean && var && targets && gloss && dmean && dvar && n >= 0 && stride >= 1 && dstride >= 1);
  if (n == 0) return RR_OK;
  pointwise_bwd_kernel<<<pointwise_blocks(n), 256, 0, static_cast<hipStream_t>(stream)>>>(
      mean, var, stride, targets, n, 1, gloss, dmean, dvar, dstride);
  return rr_launch_status();
}

int rr_ranking_metrics_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off,
                           int Q, int max_len, double ratio, double ndcg_cut, int32_t* order, double* stats,
                           rr_stream_t stream) {
  RR_CHECK_ARG(list_args_ok(score, targets, seg_off, Q, max_len) && order && stats && score_stride >= 1);
  RR_CHECK_ARG(ratio >= 0.0 && ratio <= 1.0 && ndcg_cut >= 0.0 && ndcg_cut <= 1.0);
  if (max_len > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (Q == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = max_len > 0 ? max_len : 1;
  const size_t lds = 2u * L * sizeof(float) + 4u * L * sizeof(uint16_t);
  if (set_lds(ranking_metrics_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  ranking_metrics_kernel<<<Q, RR_WAVE, lds, s>>>(score, score_stride, targets, seg_off, L, ratio, ndcg_cut, order, stats);
  return rr_launch_status();
}

int rr_logcumsumexp_fwd_f32(const float* x, int n, float* y, rr_stream_t stream) {
  RR_CHECK_ARG(x && y && n >= 0);
  if (n > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (n == 0) return RR_OK;
  const size_t lds = 2u * n * sizeof(float);
  if (set_lds(lce_fwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  lce_fwd_kernel<<<1, RR_WAVE, lds, static_cast<hipStream_t>(stream)>>>(x, n, y);
  return rr_launch_status();
}

int rr_logcumsumexp_bwd_f32(const float* x, const float* y, const float* gy, int n, float* gx, rr_stream_t stream) {
  RR_CHECK_ARG(x && y && gy && gx && n >= 0);
  if (n > kMaxLen) return RR_ERR_UNSUPPORTED;
  if (n == 0) return RR_OK;
  const size_t lds = 2u * n * sizeof(float);
  if (set_lds(lce_bwd_kernel, lds) != RR_OK) return RR_ERR_LAUNCH;
  lce_bwd_kernel<<<1, RR_WAVE, lds, static_cast<hipStream_t>(stream)>>>(x, y, gy, n, gx);
  return rr_launch_status();
}

}  // extern "C"
