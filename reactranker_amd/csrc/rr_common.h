// Shared host/device helpers for the gfx950 kernels behind include/reactranker_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/reactranker_hip.h"

#define RR_ABI_VERSION 1
#define RR_WAVE 64
#define RR_NUM_CU 256   // MI355X: 8 XCDs x 32 CUs

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RR_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return RR_ERR_ARG; \
  } while (0)

static inline int rr_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RR_OK : RR_ERR_LAUNCH;
}

static inline bool rr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Grid cap of the grid-stride streaming kernels.  Every pass of such a loop ends in a store and starts with
// a load whose s_waitcnt also retires that store (vmcnt counts loads and stores in issue order), so a thread
// pays a write round trip per pass: few passes per thread (32 blocks per CU) measured 1.8 % faster on the whole
// training step than 8 blocks per CU, and the same as one pass per thread.
#ifndef RR_GRID_CAP
#define RR_GRID_CAP (RR_NUM_CU * 32)
#endif
static inline int rr_grid_for(int64_t work_items, int block, int max_blocks = RR_GRID_CAP) {
  int64_t b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return static_cast<int>(b);
}

// Counter-based dropout stream: element `index` of stream `seed` is kept iff
// hash(seed, index) >= p * 2^32.  32-bit murmur3-style mixing (cheap enough for a GEMM
// epilogue); oracle/dropout_ref.py restates it bit for bit in numpy.
__host__ __device__ static inline uint32_t rr_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ static inline uint32_t rr_hash_u32(uint64_t seed, uint64_t index) {
  const uint32_t lo = static_cast<uint32_t>(index), hi = static_cast<uint32_t>(index >> 32);
  const uint32_t s0 = static_cast<uint32_t>(seed), s1 = static_cast<uint32_t>(seed >> 32);
  uint32_t h = rr_fmix32(lo ^ s0);
  h = rr_fmix32(h + hi * 0x9E3779B1u + s1);
  return h;
}
__host__ __device__ static inline uint32_t rr_drop_threshold(float p) {
  double t = static_cast<double>(p) * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return static_cast<uint32_t>(t);
}
__host__ __device__ static inline bool rr_keep(uint64_t seed, uint64_t index, uint32_t threshold) {
  return rr_hash_u32(seed, index) >= threshold;
}

// torch.nn.Softplus(beta=1, threshold=20): x > 20 -> x, else log1p(exp(x)).
__device__ static inline float rr_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ static inline float rr_softplus_grad(float x) { return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)); }

// ---- wave64 reductions / scans (all 64 lanes must be active) ---------------------------
__device__ static inline float rr_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ static inline float rr_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// inclusive prefix sum across lanes (lane 0 .. lane 63)
__device__ static inline float rr_wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
