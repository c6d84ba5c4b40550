// Shared host/device helpers for the gfx950 kernels behind include/reactranker_hip.h.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else            // host-only translation unit (the AddressSanitizer build of pack.cpp: `make asan`, plain g++)
#define __host__
#define __device__
#endif
#include <stdint.h>

#include "../../include/reactranker_hip.h"

#define RR_ABI_VERSION 8
#define RR_WAVE 64
#define RR_NUM_CU 256   // MI355X: 8 XCDs x 32 CUs

#if defined(__HIPCC__)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

#define RR_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return RR_ERR_ARG; \
  } while (0)

#if defined(__HIPCC__)
static inline int rr_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RR_OK : RR_ERR_LAUNCH;
}
#endif

// A magnitude slot (rr_linear_args.a1_amax, c_amax_out, rr_gather_epi.amax_out ...) is RR_AMAX_LANES floats RR_AMAX_STRIDE
// floats apart - one 128-byte line each - and the bound is their maximum: a producer's workgroups max their values into lane
// blockIdx.x % RR_AMAX_LANES, so a launch of 8,192 workgroups queues 512 atomics per address instead of 8,192 on one (an
// atomic on a busy address costs ~5 ns: measured +70-80 % on a gather launch with a single float per tensor).
#ifdef __HIPCC__
__device__ __forceinline__ float rr_amax_read(const float* slot) {        // (uniform address: scalar loads)
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < RR_AMAX_LANES; ++i) m = fmaxf(m, slot[i * RR_AMAX_STRIDE]);
  return m;
}
__device__ __forceinline__ void rr_amax_put(float* slot, float m) {       // one thread per workgroup
  if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(slot + RR_AMAX_STRIDE * (blockIdx.x & (RR_AMAX_LANES - 1))), __float_as_uint(m));
}
#endif

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

// Counter-based dropout stream: element `index` of stream `seed` is kept iff hash(seed, index) >= p * 2^32.
// The hash is two-level so that a GEMM epilogue, which owns 4 consecutive elements per lane, pays the two
// murmur3 finaliser rounds once per aligned group of 4 (index >> 2) and one multiply per element:
//   w = fmix32(fmix32(lo(g) ^ s0) + hi(g) * 0x9E3779B1 + s1),  g = index >> 2
//   h = (w ^ K[index & 3]) * 0x85EBCA6B;  h ^= h >> 15
// (v_mul_lo_u32 is quarter rate: the one-level form cost ~20 multiplies per 4 elements, ~15 us per round of a
// dropout-carrying GEMM; this one costs 9).  oracle/dropout_ref.py restates it bit for bit in numpy.
__host__ __device__ static inline uint32_t rr_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// first level: one word per aligned group of 4 elements
__host__ __device__ static inline uint32_t rr_hash_group(uint64_t seed, uint64_t group) {
  const uint32_t lo = static_cast<uint32_t>(group), hi = static_cast<uint32_t>(group >> 32);
  const uint32_t s0 = static_cast<uint32_t>(seed), s1 = static_cast<uint32_t>(seed >> 32);
  uint32_t h = rr_fmix32(lo ^ s0);
  h = rr_fmix32(h + hi * 0x9E3779B1u + s1);
  return h;
}
// second level: element e (0..3) of the group
__host__ __device__ static inline uint32_t rr_hash_lane(uint32_t w, uint32_t e) {
  uint32_t h = (w ^ (e * 0x9E3779B9u + 0x7F4A7C15u)) * 0x85EBCA6Bu;
  h ^= h >> 15;
  return h;
}
__host__ __device__ static inline uint32_t rr_hash_u32(uint64_t seed, uint64_t index) {
  return rr_hash_lane(rr_hash_group(seed, index >> 2), static_cast<uint32_t>(index) & 3u);
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

#if defined(__HIPCC__)
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
#endif  // __HIPCC__
