// Gather / segment kernels of the D-MPNN message-passing path (HBM-bound, no MFMA).
//
// Rows are H contiguous floats (H = 300 -> 75 float4).  Threads are laid out flat over
// (row, float4-column) so a wave always issues full 16-byte lanes and consecutive lanes
// read consecutive addresses of one source row; a molecule's rows sit next to each other
// in memory, so the K neighbour rows of a destination are L2-resident and each source row
// leaves HBM once (algorithmic bytes: DESIGN.md section 4).
#include "rr_common.h"

namespace {

template <int VEC> struct Vec;
template <> struct Vec<4> { using T = f32x4; };
template <> struct Vec<1> { using T = float; };

template <int VEC>
__device__ inline typename Vec<VEC>::T ld(const float* p) {
  return *reinterpret_cast<const typename Vec<VEC>::T*>(p);
}
template <int VEC>
__device__ inline void st(float* p, typename Vec<VEC>::T v) {
  *reinterpret_cast<typename Vec<VEC>::T*>(p) = v;
}

// out[r] = sum_k src[idx[r,k]]   (idx < 0 skipped)
__device__ __attribute__((aligned(16))) const float gather_zero[4] = {0.f, 0.f, 0.f, 0.f};

// Element range of a block for the gather kernels.  Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one, each
// with its own L2): with a plain grid-stride loop the K destinations that read one source row run on different XCDs and
// every XCD pulls its own copy of the row.  Here block b works on the contiguous range number
// (b % 8) * (G / 8) + b / 8, so one XCD covers one contiguous eighth of the rows - a molecule's rows sit next to each other,
// its gathers stay inside one L2 (speed only; any mapping gives the same results).  G % 8 != 0: identity mapping.
#ifndef RR_GATHER_NO_XCD_MAP
__device__ inline void gather_block_range(int gblocks, int64_t total, int64_t* beg, int64_t* end) {
  int vb = static_cast<int>(blockIdx.x);
  if ((gblocks & 7) == 0) vb = (vb & 7) * (gblocks >> 3) + (vb >> 3);
  int64_t per = (total + gblocks - 1) / gblocks;
  per = (per + 255) & ~int64_t(255);                    // whole 256-thread passes: waves stay aligned to 1 KiB
  *beg = static_cast<int64_t>(vb) * per;
  const int64_t e = *beg + per;
  *end = e < total ? e : total;
}
#define RR_GATHER_LOOP(e, gblocks, total)                                              \
  int64_t e##_beg, e##_end;                                                             \
  gather_block_range(gblocks, total, &e##_beg, &e##_end);                               \
  for (int64_t e = e##_beg + threadIdx.x; e < e##_end; e += 256)
#else
#define RR_GATHER_LOOP(e, gblocks, total)                                              \
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<int64_t>(gblocks) * blockDim.x)
#endif

// The element walk of the gather-sum kernels for pad widths 1..4 (16-byte chunks).  A destination chunk costs two dependent
// memory round trips - its index row (L2), then the source rows (HBM) - and a thread walks ~5 chunks of its block's range.
// Here (i) all KT row loads of a chunk are in flight together whatever KT is (the generic loop below serialises the
// K % 4 remainder: with the bond-to-bond table's K = 3 that was three dependent round-trip pairs per chunk), (ii) the
// index row of the thread's NEXT chunk is fetched before the current chunk's rows are waited for, and (iii) (row, column)
// advance by increments instead of a 64-bit division per chunk.  Sums are formed in table order, ((0 + v0) + v1) + ...,
// exactly as before.  pre(r, c) issues whatever else the epilogue reads (its loads fly with the rows); finish(r, c, acc, aux)
// applies it and stores.  Two chunks per thread in flight (e and e + 256 together) were measured and are no faster: with the
// index prefetch the walk is no longer latency-bound (profiles/r03_experiments.txt).
#if !defined(RR_GATHER_NO_XCD_MAP) && !defined(RR_GATHER_NO_WALK)
#define RR_GATHER_WALK 1
template <int KT, bool FROM_ZERO = true, typename PRE, typename FIN>
__device__ __forceinline__ void gather_walk(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx,
                                            int K, int HV, int64_t beg, int64_t end, PRE&& pre, FIN&& finish) {
  int64_t e = beg + threadIdx.x;
  if (e >= end) return;
  const int quo = 256 / HV, rem = 256 - quo * HV;
  int64_t r = e / HV;
  int q = static_cast<int>(e - r * HV);
  int32_t j[KT], jn[KT];
  for (int k = 0; k < KT; ++k) j[k] = idx[r * K + k];
  while (true) {
    int64_t rn = r + quo;
    int qn = q + rem;
    if (qn >= HV) {
      qn -= HV;
      ++rn;
    }
    const bool more = e + 256 < end;
    const int32_t* irn = idx + (more ? rn : r) * K;     // (unconditional loads from a valid row)
    for (int k = 0; k < KT; ++k) jn[k] = irn[k];
    const int c = q * 4;
    f32x4 v[KT];
    for (int k = 0; k < KT; ++k) v[k] = ld<4>(j[k] >= 0 ? src + j[k] * ld_src + c : gather_zero);
    auto aux = pre(r, c);
    f32x4 acc = FROM_ZERO ? f32x4(0.0f) + v[0] : v[0];  // (0 + v0 turns a -0.0 into +0.0, as the sum loop always did)
    for (int k = 1; k < KT; ++k) acc = acc + v[k];
    finish(r, c, acc, aux);
    if (!more) break;
    e += 256;
    r = rn;
    q = qn;
    for (int k = 0; k < KT; ++k) j[k] = jn[k];
  }
}
#endif

// Largest stored magnitude of a launch, for the two-f16-term GEMM that reads the result (rr_linear_args.a1_amax): every
// thread folds what it stores into a running maximum and the workgroup maxes it into its lane of the magnitude slot
// (rr_amax_put) - one atomic per workgroup, fire and forget.  Measured on the way here (profiles/r04_experiments.txt item 10):
// a single float per tensor costs +70-80 % per launch (8,192 atomics on one address, ~5 ns each); checking the float first so
// that only record-breaking workgroups touch it does not help - read at the end, the load waits behind the thread's last
// stores (+12-15 %: ~2 us of idle tail per workgroup); read earlier, 2,048 concurrent workgroups all see the same stale value;
// read with a scalar glc load, the launch takes 3-4 times as long.
__device__ __forceinline__ float amax_fold(float m, f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ float amax_fold(float m, float v) { return fmaxf(m, fabsf(v)); }
__device__ inline void amax_commit(float m, float* out) {      // every thread of the workgroup calls it (out is uniform)
  if (out == nullptr) return;
  __shared__ float amax_part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) amax_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    rr_amax_put(out, fmaxf(fmaxf(amax_part[0], amax_part[1]), fmaxf(amax_part[2], amax_part[3])));
  }
}

// out[r] = sum_k (mask[j_k] > 0 ? src[j_k] * scale : 0): a ReLU / dropout backward (rr_relu_bwd_f32) folded into the gather
// that consumes it - the same products in the same order, so the result equals the two-kernel sequence bit for bit, without
// writing and re-reading the masked tensor.  Used where the masked tensor has no other reader: the shared-prefix reactant
// pass sums each distinct bond's gradient over its (up to 64) copies.
template <int VEC>
__global__ void __launch_bounds__(256) gather_sum_masked_kernel(const float* __restrict__ src, const float* __restrict__ mask,
                                                                int64_t ld_src, const int32_t* __restrict__ idx, int64_t n_out,
                                                                int K, int HV, float scale, float* __restrict__ out,
                                                                int64_t ld_out) {
  using V = typename Vec<VEC>::T;
  const int64_t total = n_out * HV;
  auto masked = [&](V v, V m) -> V {
    if constexpr (VEC == 4) {
      V r;
      r.x = m.x > 0.f ? v.x * scale : 0.f; r.y = m.y > 0.f ? v.y * scale : 0.f;
      r.z = m.z > 0.f ? v.z * scale : 0.f; r.w = m.w > 0.f ? v.w * scale : 0.f;
      return r;
    } else {
      return m > 0.f ? v * scale : 0.f;
    }
  };
  RR_GATHER_LOOP(e, static_cast<int>(gridDim.x), total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * VEC;
    const int32_t* ir = idx + r * K;
    V acc = V(0.0f);
    int k = 0;
    for (; k + 4 <= K; k += 4) {          // 8 independent row loads in flight; pad entries read the zero chunk
      const int32_t j0 = ir[k], j1 = ir[k + 1], j2 = ir[k + 2], j3 = ir[k + 3];
      const V v0 = ld<VEC>(j0 >= 0 ? src + j0 * ld_src + c : gather_zero), m0 = ld<VEC>(j0 >= 0 ? mask + j0 * ld_src + c : gather_zero);
      const V v1 = ld<VEC>(j1 >= 0 ? src + j1 * ld_src + c : gather_zero), m1 = ld<VEC>(j1 >= 0 ? mask + j1 * ld_src + c : gather_zero);
      const V v2 = ld<VEC>(j2 >= 0 ? src + j2 * ld_src + c : gather_zero), m2 = ld<VEC>(j2 >= 0 ? mask + j2 * ld_src + c : gather_zero);
      const V v3 = ld<VEC>(j3 >= 0 ? src + j3 * ld_src + c : gather_zero), m3 = ld<VEC>(j3 >= 0 ? mask + j3 * ld_src + c : gather_zero);
      acc = (((acc + masked(v0, m0)) + masked(v1, m1)) + masked(v2, m2)) + masked(v3, m3);
    }
    for (; k < K; ++k) {
      const int32_t j = ir[k];
      acc = acc + masked(ld<VEC>(j >= 0 ? src + j * ld_src + c : gather_zero), ld<VEC>(j >= 0 ? mask + j * ld_src + c : gather_zero));
    }
    st<VEC>(out + r * ld_out + c, acc);
  }
}

// The same sum over the copies with the mask DERIVED instead of read: the copies' activations are
// dropout_j(y[u]) (rr_gather_dropout_f32), so "activation of copy row j > 0" is "kept(j, c) and y[u, c] > 0" - the keep bit
// comes out of the counter-based stream (element j * H + c, as the forward drew it), y is the small pre-dropout tensor of
// the DESTINATION rows.  Same selects, same products, same order as gather_sum_masked_kernel on the materialised mask,
// without reading that [n_src, H] tensor (half of the kernel's bytes).
__global__ void __launch_bounds__(256) gather_sum_dropmask_kernel(const float* __restrict__ src, int64_t ld_src,
                                                                  const float* __restrict__ y, int64_t ld_y,
                                                                  const int32_t* __restrict__ idx, int64_t n_out, int K, int HV,
                                                                  int H, uint32_t thr, uint64_t seed, float scale,
                                                                  float* __restrict__ out, int64_t ld_out) {
  const int64_t total = n_out * HV;
  RR_GATHER_LOOP(e, static_cast<int>(gridDim.x), total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * 4;
    const int32_t* ir = idx + r * K;
    const f32x4 yv = ld<4>(y + r * ld_y + c);
    auto term = [&](int32_t j, f32x4 v) -> f32x4 {       // (mask > 0 ? v * scale : 0) per element
      uint32_t w = 0u;
      if (thr != 0u) w = rr_hash_group(seed, (static_cast<uint64_t>(j < 0 ? 0 : j) * static_cast<uint64_t>(H) + static_cast<uint64_t>(c)) >> 2);
      f32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool keep = thr == 0u || rr_hash_lane(w, q) >= thr;
        t[q] = (j >= 0 && keep && yv[q] > 0.f) ? v[q] * scale : 0.f;
      }
      return t;
    };
    f32x4 acc = f32x4(0.0f);
    int k = 0;
    for (; k + 4 <= K; k += 4) {          // four row loads in flight; pad entries read the zero chunk
      const int32_t j0 = ir[k], j1 = ir[k + 1], j2 = ir[k + 2], j3 = ir[k + 3];
      const f32x4 v0 = ld<4>(j0 >= 0 ? src + j0 * ld_src + c : gather_zero);
      const f32x4 v1 = ld<4>(j1 >= 0 ? src + j1 * ld_src + c : gather_zero);
      const f32x4 v2 = ld<4>(j2 >= 0 ? src + j2 * ld_src + c : gather_zero);
      const f32x4 v3 = ld<4>(j3 >= 0 ? src + j3 * ld_src + c : gather_zero);
      acc = (((acc + term(j0, v0)) + term(j1, v1)) + term(j2, v2)) + term(j3, v3);
    }
    for (; k < K; ++k) {
      const int32_t j = ir[k];
      acc = acc + term(j, ld<4>(j >= 0 ? src + j * ld_src + c : gather_zero));
    }
    st<4>(out + r * ld_out + c, acc);
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) gather_sum_kernel(const float* __restrict__ src, int64_t ld_src,
                                                         const int32_t* __restrict__ idx, int64_t n_out, int K,
                                                         int HV, float* __restrict__ out, int64_t ld_out,
                                                         const float* __restrict__ row0_partial, int64_t n_partial,
                                                         int64_t ld_partial, float* __restrict__ amax_out) {
  using V = typename Vec<VEC>::T;
  const int64_t total = n_out * HV;
  float am = 0.f;
  // With a padding-row reduction the LAST HV blocks of the grid do that instead of gathering: block j sums column
  // group j of all partial rows (256 threads stride over the rows, then a fixed-order LDS + shuffle tree) and writes
  // out[0, group j] - the reduction runs next to the gather instead of as a straggler thread or an extra launch.
  const int gblocks = row0_partial ? static_cast<int>(gridDim.x) - HV : static_cast<int>(gridDim.x);
  if (row0_partial != nullptr && static_cast<int>(blockIdx.x) >= gblocks) {
    __shared__ float red[256 * VEC];
    const int c = (static_cast<int>(blockIdx.x) - gblocks) * VEC;
    V a0 = V(0.f), a1 = V(0.f);
    int64_t i = threadIdx.x;
    for (; i + 256 < n_partial; i += 512) {
      a0 = a0 + ld<VEC>(row0_partial + i * ld_partial + c);
      a1 = a1 + ld<VEC>(row0_partial + (i + 256) * ld_partial + c);
    }
    if (i < n_partial) a0 = a0 + ld<VEC>(row0_partial + i * ld_partial + c);
    st<VEC>(&red[threadIdx.x * VEC], a0 + a1);
    __syncthreads();
    if (threadIdx.x < 64) {
      V v = (ld<VEC>(&red[threadIdx.x * VEC]) + ld<VEC>(&red[(threadIdx.x + 64) * VEC])) +
            (ld<VEC>(&red[(threadIdx.x + 128) * VEC]) + ld<VEC>(&red[(threadIdx.x + 192) * VEC]));
      if constexpr (VEC == 4) {
        v.x = rr_wave_sum(v.x); v.y = rr_wave_sum(v.y); v.z = rr_wave_sum(v.z); v.w = rr_wave_sum(v.w);
      } else {
        v = rr_wave_sum(v);
      }
      if (threadIdx.x == 0) {
        st<VEC>(out + c, v);
        am = amax_fold(am, v);
      }
    }
    amax_commit(am, amax_out);
    return;
  }
#ifdef RR_GATHER_WALK
  if constexpr (VEC == 4) {
    if (K >= 1 && K <= 4) {
      int64_t wbeg, wend;
      gather_block_range(gblocks, total, &wbeg, &wend);
      const bool skip0 = row0_partial != nullptr;
      auto pre = [](int64_t, int) { return 0; };
      auto fin = [&](int64_t r, int c, f32x4 acc, int) {
        if (!(skip0 && r == 0)) {                                   // (row 0: written by the reduction blocks)
          st<4>(out + r * ld_out + c, acc);
          am = amax_fold(am, acc);
        }
      };
      if (K == 4) gather_walk<4>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else if (K == 3) gather_walk<3>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else if (K == 2) gather_walk<2>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else gather_walk<1>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      amax_commit(am, amax_out);
      return;
    }
  }
#endif
  RR_GATHER_LOOP(e, gblocks, total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * VEC;
    if (r == 0 && row0_partial != nullptr) continue;          // written by the reduction blocks
    const int32_t* ir = idx + r * K;
    V acc = V(0.0f);
    int k = 0;
    // pad entries (index < 0) read a zero chunk: the pointer is selected, the load is unconditional - a load
    // under a per-lane branch makes the compiler drain outstanding loads at the join
    for (; k + 4 <= K; k += 4) {          // 4 independent row loads in flight
      const int32_t j0 = ir[k], j1 = ir[k + 1], j2 = ir[k + 2], j3 = ir[k + 3];
      const V v0 = ld<VEC>(j0 >= 0 ? src + j0 * ld_src + c : gather_zero);
      const V v1 = ld<VEC>(j1 >= 0 ? src + j1 * ld_src + c : gather_zero);
      const V v2 = ld<VEC>(j2 >= 0 ? src + j2 * ld_src + c : gather_zero);
      const V v3 = ld<VEC>(j3 >= 0 ? src + j3 * ld_src + c : gather_zero);
      acc = (((acc + v0) + v1) + v2) + v3;   // k order, like sum(dim=1)
    }
    for (; k < K; ++k) {
      const int32_t j = ir[k];
      acc = acc + ld<VEC>(j >= 0 ? src + j * ld_src + c : gather_zero);
    }
    st<VEC>(out + r * ld_out + c, acc);
    am = amax_fold(am, acc);
  }
  amax_commit(am, amax_out);
}

// out[r] = sum_k (((s0[j_k] + s1[j_k]) + s2[j_k]) + s3[j_k]),  j_k = idx[r, k]: the sum over a distinct bond's copies of a
// per-copy tensor that is itself a sum of NS per-copy tensors (the shared-prefix backward at depth >= 4: d input of a copy =
// the dZ of every per-copy layer).  The inner sums used to be NS - 1 rr_axpby_f32 passes over [n_src, H] (two reads and a
// write each) in front of a one-source gather; here every addend is read once and nothing of [n_src, H] is written.  Same
// additions in the same order as that sequence - the result is bit-identical.
struct GatherSrcs { const float* s[RR_MAX_GATHER_SRCS]; };
template <int NS>
__global__ void __launch_bounds__(256) gather_sum_multi_kernel(GatherSrcs S, int64_t ld_src, const int32_t* __restrict__ idx,
                                                               int64_t n_out, int K, int HV, float* __restrict__ out,
                                                               int64_t ld_out) {
  const int64_t total = n_out * HV;
  RR_GATHER_LOOP(e, static_cast<int>(gridDim.x), total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * 4;
    const int32_t* ir = idx + r * K;
    f32x4 acc = f32x4(0.0f);
    int k = 0;
    for (; k + 2 <= K; k += 2) {          // 2 * NS independent row loads in flight; pad entries read the zero chunk
      const int32_t j0 = ir[k], j1 = ir[k + 1];
      f32x4 v0[NS], v1[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        v0[q] = ld<4>(j0 >= 0 ? S.s[q] + j0 * ld_src + c : gather_zero);
        v1[q] = ld<4>(j1 >= 0 ? S.s[q] + j1 * ld_src + c : gather_zero);
      }
      f32x4 t0 = v0[0], t1 = v1[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) {
        t0 = t0 + v0[q];
        t1 = t1 + v1[q];
      }
      acc = (acc + t0) + t1;
    }
    for (; k < K; ++k) {
      const int32_t j = ir[k];
      f32x4 t = ld<4>(j >= 0 ? S.s[0] + j * ld_src + c : gather_zero);
#pragma unroll
      for (int q = 1; q < NS; ++q) t = t + ld<4>(j >= 0 ? S.s[q] + j * ld_src + c : gather_zero);
      acc = acc + t;
    }
    st<4>(out + r * ld_out + c, acc);
  }
}

// ------------------------------------------------------------------------ gather-sum with a fused epilogue
// out[r] = mask_r (.) (sum_k src[idx[r,k]]) * scale  +  sum_j adds[j][r]
// The backward chain never needs a gathered gradient as such: d message of one layer is consumed masked by the ReLU /
// dropout pattern of the layer below (dZ = d message * (y > 0) / (1 - p)), and the gradient of the residual `input`
// (models/mpn.py:94) is the sum of every iteration's dZ.  Both used to be separate passes over [rows, H] tensors - the mask
// in the operand loader of the next dX GEMM plus a dZ side output written from inside its k-loop, the sum in
// rr_relu_bwd_sum_f32.  Here they ride on the HBM-bound gather that produces the gradient: the GEMMs downstream read dZ
// as a plain operand (no mask, no stores inside their k-loop) and the residual's gradient needs no pass of its own.
// Same products in the same order as the kernels this replaces, so results are bit-identical to that sequence.
constexpr int GMAX = RR_MAX_GATHER_ADDS;
struct GatherEpi {
  const float* mask;   int64_t ld_mask;
  const uint8_t* bits; int64_t bits_row;
  float scale;
  int n_adds;          int64_t ld_add;
  const float* adds[GMAX];
  float* amax_out;
};

// the 4 sign bits of columns c .. c+3 (c % 4 == 0) of row r in a rr_linear_args.mask_bits_out image
__device__ inline uint32_t epi_nibble(const uint8_t* bits, int64_t bits_row, int64_t r, int c) {
  const int blk = c / 304, cc = c - blk * 304;
  const uint32_t b = bits[r * bits_row + blk * 40 + ((cc & 15) >> 3) * 20 + (cc >> 4)];
  return (b >> (cc & 4)) & 0xFu;
}

// what the epilogue reads for one chunk, issued before the gathered rows are waited for (gather_walk's pre())
template <int NADD>
struct EpiAux {
  uint32_t nib;
  f32x4 m;
  f32x4 a[NADD > 0 ? NADD : 1];
};
template <int NADD>
__device__ __forceinline__ EpiAux<NADD> epi_load(const GatherEpi& E, int64_t r, int c) {
  static_assert(NADD >= 0, "run-time addend counts take epi_apply");
  EpiAux<NADD> x;
  x.nib = 0u;
  x.m = f32x4(0.f);
  if (E.bits != nullptr) x.nib = epi_nibble(E.bits, E.bits_row, r, c);
  else if (E.mask != nullptr) x.m = ld<4>(E.mask + r * E.ld_mask + c);
#pragma unroll
  for (int j = 0; j < NADD; ++j) x.a[j] = ld<4>(E.adds[j] + r * E.ld_add + c);
  return x;
}
template <int NADD>
__device__ __forceinline__ f32x4 epi_finish(const GatherEpi& E, f32x4 g, const EpiAux<NADD>& x) {   // = epi_apply on loaded values
  f32x4 v = g;
  if (E.bits != nullptr) {
    v.x = (x.nib & 1u) ? g.x * E.scale : 0.f; v.y = (x.nib & 2u) ? g.y * E.scale : 0.f;
    v.z = (x.nib & 4u) ? g.z * E.scale : 0.f; v.w = (x.nib & 8u) ? g.w * E.scale : 0.f;
  } else if (E.mask != nullptr) {
    v.x = x.m.x > 0.f ? g.x * E.scale : 0.f; v.y = x.m.y > 0.f ? g.y * E.scale : 0.f;
    v.z = x.m.z > 0.f ? g.z * E.scale : 0.f; v.w = x.m.w > 0.f ? g.w * E.scale : 0.f;
  }
  if (NADD == 0) return v;
  f32x4 s = f32x4(0.f);
#pragma unroll
  for (int j = 0; j < NADD; ++j) s = s + x.a[j];
  return s + v;
}

// NADD >= 0: that many addends, unrolled (their loads fly with the gather's); NADD < 0: E.n_adds at run time
template <int NADD>
__device__ inline f32x4 epi_apply(const GatherEpi& E, int64_t r, int c, f32x4 g) {
  f32x4 v = g;
  if (E.bits != nullptr) {
    const uint32_t nib = epi_nibble(E.bits, E.bits_row, r, c);
    v.x = (nib & 1u) ? g.x * E.scale : 0.f; v.y = (nib & 2u) ? g.y * E.scale : 0.f;
    v.z = (nib & 4u) ? g.z * E.scale : 0.f; v.w = (nib & 8u) ? g.w * E.scale : 0.f;
  } else if (E.mask != nullptr) {
    const f32x4 m = ld<4>(E.mask + r * E.ld_mask + c);
    v.x = m.x > 0.f ? g.x * E.scale : 0.f; v.y = m.y > 0.f ? g.y * E.scale : 0.f;
    v.z = m.z > 0.f ? g.z * E.scale : 0.f; v.w = m.w > 0.f ? g.w * E.scale : 0.f;
  }
  if (NADD == 0) return v;
  f32x4 s = f32x4(0.f);                               // fixed order: ((0 + add_0) + add_1) + ... , then + v (rr_relu_bwd_sum_f32's)
  if (NADD > 0) {
#pragma unroll
    for (int j = 0; j < (NADD > 0 ? NADD : 1); ++j) s = s + ld<4>(E.adds[j] + r * E.ld_add + c);
  } else {
    for (int j = 0; j < E.n_adds; ++j) s = s + ld<4>(E.adds[j] + r * E.ld_add + c);
  }
  return s + v;
}

template <int NADD>
__global__ void __launch_bounds__(256) gather_sum_epi_kernel(const float* __restrict__ src, int64_t ld_src,
                                                             const int32_t* __restrict__ idx, int64_t n_out, int K, int HV,
                                                             float* __restrict__ out, int64_t ld_out,
                                                             const float* __restrict__ row0_partial, int64_t n_partial,
                                                             int64_t ld_partial, const GatherEpi E) {
  using V = f32x4;
  const int64_t total = n_out * HV;
  float am = 0.f;
  const int gblocks = row0_partial ? static_cast<int>(gridDim.x) - HV : static_cast<int>(gridDim.x);
  if (row0_partial != nullptr && static_cast<int>(blockIdx.x) >= gblocks) {      // padding-row reduction, see gather_sum_kernel
    __shared__ float red[256 * 4];
    const int c = (static_cast<int>(blockIdx.x) - gblocks) * 4;
    V a0 = V(0.f), a1 = V(0.f);
    int64_t i = threadIdx.x;
    for (; i + 256 < n_partial; i += 512) {
      a0 = a0 + ld<4>(row0_partial + i * ld_partial + c);
      a1 = a1 + ld<4>(row0_partial + (i + 256) * ld_partial + c);
    }
    if (i < n_partial) a0 = a0 + ld<4>(row0_partial + i * ld_partial + c);
    st<4>(&red[threadIdx.x * 4], a0 + a1);
    __syncthreads();
    if (threadIdx.x < 64) {
      V v = (ld<4>(&red[threadIdx.x * 4]) + ld<4>(&red[(threadIdx.x + 64) * 4])) +
            (ld<4>(&red[(threadIdx.x + 128) * 4]) + ld<4>(&red[(threadIdx.x + 192) * 4]));
      v.x = rr_wave_sum(v.x); v.y = rr_wave_sum(v.y); v.z = rr_wave_sum(v.z); v.w = rr_wave_sum(v.w);
      if (threadIdx.x == 0) {
        const f32x4 o = epi_apply<NADD>(E, 0, c, v);
        st<4>(out + c, o);
        am = amax_fold(am, o);
      }
    }
    amax_commit(am, E.amax_out);
    return;
  }
#ifdef RR_GATHER_WALK
  if constexpr (NADD >= 0) {
    if (K >= 1 && K <= 4) {
      int64_t wbeg, wend;
      gather_block_range(gblocks, total, &wbeg, &wend);
      const bool skip0 = row0_partial != nullptr;
      auto pre = [&](int64_t r, int c) { return epi_load<NADD>(E, r, c); };
      auto fin = [&](int64_t r, int c, f32x4 acc, const EpiAux<NADD>& x) {
        if (!(skip0 && r == 0)) {
          const f32x4 o = epi_finish<NADD>(E, acc, x);
          st<4>(out + r * ld_out + c, o);
          am = amax_fold(am, o);
        }
      };
      if (K == 4) gather_walk<4>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else if (K == 3) gather_walk<3>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else if (K == 2) gather_walk<2>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      else gather_walk<1>(src, ld_src, idx, K, HV, wbeg, wend, pre, fin);
      amax_commit(am, E.amax_out);
      return;
    }
  }
#endif
  RR_GATHER_LOOP(e, gblocks, total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * 4;
    if (r == 0 && row0_partial != nullptr) continue;
    const int32_t* ir = idx + r * K;
    V acc = V(0.0f);
    int k = 0;
    for (; k + 4 <= K; k += 4) {
      const int32_t j0 = ir[k], j1 = ir[k + 1], j2 = ir[k + 2], j3 = ir[k + 3];
      const V v0 = ld<4>(j0 >= 0 ? src + j0 * ld_src + c : gather_zero);
      const V v1 = ld<4>(j1 >= 0 ? src + j1 * ld_src + c : gather_zero);
      const V v2 = ld<4>(j2 >= 0 ? src + j2 * ld_src + c : gather_zero);
      const V v3 = ld<4>(j3 >= 0 ? src + j3 * ld_src + c : gather_zero);
      acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; k < K; ++k) {
      const int32_t j = ir[k];
      acc = acc + ld<4>(j >= 0 ? src + j * ld_src + c : gather_zero);
    }
    const f32x4 o = epi_apply<NADD>(E, r, c, acc);
    st<4>(out + r * ld_out + c, o);
    am = amax_fold(am, o);
  }
  amax_commit(am, E.amax_out);
}

// out[r] = sum_{j in [offs[r], offs[r+1])} src[idx[j]]  (CSR form: rows with arbitrarily many sources - the adjoint of a
// gather through a GENERIC index table, where no pad width is known without a device->host sync)
template <int VEC>
__global__ void __launch_bounds__(256) gather_sum_csr_kernel(const float* __restrict__ src, int64_t ld_src,
                                                             const int32_t* __restrict__ offs,
                                                             const int32_t* __restrict__ idx, int64_t n_out, int HV,
                                                             float* __restrict__ out, int64_t ld_out) {
  using V = typename Vec<VEC>::T;
  const int64_t total = n_out * HV;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * VEC;
    const int32_t beg = offs[r], end = offs[r + 1];
    V acc = V(0.0f);
    int32_t j = beg;
    for (; j + 2 <= end; j += 2) {        // two independent row loads in flight, summed in table order
      const V v0 = ld<VEC>(src + static_cast<int64_t>(idx[j]) * ld_src + c);
      const V v1 = ld<VEC>(src + static_cast<int64_t>(idx[j + 1]) * ld_src + c);
      acc = (acc + v0) + v1;
    }
    if (j < end) acc = acc + ld<VEC>(src + static_cast<int64_t>(idx[j]) * ld_src + c);
    st<VEC>(out + r * ld_out + c, acc);
  }
}

// out[r] = a[ia[r]] - m[im[r]]
template <int VEC>
__global__ void __launch_bounds__(256) gather_diff_kernel(const float* __restrict__ a, int64_t ld_a,
                                                          const int32_t* __restrict__ ia,
                                                          const float* __restrict__ m, int64_t ld_m,
                                                          const int32_t* __restrict__ im, int64_t n_out, int HV,
                                                          float* __restrict__ out, int64_t ld_out) {
  using V = typename Vec<VEC>::T;
  const int64_t total = n_out * HV;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * VEC;
    const int32_t ja = ia[r], jm = im[r];
    const V va = ld<VEC>(ja >= 0 ? a + ja * ld_a + c : gather_zero);      // both row loads in flight together
    const V vm = ld<VEC>(jm >= 0 ? m + jm * ld_m + c : gather_zero);
    st<VEC>(out + r * ld_out + c, va - vm);
  }
}

// out[r] = dropout(src[idx[r]]) with the destination row's own mask (index r*H + c)
template <int VEC>
__global__ void __launch_bounds__(256) gather_dropout_kernel(const float* __restrict__ src, int64_t ld_src,
                                                             const int32_t* __restrict__ idx, int64_t n_out, int HV, int H,
                                                             uint32_t thr, float keep_scale, uint64_t seed,
                                                             float* __restrict__ out, int64_t ld_out, float* __restrict__ amax_out) {
  using V = typename Vec<VEC>::T;
  const int64_t total = n_out * HV;
  float am = 0.f;
#ifdef RR_GATHER_WALK
  if constexpr (VEC == 4) {
    int64_t wbeg, wend;
    gather_block_range(static_cast<int>(gridDim.x), total, &wbeg, &wend);
    auto pre = [](int64_t, int) { return 0; };
    auto fin = [&](int64_t r, int c, f32x4 v, int) {
      if (thr != 0u) {                                  // H % 4 == 0 and c % 4 == 0: one aligned group of the mask stream
        const uint64_t base = static_cast<uint64_t>(r) * static_cast<uint64_t>(H) + static_cast<uint64_t>(c);
        const uint32_t w = rr_hash_group(seed, base >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = rr_hash_lane(w, q) >= thr ? v[q] * keep_scale : 0.f;
      }
      st<4>(out + r * ld_out + c, v);
      am = amax_fold(am, v);
    };
    gather_walk<1, false>(src, ld_src, idx, 1, HV, wbeg, wend, pre, fin);
    amax_commit(am, amax_out);
    return;
  }
#endif
  RR_GATHER_LOOP(e, static_cast<int>(gridDim.x), total) {
    const int64_t r = e / HV;
    const int c = static_cast<int>(e - r * HV) * VEC;
    const int32_t j = idx[r];
    V v = ld<VEC>(j >= 0 ? src + j * ld_src + c : gather_zero);
    if (thr != 0u) {
      const uint64_t base = static_cast<uint64_t>(r) * static_cast<uint64_t>(H) + static_cast<uint64_t>(c);
      if constexpr (VEC == 4) {                       // H % 4 == 0 and c % 4 == 0: one aligned group of the mask stream
        const uint32_t w = rr_hash_group(seed, base >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = rr_hash_lane(w, q) >= thr ? v[q] * keep_scale : 0.f;
      } else {
        v = rr_keep(seed, base, thr) ? v * keep_scale : 0.f;
      }
    }
    st<VEC>(out + r * ld_out + c, v);
    am = amax_fold(am, v);
  }
  amax_commit(am, amax_out);
}

// f_bonds[b] = [ f_atoms[b2a[b], 0:Fa] | fbond[b, 0:Fb] | 0 ... ]  (features/featurization.py:198-199: a directed bond's
// feature row is its source atom's features followed by the bond's own) - rebuilt on the device so that only the
// Fb bond columns travel over PCIe with a packed step (the atom half is 3x the bytes and already resident).
__global__ void __launch_bounds__(256) build_fbonds_kernel(const float* __restrict__ f_atoms, int64_t ld_fa, int Fa,
                                                           const int32_t* __restrict__ b2a,
                                                           const float* __restrict__ fbond, int64_t ld_fbb, int Fb,
                                                           int64_t nB, float* __restrict__ out, int64_t ld_out) {
  const int64_t total = nB * ld_out;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t b = e / ld_out;
    const int c = static_cast<int>(e - b * ld_out);
    float v = 0.f;
    if (c < Fa) v = f_atoms[static_cast<int64_t>(b2a[b]) * ld_fa + c];
    else if (c < Fa + Fb) v = fbond[b * ld_fbb + (c - Fa)];
    out[e] = v;
  }
}

// stage 1 of the deterministic weighted column sum: block b sums its row chunk.  256 threads =
// RL row-lanes x CG column groups of 4 floats (H = 300 -> 75 groups x 3 row-lanes), so every lane
// issues 16-byte loads; the row-lanes are combined through LDS in a fixed order.
template <int VEC>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ x, int64_t n, int64_t ldx,
                                                             const float* __restrict__ w, int H,
                                                             int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float red[256 * VEC];
  using V = typename Vec<VEC>::T;
  const int CG = (H + VEC - 1) / VEC;                  // column groups
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  for (int c0 = 0; c0 < CG; c0 += 256) {               // H <= 1024 floats in one pass for VEC = 4
    const int ncg = min(CG - c0, 256);
    const int RL = 256 / ncg;                          // row lanes
    const int rl = threadIdx.x / ncg, cg = threadIdx.x - rl * ncg;
    V a0 = V(0.f), a1 = V(0.f);
    if (rl < RL) {
      const float* xp = x + (c0 + cg) * VEC;
      int64_t r = r0 + rl;
      for (; r + RL < r1; r += 2 * RL) {               // two independent rows in flight
        const float w0 = w ? w[r] : 1.f, w1 = w ? w[r + RL] : 1.f;
        a0 = a0 + ld<VEC>(xp + r * ldx) * w0;
        a1 = a1 + ld<VEC>(xp + (r + RL) * ldx) * w1;
      }
      if (r < r1) a0 = a0 + ld<VEC>(xp + r * ldx) * (w ? w[r] : 1.f);
      a0 = a0 + a1;
    }
    __syncthreads();
    if (rl < RL) st<VEC>(&red[threadIdx.x * VEC], a0);
    __syncthreads();
    if (rl == 0) {
      V acc = ld<VEC>(&red[cg * VEC]);
      for (int k = 1; k < RL; ++k) acc = acc + ld<VEC>(&red[(k * ncg + cg) * VEC]);
      st<VEC>(partial + static_cast<int64_t>(blockIdx.x) * (CG * VEC) + (c0 + cg) * VEC, acc);
    }
  }
}

// stage 2: one wavefront per column sums the block partials (lane-strided, then a shuffle tree:
// fixed order, run-to-run identical).
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int nblocks, int Hp, int H,
                                                           float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= H) return;
  float acc = 0.f;
  for (int b = lane; b < nblocks; b += 64) acc += partial[static_cast<int64_t>(b) * Hp + c];
  acc = rr_wave_sum(acc);
  if (lane == 0) out[c] = accumulate ? out[c] + acc : acc;
}

// out[m, 0:H] = mean of x rows [start, start+size); out[m, H:H+F] = feat[m]; optional dropout.
__global__ void __launch_bounds__(256) segment_mean_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const int32_t* __restrict__ a_scope, int64_t M, int H,
                                                               const float* __restrict__ feat, int F, uint32_t thr,
                                                               float keep_scale, uint64_t seed,
                                                               float* __restrict__ out, int64_t ld_out) {
  const int W = H + F;
  const int64_t total = M * W;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t m = e / W;
    const int c = static_cast<int>(e - m * W);
    float v;
    if (c < H) {
      const int32_t start = a_scope[2 * m], size = a_scope[2 * m + 1];
      // the molecule's rows in groups of 8 independent loads, added in row order (a plain `acc += x[...]` loop over a
      // run-time count is one memory round trip per atom: ~18 in a row per thread)
      float acc = 0.f;
      const float* p = x + static_cast<int64_t>(start) * ldx + c;
      int32_t i = 0;
      for (; i + 8 <= size; i += 8, p += 8 * ldx) {
        const float l0 = p[0], l1 = p[ldx], l2 = p[2 * ldx], l3 = p[3 * ldx];
        const float l4 = p[4 * ldx], l5 = p[5 * ldx], l6 = p[6 * ldx], l7 = p[7 * ldx];
        acc = (((((((acc + l0) + l1) + l2) + l3) + l4) + l5) + l6) + l7;
      }
      if (i + 4 <= size) {
        const float l0 = p[0], l1 = p[ldx], l2 = p[2 * ldx], l3 = p[3 * ldx];
        acc = (((acc + l0) + l1) + l2) + l3;
        i += 4;
        p += 4 * ldx;
      }
      if (i < size) {                                   // 1..3 rows left: loads from clamped rows, adds by count
        const int32_t n = size - i;
        const float l0 = p[0], l1 = p[n > 1 ? ldx : 0], l2 = p[n > 2 ? 2 * ldx : 0];
        acc += l0;
        if (n > 1) acc += l1;
        if (n > 2) acc += l2;
      }
      v = size > 0 ? acc / static_cast<float>(size) : 0.f;
    } else {
      v = feat[m * F + (c - H)];
    }
    if (thr != 0u) v = rr_keep(seed, static_cast<uint64_t>(e), thr) ? v * keep_scale : 0.f;
    out[m * ld_out + c] = v;
  }
}

// dx[a] = dout[mol(a)] * keep/(1-p) / size(mol(a))
__global__ void __launch_bounds__(256) segment_mean_bwd_kernel(const float* __restrict__ dout, int64_t ld_dout,
                                                               const int32_t* __restrict__ a_scope,
                                                               const int32_t* __restrict__ atom2mol, int64_t n_atoms,
                                                               int H, int F, uint32_t thr, float keep_scale,
                                                               uint64_t seed, float* __restrict__ dx, int64_t ldx) {
  const int64_t total = n_atoms * H;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int W = H + F;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t a = e / H;
    const int c = static_cast<int>(e - a * H);
    const int32_t m = atom2mol[a];
    float v = 0.f;
    if (m >= 0) {
      const int32_t size = a_scope[2 * static_cast<int64_t>(m) + 1];
      v = dout[static_cast<int64_t>(m) * ld_dout + c] / static_cast<float>(size);
      if (thr != 0u)
        v = rr_keep(seed, static_cast<uint64_t>(m) * W + c, thr) ? v * keep_scale : 0.f;
    }
    dx[a * ldx + c] = v;
  }
}

// the same, four columns per thread (H % 4 == 0, 16-byte aligned rows): one 16-byte store, and the dropout stream's
// first-level hash once or twice per four elements instead of four times (a molecule's row of the stream starts at
// m * (H + F), which is not a multiple of four, so the four elements straddle at most two hash groups)
__global__ void __launch_bounds__(256) segment_mean_bwd_vec_kernel(const float* __restrict__ dout, int64_t ld_dout,
                                                                   const int32_t* __restrict__ a_scope,
                                                                   const int32_t* __restrict__ atom2mol, int64_t n_atoms,
                                                                   int HV, int W, uint32_t thr, float keep_scale,
                                                                   uint64_t seed, float* __restrict__ dx, int64_t ldx,
                                                                   const float* __restrict__ mask, int64_t ld_mask,
                                                                   const uint8_t* __restrict__ mask_bits, int64_t bits_row,
                                                                   float mask_scale, float* __restrict__ amax_out) {
  const int64_t total = n_atoms * HV;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float am = 0.f;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t a = e / HV;
    const int c = static_cast<int>(e - a * HV) * 4;
    const int32_t m = atom2mol[a];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m >= 0) {
      const float size = static_cast<float>(a_scope[2 * static_cast<int64_t>(m) + 1]);
      const float* d = dout + static_cast<int64_t>(m) * ld_dout + c;
      v = make_float4(d[0] / size, d[1] / size, d[2] / size, d[3] / size);
      if (thr != 0u) {
        const uint64_t i0 = static_cast<uint64_t>(m) * W + c;
        const uint32_t w0 = rr_hash_group(seed, i0 >> 2);
        const uint32_t w1 = (i0 & 3u) ? rr_hash_group(seed, (i0 >> 2) + 1) : w0;
        float* pv = &v.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint64_t i = i0 + k;
          const uint32_t w = (i >> 2) == (i0 >> 2) ? w0 : w1;
          pv[k] = rr_hash_lane(w, static_cast<uint32_t>(i) & 3u) >= thr ? pv[k] * keep_scale : 0.f;
        }
      }
    }
    // optional: the ReLU / dropout backward of the layer that produced the readout's input (dZ = dx * (y > 0) * scale),
    // so the dX GEMMs downstream read a plain operand
    if (mask_bits != nullptr) {
      const uint32_t nib = epi_nibble(mask_bits, bits_row, a, c);
      v.x = (nib & 1u) ? v.x * mask_scale : 0.f; v.y = (nib & 2u) ? v.y * mask_scale : 0.f;
      v.z = (nib & 4u) ? v.z * mask_scale : 0.f; v.w = (nib & 8u) ? v.w * mask_scale : 0.f;
    } else if (mask != nullptr) {
      const float4 m = *reinterpret_cast<const float4*>(mask + a * ld_mask + c);
      v.x = m.x > 0.f ? v.x * mask_scale : 0.f; v.y = m.y > 0.f ? v.y * mask_scale : 0.f;
      v.z = m.z > 0.f ? v.z * mask_scale : 0.f; v.w = m.w > 0.f ? v.w * mask_scale : 0.f;
    }
    *reinterpret_cast<float4*>(dx + a * ldx + c) = v;
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  amax_commit(am, amax_out);
}

}  // namespace

extern "C" {

// row0_partial may be NULL (no padding-row reduction), amax_out may be NULL
int rr_gather_sum_amax_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K,
                           int H, const float* row0_partial, int64_t n_partial, int64_t ld_partial, float* out,
                           int64_t ld_out, float* amax_out, rr_stream_t stream) {
  RR_CHECK_ARG(src && idx && out && n_src >= 0 && n_out >= 0 && K >= 1 && H >= 1 && ld_src >= H && ld_out >= H);
  RR_CHECK_ARG(!row0_partial || (n_out >= 1 && n_partial >= 0 && ld_partial >= H));
  if (n_out == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(src) && rr_aligned16(out);
  if (row0_partial) vec = vec && (ld_partial % 4 == 0) && rr_aligned16(row0_partial);
  if (vec) {
    const int HV = H / 4;
    gather_sum_kernel<4><<<rr_grid_for(n_out * HV, 256) + (row0_partial ? HV : 0), 256, 0, s>>>(
        src, ld_src, idx, n_out, K, HV, out, ld_out, row0_partial, n_partial, ld_partial, amax_out);
  } else {
    gather_sum_kernel<1><<<rr_grid_for(n_out * H, 256) + (row0_partial ? H : 0), 256, 0, s>>>(
        src, ld_src, idx, n_out, K, H, out, ld_out, row0_partial, n_partial, ld_partial, amax_out);
  }
  return rr_launch_status();
}

int rr_gather_sum_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K,
                      int H, float* out, int64_t ld_out, rr_stream_t stream) {
  return rr_gather_sum_amax_f32(src, n_src, ld_src, idx, n_out, K, H, nullptr, 0, 0, out, ld_out, nullptr, stream);
}

int rr_gather_sum_multi_f32(const float* const* srcs, int n_srcs, int64_t n_src, int64_t ld_src, const int32_t* idx,
                            int64_t n_out, int K, int H, float* out, int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(srcs && idx && out && n_srcs >= 1 && n_srcs <= RR_MAX_GATHER_SRCS && n_src >= 0 && n_out >= 0 && K >= 1 && H >= 1 &&
               ld_src >= H && ld_out >= H);
  GatherSrcs S{};
  bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(out);
  for (int q = 0; q < n_srcs; ++q) {
    RR_CHECK_ARG(srcs[q] != nullptr);
    S.s[q] = srcs[q];
    vec = vec && rr_aligned16(srcs[q]);
  }
  if (!vec) return RR_ERR_ALIGN;                        // (16-byte chunks only: callers fall back to rr_axpby_f32 + rr_gather_sum_f32)
  if (n_out == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int HV = H / 4;
  const int grid = rr_grid_for(n_out * HV, 256);
  switch (n_srcs) {
    case 1: gather_sum_multi_kernel<1><<<grid, 256, 0, s>>>(S, ld_src, idx, n_out, K, HV, out, ld_out); break;
    case 2: gather_sum_multi_kernel<2><<<grid, 256, 0, s>>>(S, ld_src, idx, n_out, K, HV, out, ld_out); break;
    case 3: gather_sum_multi_kernel<3><<<grid, 256, 0, s>>>(S, ld_src, idx, n_out, K, HV, out, ld_out); break;
    default: gather_sum_multi_kernel<4><<<grid, 256, 0, s>>>(S, ld_src, idx, n_out, K, HV, out, ld_out); break;
  }
  return rr_launch_status();
}

int rr_gather_sum_padrow_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K,
                             int H, const float* row0_partial, int64_t n_partial, int64_t ld_partial, float* out,
                             int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(row0_partial && n_out >= 1);
  return rr_gather_sum_amax_f32(src, n_src, ld_src, idx, n_out, K, H, row0_partial, n_partial, ld_partial, out, ld_out, nullptr, stream);
}

int rr_gather_sum_epi_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K, int H,
                          const float* row0_partial, int64_t n_partial, int64_t ld_partial, const rr_gather_epi* epi,
                          float* out, int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(src && idx && out && epi && n_src >= 0 && n_out >= 1 && K >= 1 && H >= 1 && ld_src >= H && ld_out >= H);
  RR_CHECK_ARG(!row0_partial || (n_partial >= 0 && ld_partial >= H));
  RR_CHECK_ARG(epi->n_adds >= 0 && epi->n_adds <= RR_MAX_GATHER_ADDS && (epi->n_adds == 0 || epi->ld_add >= H));
  RR_CHECK_ARG(!epi->mask || epi->ld_mask >= H);
  // the fused form exists for the straight-line geometry only (what the packer and the step plans produce)
  bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(src) && rr_aligned16(out);
  if (row0_partial) vec = vec && (ld_partial % 4 == 0) && rr_aligned16(row0_partial);
  if (epi->mask && !epi->mask_bits) vec = vec && (epi->ld_mask % 4 == 0) && rr_aligned16(epi->mask);
  GatherEpi E;
  E.mask = epi->mask_bits ? nullptr : epi->mask;
  E.ld_mask = epi->ld_mask;
  E.bits = epi->mask_bits;
  E.bits_row = rr_mask_bits_row_bytes(H);
  E.scale = epi->mask_scale;
  E.n_adds = epi->n_adds;
  E.ld_add = epi->ld_add;
  E.amax_out = epi->amax_out;
  for (int j = 0; j < GMAX; ++j) {
    E.adds[j] = j < epi->n_adds ? epi->adds[j] : nullptr;
    if (j < epi->n_adds) {
      RR_CHECK_ARG(epi->adds[j]);
      vec = vec && (epi->ld_add % 4 == 0) && rr_aligned16(epi->adds[j]);
    }
  }
  if (!vec) return RR_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int HV = H / 4;
  const int grid = rr_grid_for(n_out * HV, 256) + (row0_partial ? HV : 0);
#define RR_EPI_LAUNCH(NADD)                                                                                          \
  gather_sum_epi_kernel<NADD><<<grid, 256, 0, s>>>(src, ld_src, idx, n_out, K, HV, out, ld_out, row0_partial, n_partial, \
                                                   ld_partial, E)
  // The last gather of a backward pass adds the dZ of every earlier iteration: n_adds = depth - 1 there, 0 elsewhere.
  // Depths 1..6 (the reference's default is 3, BASELINE configs[4] runs 6) get the unrolled form - addends prefetched
  // beside the gather walk; deeper models take the generic form (run-time addend loop after the gather: same result).
  switch (epi->n_adds) {
    case 0: RR_EPI_LAUNCH(0); break;
    case 1: RR_EPI_LAUNCH(1); break;
    case 2: RR_EPI_LAUNCH(2); break;
    case 3: RR_EPI_LAUNCH(3); break;
    case 4: RR_EPI_LAUNCH(4); break;
    case 5: RR_EPI_LAUNCH(5); break;
    default: RR_EPI_LAUNCH(-1); break;
  }
#undef RR_EPI_LAUNCH
  return rr_launch_status();
}

int rr_gather_sum_masked_f32(const float* src, const float* mask, int64_t n_src, int64_t ld_src, const int32_t* idx,
                             int64_t n_out, int K, int H, float scale, float* out, int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(src && mask && idx && out && n_src >= 0 && n_out >= 1 && K >= 1 && H >= 1 && ld_src >= H && ld_out >= H);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(src) && rr_aligned16(mask) &&
                   rr_aligned16(out);
  if (vec) {
    const int HV = H / 4;
    gather_sum_masked_kernel<4><<<rr_grid_for(n_out * HV, 256), 256, 0, s>>>(src, mask, ld_src, idx, n_out, K, HV, scale, out,
                                                                             ld_out);
  } else {
    gather_sum_masked_kernel<1><<<rr_grid_for(n_out * H, 256), 256, 0, s>>>(src, mask, ld_src, idx, n_out, K, H, scale, out,
                                                                            ld_out);
  }
  return rr_launch_status();
}

int rr_gather_sum_dropmask_f32(const float* src, int64_t n_src, int64_t ld_src, const float* y, int64_t ld_y,
                               const int32_t* idx, int64_t n_out, int K, int H, float drop_p, uint64_t drop_seed, float scale,
                               float* out, int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(src && y && idx && out && n_src >= 0 && n_out >= 1 && K >= 1 && H >= 1 && ld_src >= H && ld_y >= H && ld_out >= H);
  RR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  if (!(H % 4 == 0 && ld_src % 4 == 0 && ld_y % 4 == 0 && ld_out % 4 == 0 && rr_aligned16(src) && rr_aligned16(y) && rr_aligned16(out)))
    return RR_ERR_ALIGN;                                // (the dropout stream is hashed per aligned group of four elements)
  const int HV = H / 4;
  gather_sum_dropmask_kernel<<<rr_grid_for(n_out * HV, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      src, ld_src, y, ld_y, idx, n_out, K, HV, H, rr_drop_threshold(drop_p), drop_seed, scale, out, ld_out);
  return rr_launch_status();
}

int rr_gather_sum_csr_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* offsets, const int32_t* idx,
                          int64_t n_out, int H, float* out, int64_t ld_out, rr_stream_t stream) {
  RR_CHECK_ARG(src && offsets && out && n_src >= 0 && n_out >= 0 && H >= 1 && ld_src >= H && ld_out >= H);
  if (n_out == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(src) && rr_aligned16(out);
  if (vec) {
    const int HV = H / 4;
    gather_sum_csr_kernel<4><<<rr_grid_for(n_out * HV, 256), 256, 0, s>>>(src, ld_src, offsets, idx, n_out, HV, out,
                                                                          ld_out);
  } else {
    gather_sum_csr_kernel<1><<<rr_grid_for(n_out * H, 256), 256, 0, s>>>(src, ld_src, offsets, idx, n_out, H, out,
                                                                         ld_out);
  }
  return rr_launch_status();
}

int rr_gather_diff_f32(const float* a, int64_t n_a, int64_t ld_a, const int32_t* ia, const float* m, int64_t n_m,
                       int64_t ld_m, const int32_t* im, int64_t n_out, int H, float* out, int64_t ld_out,
                       rr_stream_t stream) {
  RR_CHECK_ARG(a && ia && m && im && out && n_a >= 0 && n_m >= 0 && n_out >= 0 && H >= 1 && ld_a >= H &&
               ld_m >= H && ld_out >= H);
  if (n_out == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (H % 4 == 0) && (ld_a % 4 == 0) && (ld_m % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(a) &&
                   rr_aligned16(m) && rr_aligned16(out);
  if (vec) {
    const int HV = H / 4;
    gather_diff_kernel<4><<<rr_grid_for(n_out * HV, 256), 256, 0, s>>>(a, ld_a, ia, m, ld_m, im, n_out, HV, out,
                                                                       ld_out);
  } else {
    gather_diff_kernel<1><<<rr_grid_for(n_out * H, 256), 256, 0, s>>>(a, ld_a, ia, m, ld_m, im, n_out, H, out,
                                                                      ld_out);
  }
  return rr_launch_status();
}

int rr_gather_dropout_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int H,
                          float drop_p, uint64_t drop_seed, float* out, int64_t ld_out, rr_stream_t stream) {
  return rr_gather_dropout_amax_f32(src, n_src, ld_src, idx, n_out, H, drop_p, drop_seed, out, ld_out, nullptr, stream);
}

int rr_gather_dropout_amax_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int H,
                               float drop_p, uint64_t drop_seed, float* out, int64_t ld_out, float* amax_out,
                               rr_stream_t stream) {
  RR_CHECK_ARG(src && idx && out && n_src >= 0 && n_out >= 0 && H >= 1 && ld_src >= H && ld_out >= H);
  RR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  if (n_out == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t thr = rr_drop_threshold(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  const bool vec = (H % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && rr_aligned16(src) && rr_aligned16(out);
  if (vec) {
    gather_dropout_kernel<4><<<rr_grid_for(n_out * (H / 4), 256), 256, 0, s>>>(src, ld_src, idx, n_out, H / 4, H, thr, ks,
                                                                             drop_seed, out, ld_out, amax_out);
  } else {
    gather_dropout_kernel<1><<<rr_grid_for(n_out * H, 256), 256, 0, s>>>(src, ld_src, idx, n_out, H, H, thr, ks,
                                                                         drop_seed, out, ld_out, amax_out);
  }
  return rr_launch_status();
}

int rr_build_fbonds_f32(const float* f_atoms, int64_t n_atoms, int64_t ld_fa, int atom_fdim, const int32_t* b2a,
                        const float* fbond, int64_t ld_fbb, int bond_fdim, int64_t n_bonds, float* out, int64_t ld_out,
                        rr_stream_t stream) {
  RR_CHECK_ARG(f_atoms && b2a && fbond && out && n_atoms >= 1 && n_bonds >= 0 && atom_fdim >= 1 && bond_fdim >= 0);
  RR_CHECK_ARG(ld_fa >= atom_fdim && ld_fbb >= bond_fdim && ld_out >= atom_fdim + bond_fdim);
  if (n_bonds == 0) return RR_OK;
  build_fbonds_kernel<<<rr_grid_for(n_bonds * ld_out, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      f_atoms, ld_fa, atom_fdim, b2a, fbond, ld_fbb, bond_fdim, n_bonds, out, ld_out);
  return rr_launch_status();
}

static int colsum_blocks(int64_t n) {
  // HBM-bound streaming reduction: 8 blocks per CU keep enough 16-byte loads in flight (with 2 per CU the pass ran
  // at ~2 TB/s); >= 64 rows per block so the partial slab stays small against the input
  int64_t b = (n + 63) / 64;
  if (b < 1) b = 1;
  if (b > RR_NUM_CU * 8) b = RR_NUM_CU * 8;
  return static_cast<int>(b);
}

size_t rr_colsum_workspace_bytes(int64_t n, int H) {
  if (n < 0 || H < 1) return 0;
  return static_cast<size_t>(colsum_blocks(n)) * static_cast<size_t>((H + 3) / 4 * 4) * sizeof(float);
}

int rr_weighted_colsum_f32(const float* x, int64_t n, int64_t ld, const float* w, int H, float* out, int accumulate,
                           void* workspace, size_t workspace_bytes, rr_stream_t stream) {
  RR_CHECK_ARG(x && out && workspace && n >= 0 && H >= 1 && ld >= H);
  if (workspace_bytes < rr_colsum_workspace_bytes(n, H)) return RR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = colsum_blocks(n);
  int64_t rpb = (n + nb - 1) / nb;
  if (rpb < 1) rpb = 1;
  float* partial = static_cast<float*>(workspace);
  const bool vec = (H % 4 == 0) && (ld % 4 == 0) && rr_aligned16(x) && rr_aligned16(workspace);
  const int Hp = (H + 3) / 4 * 4;
  if (vec) colsum_partial_kernel<4><<<nb, 256, 0, s>>>(x, n, ld, w, H, rpb, partial);
  else colsum_partial_kernel<1><<<nb, 256, 0, s>>>(x, n, ld, w, H, rpb, partial);
  colsum_final_kernel<<<(H + 3) / 4, 256, 0, s>>>(partial, nb, vec ? Hp : H, H, out, accumulate);
  return rr_launch_status();
}

int rr_segment_mean_fwd_f32(const float* x, int64_t ldx, const int32_t* a_scope, int64_t M, int H, const float* feat,
                            int F, float drop_p, uint64_t drop_seed, float* out, int64_t ld_out,
                            rr_stream_t stream) {
  RR_CHECK_ARG(x && a_scope && out && M >= 0 && H >= 1 && F >= 0 && ldx >= H && ld_out >= H + F);
  RR_CHECK_ARG(F == 0 || feat);
  RR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  if (M == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t thr = rr_drop_threshold(drop_p);
  segment_mean_fwd_kernel<<<rr_grid_for(M * (H + F), 256), 256, 0, s>>>(x, ldx, a_scope, M, H, feat, F, thr,
                                                                         1.0f / (1.0f - drop_p), drop_seed, out,
                                                                         ld_out);
  return rr_launch_status();
}

int rr_segment_mean_bwd_masked_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope, const int32_t* atom2mol,
                                   int64_t n_atoms, int H, int F, float drop_p, uint64_t drop_seed, const float* mask,
                                   int64_t ld_mask, const uint8_t* mask_bits, float mask_scale, float* dx, int64_t ldx,
                                   rr_stream_t stream) {
  return rr_segment_mean_bwd_masked_amax_f32(dout, ld_dout, a_scope, atom2mol, n_atoms, H, F, drop_p, drop_seed, mask, ld_mask,
                                             mask_bits, mask_scale, dx, ldx, nullptr, stream);
}

int rr_segment_mean_bwd_masked_amax_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope, const int32_t* atom2mol,
                                        int64_t n_atoms, int H, int F, float drop_p, uint64_t drop_seed, const float* mask,
                                        int64_t ld_mask, const uint8_t* mask_bits, float mask_scale, float* dx, int64_t ldx,
                                        float* amax_out, rr_stream_t stream) {
  RR_CHECK_ARG(dout && a_scope && atom2mol && dx && n_atoms >= 0 && H >= 1 && F >= 0 && ld_dout >= H && ldx >= H);
  RR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  RR_CHECK_ARG((mask || mask_bits) && (!mask || ld_mask >= H));
  if (n_atoms == 0) return RR_OK;
  if (!(H % 4 == 0 && ldx % 4 == 0 && rr_aligned16(dx))) return RR_ERR_ALIGN;
  if (!mask_bits && !(ld_mask % 4 == 0 && rr_aligned16(mask))) return RR_ERR_ALIGN;
  // (A two-launch form - the per-molecule division and dropout once per molecule, then a K = 1 masked gather over
  // atom2mol - was measured in round 4: +0.2 % on the step.  This kernel writes 86 MB at 71k atoms in ~45 us, within
  // 1.6x of the write-only HBM rate, so its per-atom arithmetic is not what the step waits for.)
  segment_mean_bwd_vec_kernel<<<rr_grid_for(n_atoms * (H / 4), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      dout, ld_dout, a_scope, atom2mol, n_atoms, H / 4, H + F, rr_drop_threshold(drop_p), 1.0f / (1.0f - drop_p), drop_seed,
      dx, ldx, mask_bits ? nullptr : mask, ld_mask, mask_bits, rr_mask_bits_row_bytes(H), mask_scale, amax_out);
  return rr_launch_status();
}

int rr_segment_mean_bwd_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope, const int32_t* atom2mol,
                            int64_t n_atoms, int H, int F, float drop_p, uint64_t drop_seed, float* dx, int64_t ldx,
                            rr_stream_t stream) {
  RR_CHECK_ARG(dout && a_scope && atom2mol && dx && n_atoms >= 0 && H >= 1 && F >= 0 && ld_dout >= H && ldx >= H);
  RR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  if (n_atoms == 0) return RR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t thr = rr_drop_threshold(drop_p);
  if (H % 4 == 0 && ldx % 4 == 0 && rr_aligned16(dx)) {
    segment_mean_bwd_vec_kernel<<<rr_grid_for(n_atoms * (H / 4), 256), 256, 0, s>>>(
        dout, ld_dout, a_scope, atom2mol, n_atoms, H / 4, H + F, thr, 1.0f / (1.0f - drop_p), drop_seed, dx, ldx,
        nullptr, 0, nullptr, 0, 1.0f, nullptr);
  } else {
    segment_mean_bwd_kernel<<<rr_grid_for(n_atoms * H, 256), 256, 0, s>>>(dout, ld_dout, a_scope, atom2mol, n_atoms, H,
                                                                          F, thr, 1.0f / (1.0f - drop_p), drop_seed, dx,
                                                                          ldx);
  }
  return rr_launch_status();
}

}  // extern "C"
