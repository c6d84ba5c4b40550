// Dense layers of the D-MPNN path: f32 GEMMs with fused prologues / epilogues, in two arithmetic forms -
//   * linear_fast_kernel / wgrad_fast_kernel: every product on the exact-f32 matrix core (v_mfma_f32_16x16x4_f32);
//     the FFN head, shapes off the fast path, and every GEMM under RR_PLAN_F32_GEMM;
//   * linear_split_kernel / wgrad_split_kernel (further down): the encoder's GEMMs on the bf16 matrix core through
//     exact three-term operand splits (x = x0 + x1 + x2 in bf16, six products per f32 multiply, f32 accumulation).
//
//   rr_linear_f32        C = dropout(act(residual + bias + [A1|A2] * W^T))   (forward, and dX with W^T)
//   rr_linear_wgrad_f32  dW = dZ^T * [X1|X2],  dbias = colsum(dZ)            (weight gradients)
//
// Both fuse the reference's surrounding ATen ops into the operand load / epilogue:
// row gather + subtract (a_message[b2a] - message[b2revb], models/mpn.py:91-92), the
// concat of two sources (models/mpn.py:103, 208, 217), ReLU/dropout backward masks, bias,
// residual add, ReLU and dropout (models/mpn.py:94-97).
//
// Geometry (forward): workgroup = 4 waves = 64 rows of A x up to 304 output columns
// (19 MFMA tiles of 16), so the streamed / gathered operand A leaves HBM exactly once and
// the small weight matrix is served from L2.  K is walked in 16-wide tiles, double
// buffered in LDS (47 KB -> 3 workgroups per CU), one barrier per tile.  The MFMA is fed
// with W as its "A" operand and the activations as "B", so a lane's 4 accumulator
// registers are 4 consecutive output columns of one row: the epilogue is 16-byte loads and
// stores.  LDS rows are 64 B with the four 16-byte chunks XOR-swizzled so that
// ds_read_b128 fragment reads are bank-conflict free.
//
// The f32 MFMA is bit-for-bit an fmaf chain; the split path drops no operand bit (its omitted cross terms lie below
// 2^-26 |x w| per product) and measures at or below the f32 chain's error against f64 (tests/test_gpu_split.py,
// tests/test_gpu_headline_kernels.py).  Results differ from the CPU reference by summation order.
#include "rr_common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int BM = 64;      // rows per workgroup
constexpr int BK = 16;      // k-tile
constexpr int THREADS = 256;

enum : int {
  F_A1_VEC = 1, F_A2_VEC = 2, F_SUB_VEC = 4, F_MASK_VEC = 8, F_W1_VEC = 16, F_W2_VEC = 32, F_EPI_VEC = 64, F_PRE_VEC = 128
};

__host__ __device__ constexpr int r16(int k) { return (k + 15) & ~15; }

struct LinearParams {
  rr_linear_args a;
  int w_k1_off;         // column of W where segment 2 starts (k1, or r16(k1) for packed weights)
  int t1, t2;           // k-tiles of segment 1 / 2
  int flags;
  uint32_t drop_thr;
  float keep_scale;
  int persist;          // linear_split_kernel: one workgroup per CU walks row blocks blockIdx.x, + gridDim.x, ... (see there)
};

__device__ __forceinline__ int swz(int row, int kq) { return kq ^ ((0 - (row >> 2)) & 3); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Loads whose address is known to be GLOBAL memory.  A pointer that went through a select with a __device__
// constant (the zero / ones chunks) is a generic pointer to hipcc, which then emits flat_load (counted on lgkmcnt
// as well as vmcnt); these keep the weight-gradient loader on global_load (42 flat_load -> 0 in its ISA).
typedef const __attribute__((address_space(1))) f32x4* rr_gptr4;
typedef const __attribute__((address_space(1))) int32_t* rr_gptri;
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(rr_gptr4)(p); }
__device__ __forceinline__ int32_t ldgi(const int32_t* p) { return *(rr_gptri)(p); }
typedef const __attribute__((address_space(1))) uint8_t* rr_gptrb;
__device__ __forceinline__ uint32_t ldgb(const uint8_t* p) { return *(rr_gptrb)(p); }
// bytes per row of a packed sign mask over N columns: 40 per block of up to 304 columns (2 halves x 20: 19 tile bytes + pad)
__host__ __device__ constexpr int64_t mask_bits_row(int N) { return 40 * ((N + 303) / 304); }

// LDS-DMA: 16 bytes per lane, global -> LDS at (wave-uniform byte address lds_dst) + lane * 16, no VGPR
// destination.  Written as asm so that hipcc does not track it: its own bookkeeping treats an LDS-DMA in
// flight as a may-alias LDS write and drains it (vmcnt(0)) in front of the next ds_read as soon as the
// kernel has a second __shared__ object, which serialises the panel fetch with the MFMA block.  The caller
// waits for it explicitly (rr_wait_vm0) before the barrier that publishes the buffer.
__device__ __forceinline__ void rr_glds16(const float* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void rr_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t rr_lds_addr(const float* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) float*)p));
}

// 4 consecutive floats p[k..k+3] of a row with `ks` valid columns; columns >= ks read as 0.
__device__ __forceinline__ f32x4 load_chunk(const float* p, int k, int ks, bool vec) {
  f32x4 v = f32x4(0.f);
  if (p == nullptr || k >= ks) return v;
  if (vec) {
    v = ld4(p + k);
    if (k + 3 >= ks) {
      if (k + 1 >= ks) v.y = 0.f;
      if (k + 2 >= ks) v.z = 0.f;
      v.w = 0.f;
    }
  } else {
    v.x = p[k];
    if (k + 1 < ks) v.y = p[k + 1];
    if (k + 2 < ks) v.z = p[k + 2];
    if (k + 3 < ks) v.w = p[k + 3];
  }
  return v;
}

__device__ __forceinline__ f32x4 apply_mask(f32x4 v, f32x4 mk, float scale) {
  v.x = mk.x > 0.f ? v.x * scale : 0.f;
  v.y = mk.y > 0.f ? v.y * scale : 0.f;
  v.z = mk.z > 0.f ? v.z * scale : 0.f;
  v.w = mk.w > 0.f ? v.w * scale : 0.f;
  return v;
}

// MODE: 0 plain / concat, 1 = A1 minus a second (optionally gathered) source, 2 = ReLU-backward mask on A
template <int NT, int MODE>
__global__ void __launch_bounds__(THREADS) linear_kernel(const LinearParams P) {
  constexpr int BN = 16 * NT;
  constexpr int B_ITERS = (BN * 4 + THREADS - 1) / THREADS;     // float4 chunks of the W tile per thread
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * BK];

  const rr_linear_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BM;
  const int n0 = blockIdx.y * BN;
  const int flags = P.flags;

  // ---- this thread's staging assignment: one 16-byte chunk of A, B_ITERS chunks of W per k-tile
  const int srow = tid >> 2, skq = tid & 3;
  const int64_t sm = m0 + srow;
  const float* rowp1 = nullptr;
  const float* rowp2 = nullptr;
  const float* subp = nullptr;
  const float* maskp = nullptr;
  if (sm < a.M) {
    if (a.k1 > 0) {
      if (a.a1_idx) {
        const int32_t j = a.a1_idx[sm];
        if (j >= 0) rowp1 = a.a1 + static_cast<int64_t>(j) * a.lda1;
      } else {
        rowp1 = a.a1 + sm * a.lda1;
      }
      if (MODE == 1 && a.a1_sub) {
        if (a.a1_sub_idx) {
          const int32_t j = a.a1_sub_idx[sm];
          if (j >= 0) subp = a.a1_sub + static_cast<int64_t>(j) * a.lda1_sub;
        } else {
          subp = a.a1_sub + sm * a.lda1_sub;
        }
      }
    }
    if (a.k2 > 0) rowp2 = a.a2 + sm * a.lda2;
    if (MODE == 2) maskp = a.a_mask + sm * a.ld_mask;
  }
  const int ktot = a.k1 + a.k2;
  (void)ktot;

  auto load_a = [&](int kt) -> f32x4 {
    f32x4 v;
    if (kt < P.t1) {
      const int k = kt * BK + skq * 4;
      v = load_chunk(rowp1, k, a.k1, flags & F_A1_VEC);
      if (MODE == 1) v = v - load_chunk(subp, k, a.k1, flags & F_SUB_VEC);
      if (MODE == 2) v = apply_mask(v, load_chunk(maskp, k, a.k1, flags & F_MASK_VEC), a.mask_scale);
    } else {
      const int k = (kt - P.t1) * BK + skq * 4;
      v = load_chunk(rowp2, k, a.k2, flags & F_A2_VEC);
    }
    return v;
  };
  auto load_w = [&](int kt, int it) -> f32x4 {
    const int f = it * THREADS + tid;            // chunk id inside the W tile
    const int wr = f >> 2;                       // tile row (output column), kq == skq
    const int n = n0 + wr;
    if (wr >= BN || n >= a.N) return f32x4(0.f);
    const float* wp = a.w + static_cast<int64_t>(n) * a.ldw;
    if (kt < P.t1) return load_chunk(wp, kt * BK + skq * 4, a.k1, flags & F_W1_VEC);
    return load_chunk(wp + P.w_k1_off, (kt - P.t1) * BK + skq * 4, a.k2, flags & F_W2_VEC);
  };
  auto store_tile = [&](int buf, const f32x4& ra, const f32x4 (&rb)[B_ITERS]) {
    float* As = lds[buf];
    float* Bs = lds[buf] + BM * BK;
    *reinterpret_cast<f32x4*>(As + srow * BK + 4 * swz(srow, skq)) = ra;
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int wr = (it * THREADS + tid) >> 2;
      if (wr < BN) *reinterpret_cast<f32x4*>(Bs + wr * BK + 4 * swz(wr, skq)) = rb[it];
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4(0.f);

  const int nk = P.t1 + P.t2;
  f32x4 ra;
  f32x4 rb[B_ITERS];
  ra = load_a(0);
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) rb[it] = load_w(0, it);
  store_tile(0, ra, rb);
  __syncthreads();

  const int fr = lane & 15, fkq = lane >> 4;
  const int a_off = (wave * 16 + fr) * BK + 4 * swz(fr, fkq);
  const int b_off = fr * BK + 4 * swz(fr, fkq);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {                                   // next tile's global loads fly during the MFMAs
      ra = load_a(kt + 1);
#pragma unroll
      for (int it = 0; it < B_ITERS; ++it) rb[it] = load_w(kt + 1, it);
    }
    const float* As = lds[cur];
    const float* Bs = lds[cur] + BM * BK;
    const f32x4 af = ld4(As + a_off);
    constexpr int G = 5;                          // column tiles per group: MFMA dependency distance >= 4
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += G) {
      f32x4 wf[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (t0 + g < NT) wf[g] = ld4(Bs + (t0 + g) * 16 * BK + b_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (t0 + g < NT) acc[t0 + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][j], af[j], acc[t0 + g], 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1, ra, rb);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n .. n+3] per tile: m = row (lane&15), n = tile*16 + (lane>>4)*4
  const int64_t m = m0 + wave * 16 + fr;
  if (m >= a.M) return;
  const int nq = fkq * 4;
  float* crow = a.c + m * a.ldc;
  const float* rrow = nullptr;
  if (a.residual) {
    const int64_t rr = a.residual_idx ? static_cast<int64_t>(a.residual_idx[m]) : m;
    if (rr >= 0) rrow = a.residual + rr * a.ldr;
  }
  const bool evec = flags & F_EPI_VEC;
#pragma unroll
  for (int tc = 0; tc < NT; ++tc) {
    const int n = n0 + tc * 16 + nq;
    if (n >= a.N) continue;
    f32x4 v = acc[tc];
    if (evec) {
      if (a.bias) v = v + ld4(a.bias + n);
      if (rrow) v = v + ld4(rrow + n);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < a.N) {
          if (a.bias) v[e] += a.bias[n + e];
          if (rrow) v[e] += rrow[n + e];
        }
      }
    }
    if (a.c_pre) {
      float* prow = a.c_pre + m * a.ld_pre;
      if (evec && (flags & F_PRE_VEC)) {
        *reinterpret_cast<f32x4*>(prow + n) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) prow[n + e] = v[e];
      }
    }
    if (a.act == RR_ACT_RELU) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    if (P.drop_thr != 0u) {
      const uint64_t base = static_cast<uint64_t>(m) * static_cast<uint64_t>(a.N) + static_cast<uint64_t>(n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rr_keep(a.drop_seed, base + e, P.drop_thr) ? v[e] * P.keep_scale : 0.f;
    }
    if (evec) {
      *reinterpret_cast<f32x4*>(crow + n) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < a.N) crow[n + e] = v[e];
    }
  }
}


#ifdef RR_TRACE
__device__ unsigned long long* rr_trace_buf = nullptr;
#define RR_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (rr_trace_buf && threadIdx.x == 0 && blockIdx.y == 0) {                                           \
      rr_trace_buf[static_cast<size_t>(blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();     \
      if ((slot) == 1) rr_trace_buf[static_cast<size_t>(blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_memtime(); \
      if ((slot) == 2) rr_trace_buf[static_cast<size_t>(blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_memtime(); \
    }                                                                                                    \
  } while (0)
#else
#define RR_STAMP(slot)
#endif

__device__ __attribute__((aligned(16))) const float rr_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};
// a whole row of zeros (4 KiB): "no row" for loaders that walk a row with a wave-uniform column offset
constexpr int RR_ZERO_ROW = 1024;
__device__ __attribute__((aligned(16))) const float rr_zero_row[RR_ZERO_ROW] = {0.f};

// ------------------------------------------------------------------------ fast path
// Same math as linear_kernel, for the hot case: every A source 16-byte addressable and W in
// the packed layout of rr_pack_weight_f32 ([N][r16(k1) + r16(k2)], zero padded).  The loader
// is straight-line code: the W panel of a k-tile goes global -> LDS by LDS-DMA (global_load_lds,
// no VGPR round trip, no ds_write; row index clamped, pad columns are zeros, the XOR swizzle of
// the LDS image is applied to the per-lane source address), A chunks are loaded
// unconditionally from (valid ? row + k : dummy), and every
// fix-up (tail columns, invalid rows, the subtraction / ReLU mask of MODE 1 / 2) happens when
// the registers are written to LDS, i.e. AFTER the k-tile's MFMAs.  Nothing uses a loaded
// value before that point, so the compiler keeps all loads of tile t+1 in flight across the
// whole MFMA block of tile t (the generic kernel waits on each load as it is issued).
template <int NT, int MODE>
__global__ void __launch_bounds__(THREADS, 3) linear_fast_kernel(const LinearParams P) {
  constexpr int BN = 16 * NT;
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * BK];
  __shared__ __attribute__((aligned(16))) float bias_s[BN];

  const rr_linear_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BM;
  const int n0 = blockIdx.y * BN;
  RR_STAMP(0);
#ifdef RR_TRACE
  if (rr_trace_buf && threadIdx.x == 0 && blockIdx.y == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rr_trace_buf[static_cast<size_t>(blockIdx.x) * 8 + 4] = (static_cast<unsigned long long>(xcc) << 32) | hw;
  }
#endif

  const int srow = tid >> 2, skq = tid & 3;
  const int64_t sm = m0 + srow;
  const float* const dummy = a.w;                     // any valid, 16-byte aligned address
  const float* rowp1 = nullptr;
  const float* rowp2 = nullptr;
  const float* subp = nullptr;                        // MODE 1: subtract source, MODE 2: mask source
  {
    // both gather indices are loaded before either is used (one memory round trip, not two): the loads are
    // unconditional from a selected address, the row pointers are derived afterwards
    const bool in_m = sm < a.M;
    const int64_t smc = in_m ? sm : 0;
    const bool g1 = a.k1 > 0 && a.a1_idx != nullptr;
    const bool g2 = MODE == 1 && a.k1 > 0 && a.a1_sub != nullptr && a.a1_sub_idx != nullptr;
    const int32_t* const izero = reinterpret_cast<const int32_t*>(rr_zero_chunk);
    const int32_t j1 = *(g1 ? a.a1_idx + smc : izero);
    const int32_t j2 = *(g2 ? a.a1_sub_idx + smc : izero);
    if (in_m) {
      if (a.k1 > 0) {
        if (g1) {
          if (j1 >= 0) rowp1 = a.a1 + static_cast<int64_t>(j1) * a.lda1;
        } else {
          rowp1 = a.a1 + sm * a.lda1;
        }
        if (MODE == 1 && a.a1_sub) {
          if (g2) {
            if (j2 >= 0) subp = a.a1_sub + static_cast<int64_t>(j2) * a.lda1_sub;
          } else {
            subp = a.a1_sub + sm * a.lda1_sub;
          }
        }
        if (MODE == 2) subp = a.a_mask + sm * a.ld_mask;
      }
      if (a.k2 > 0) rowp2 = a.a2 + sm * a.lda2;
    }
  }
  float* dzrow = nullptr;                             // MODE 2 side output: dz_out (+)= masked operand
  if (MODE == 2 && a.dz_out && sm < a.M && blockIdx.y == 0) dzrow = a.dz_out + sm * a.ld_dz;
  // W panel by LDS-DMA: wave-instruction j fills column tile j (16 rows x 4 chunks = 1 KiB, lane-linear
  // in LDS); the XOR swizzle of the LDS image goes on each lane's SOURCE column chunk.
  constexpr int G_ITERS = (NT + 3) / 4;
  const int uwave = __builtin_amdgcn_readfirstlane(wave) & 3;
  const float* wsrc[G_ITERS];
#pragma unroll
  for (int it = 0; it < G_ITERS; ++it) {
    const int wr = (it * 4 + uwave) * 16 + (lane >> 2);
    int n = n0 + wr;
    if (n > a.N - 1) n = a.N - 1;
    wsrc[it] = a.w + static_cast<int64_t>(n) * a.ldw + 4 * swz(wr, lane & 3);
  }
  const int k1p = r16(a.k1);

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4(0.f);

  const int nk = P.t1 + P.t2;
  f32x4 ra, rs, rc;

  auto issue = [&](int kt) {                          // pure loads, no arithmetic on the results
    const bool seg1 = kt < P.t1;
    const int kl = (seg1 ? kt : kt - P.t1) * BK + skq * 4;
    const float* p = seg1 ? rowp1 : rowp2;
    const int ks = seg1 ? a.k1 : a.k2;
    ra = ld4((p != nullptr && kl < ks) ? p + kl : dummy);
    if (MODE != 0) rs = ld4((seg1 && subp != nullptr && kl < ks) ? subp + kl : dummy);
    if (MODE == 2) {
      if (a.dz_accumulate) rc = ld4((dzrow != nullptr && kl < ks) ? dzrow + kl : dummy);
    }
    const int kw = seg1 ? kt * BK : k1p + (kt - P.t1) * BK;
    float* Bd = lds[kt & 1] + BM * BK;
#pragma unroll
    for (int it = 0; it < G_ITERS; ++it) {
      const int j = it * 4 + uwave;
      if (j < NT) rr_glds16(wsrc[it] + kw, rr_lds_addr(Bd + j * 16 * BK));
    }
  };
  auto commit = [&](int kt, int buf) {                // fix-ups + LDS stores (first use of the loads)
    const bool seg1 = kt < P.t1;
    const int kl = (seg1 ? kt : kt - P.t1) * BK + skq * 4;
    const float* p = seg1 ? rowp1 : rowp2;
    const int ks = seg1 ? a.k1 : a.k2;
    const bool ok = (p != nullptr);
    f32x4 v;
    v.x = (ok && kl + 0 < ks) ? ra.x : 0.f;
    v.y = (ok && kl + 1 < ks) ? ra.y : 0.f;
    v.z = (ok && kl + 2 < ks) ? ra.z : 0.f;
    v.w = (ok && kl + 3 < ks) ? ra.w : 0.f;
    if (MODE == 1) {
      const bool oks = seg1 && (subp != nullptr);
      v.x -= (oks && kl + 0 < ks) ? rs.x : 0.f;
      v.y -= (oks && kl + 1 < ks) ? rs.y : 0.f;
      v.z -= (oks && kl + 2 < ks) ? rs.z : 0.f;
      v.w -= (oks && kl + 3 < ks) ? rs.w : 0.f;
    }
    if (MODE == 2) {
      const bool oks = seg1 && (subp != nullptr);
      v.x = (oks && kl + 0 < ks && rs.x > 0.f) ? v.x * a.mask_scale : 0.f;
      v.y = (oks && kl + 1 < ks && rs.y > 0.f) ? v.y * a.mask_scale : 0.f;
      v.z = (oks && kl + 2 < ks && rs.z > 0.f) ? v.z * a.mask_scale : 0.f;
      v.w = (oks && kl + 3 < ks && rs.w > 0.f) ? v.w * a.mask_scale : 0.f;
      if (dzrow != nullptr && kl < ks)                  // k1 % 4 == 0: the chunk is whole
        *reinterpret_cast<f32x4*>(dzrow + kl) = a.dz_accumulate ? v + rc : v;
    }
    float* As = lds[buf];
    *reinterpret_cast<f32x4*>(As + srow * BK + 4 * swz(srow, skq)) = v;
  };

  if (tid < BN / 4) {                                  // bias slice of this column block -> LDS (read in the epilogue)
    const int n = n0 + tid * 4;
    *reinterpret_cast<f32x4*>(bias_s + tid * 4) = ld4((a.bias && n < a.N) ? a.bias + n : dummy);
  }
  issue(0);
  commit(0, 0);
  rr_wait_vm0();                                       // the LDS-DMA of the first W panel
  __syncthreads();
  RR_STAMP(1);

  const int fr = lane & 15, fkq = lane >> 4;
  const int a_off = (wave * 16 + fr) * BK + 4 * swz(fr, fkq);
  const int b_off = fr * BK + 4 * swz(fr, fkq);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) issue(kt + 1);
    const float* As = lds[cur];
    const float* Bs = lds[cur] + BM * BK;
    const f32x4 af = ld4(As + a_off);
    constexpr int G = 5;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += G) {
      f32x4 wf[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (t0 + g < NT) wf[g] = ld4(Bs + (t0 + g) * 16 * BK + b_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (t0 + g < NT) acc[t0 + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][j], af[j], acc[t0 + g], 0, 0, 0);
      }
    }
    if (more) commit(kt + 1, cur ^ 1);
    rr_wait_vm0();                                     // W panel of tile kt+1 has landed in LDS
    __syncthreads();
  }

  // ---- epilogue (vector form only: the fast path requires F_EPI_VEC).
  // vmcnt retires loads and stores in issue order, so a wait for a load that was issued after a
  // store also waits for that store's write acknowledge.  The tile loop is therefore kept free of
  // path-dependent memory operations (unconditional loads from a safe address, selects instead of
  // branches, bias from LDS) and the residual chunks run D tiles ahead in a register ring: a store
  // is only ever waited for D tiles after it was issued, instead of one write round trip per tile.
  RR_STAMP(2);
  const int64_t m = m0 + wave * 16 + fr;
  const bool row_ok = m < a.M;
  const int64_t mc = row_ok ? m : a.M - 1;
  const int nq = fkq * 4;
  float* crow = a.c + mc * a.ldc;
  const float* rrow = nullptr;
  if (a.residual) {
    const int64_t rr = a.residual_idx ? static_cast<int64_t>(a.residual_idx[mc]) : mc;
    if (rr >= 0) rrow = a.residual + rr * a.ldr;
  }
  const bool has_bias = a.bias != nullptr;
  const bool relu = a.act == RR_ACT_RELU;
  auto finish = [&](f32x4 v, int n) -> f32x4 {         // activation + dropout + store of one chunk; returns what was stored
    if (relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    if (P.drop_thr != 0u) {
      // N % 4 == 0 and n % 4 == 0 on this path: the 4 elements are one aligned group of the mask stream
      const uint64_t base = static_cast<uint64_t>(m) * static_cast<uint64_t>(a.N) + static_cast<uint64_t>(n);
      const uint32_t w = rr_hash_group(a.drop_seed, base >> 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rr_hash_lane(w, e) >= P.drop_thr ? v[e] * P.keep_scale : 0.f;
    }
    if (row_ok && n < a.N) *reinterpret_cast<f32x4*>(crow + n) = v;
    return v;
  };
  float* prow = a.c_pre ? a.c_pre + mc * a.ld_pre : nullptr;   // pre-activation side output (W_i layers)
  // Optional third output: partial[blockIdx.x, n] = sum over this workgroup's 64 rows of w[m] * C[m, n] (the padding
  // row's adjoint, see rr_linear_args.colsum_partial).  A lane holds 4 columns of ONE row per tile, the 16 rows of a
  // wave sit in the 16 lanes of a DPP row: four row_shr adds leave the 16-row sum in lane 15 of every row, which
  // parks it in the (now idle) k-loop LDS; after the tile loop 76 threads add the four waves' slices in wave order.
  const bool cs_on = a.colsum_partial != nullptr;
  const float wrow = (cs_on && row_ok) ? a.colsum_w[mc] : 0.f;
  float* const cs_lds = &lds[0][0];                    // [4 waves][BN]
  {
    const bool res_ok = rrow != nullptr;
    const float* rbase = res_ok ? rrow : dummy;
    constexpr int D = 4;
    f32x4 ring[D + 1];
    auto ldres = [&](int tc) {
      const int n = n0 + tc * 16 + nq;
      return ld4(rbase + (n < a.N ? n : 0));
    };
#pragma unroll
    for (int t = 0; t < D; ++t)
      if (t < NT) ring[t] = ldres(t);
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
      const int n = n0 + tc * 16 + nq;
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + tc * 16 + nq);
      f32x4 v = acc[tc];
      const f32x4 vb = v + b;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = has_bias ? vb[e] : v[e];
      const f32x4 vr = v + ring[tc % (D + 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = res_ok ? vr[e] : v[e];
      if (tc + D < NT) ring[(tc + D) % (D + 1)] = ldres(tc + D);
      if (prow != nullptr && row_ok && n < a.N) *reinterpret_cast<f32x4*>(prow + n) = v;
      const f32x4 stored = finish(v, n);
      if (cs_on) {
        f32x4 t = stored * wrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = t[e];
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));   // row_shr:1
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));   // row_shr:2
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));   // row_shr:4
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));   // row_shr:8
          t[e] = x;
        }
        if (fr == 15) *reinterpret_cast<f32x4*>(cs_lds + wave * BN + tc * 16 + nq) = t;
      }
    }
  }
  if (cs_on) {
    __syncthreads();
    if (tid < BN / 4) {
      const int n = n0 + tid * 4;
      const f32x4 s01 = ld4(cs_lds + tid * 4) + ld4(cs_lds + BN + tid * 4);
      const f32x4 s23 = ld4(cs_lds + 2 * BN + tid * 4) + ld4(cs_lds + 3 * BN + tid * 4);
      if (n < a.N) *reinterpret_cast<f32x4*>(a.colsum_partial + static_cast<int64_t>(blockIdx.x) * a.ld_partial + n) = s01 + s23;
    }
  }
#ifdef RR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RR_STAMP(3);
#endif
}

// dst = zero-padded packed copy of a weight (or of its transpose) for the fast path
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ src, int64_t ld_src, int transpose,
                                                          int rows, int c0, int k1, int k2, float* __restrict__ dst) {
  const int k1p = r16(k1), ldd = r16(k1) + r16(k2);
  const int64_t total = static_cast<int64_t>(rows) * ldd;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int r = static_cast<int>(e / ldd), c = static_cast<int>(e - static_cast<int64_t>(r) * ldd);
    int lc = -1;                                      // logical column
    if (c < k1) lc = c;
    else if (c >= k1p && c - k1p < k2) lc = k1 + (c - k1p);
    float v = 0.f;
    if (lc >= 0) v = transpose ? src[static_cast<int64_t>(lc) * ld_src + c0 + r] : src[static_cast<int64_t>(r) * ld_src + c0 + lc];
    dst[e] = v;
  }
}

// the same for up to RR_MAX_PACK weights in ONE launch (blockIdx.y = weight): a training step re-packs ~10 weights
// for its forward and ~10 transposes for its backward, each a 2-5 us kernel with a launch boundary around it
constexpr int RR_MAX_PACK_DEV = RR_MAX_PACK;
struct PackMany {
  rr_pack_desc d[RR_MAX_PACK_DEV];
};
__device__ __forceinline__ void pack_plain_desc(const rr_pack_desc& q) {
  const int k1p = r16(q.k1), ldd = r16(q.k1) + r16(q.k2);
  const int64_t total = static_cast<int64_t>(q.rows) * ldd;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int r = static_cast<int>(e / ldd), c = static_cast<int>(e - static_cast<int64_t>(r) * ldd);
    int lc = -1;
    if (c < q.k1) lc = c;
    else if (c >= k1p && c - k1p < q.k2) lc = q.k1 + (c - k1p);
    float v = 0.f;
    if (lc >= 0)
      v = q.transpose ? q.src[static_cast<int64_t>(lc) * q.ld_src + q.c0 + r] : q.src[static_cast<int64_t>(r) * q.ld_src + q.c0 + lc];
    q.dst[e] = v;
  }
}
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackMany P) {
  const rr_pack_desc& q = P.d[blockIdx.y];
  if (q.split) return;                                 // pack_split_kernel's
  pack_plain_desc(q);
}

// ------------------------------------------------------------------------ split path (3 x bf16 terms, 6 products)
// The same GEMM on the bf16 matrix core without giving up f32 accuracy.  Every f32 operand is written EXACTLY as
// the sum of three bf16 terms, x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1: the two
// remainders are exact in f32 and the last one has at most 8 significant bits left), and
//     x * w  =  x0 w0 + (x0 w1 + x1 w0) + (x0 w2 + x1 w1 + x2 w0)  +  terms below 2^-24 |x w|
// is accumulated in f32 by six v_mfma_f32_16x16x32_bf16 per 32-deep k-step, smallest terms first.  bf16 x bf16
// products are exact in f32, and the sum of a k-step is rounded once instead of after every fmaf, so the error
// against an f64 GEMM is at or BELOW that of the f32 MFMA chain (tests/test_gpu_split.py measures both).  Six bf16
// MFMAs cost 6/16 of the f32 MFMA's cycles for the same k: the kernel moves from MFMA-bound to HBM-bound.
//
// Geometry: workgroup = 12 waves x 16 rows = 192 rows x up to 304 output columns; the activation operand goes
// global -> registers (each lane loads the 8 consecutive k of ITS row that the MFMA layout hands it, fixes them up
// - gather / subtract / ReLU mask - and splits them: every element is converted exactly once); the weight terms
// are pre-split by rr_pack_weights (w_packed = 2) into the exact LDS image of a k-step (per column tile and term:
// 64 lanes x 16 B, lane-linear) and stream L2 -> LDS by LDS-DMA, double buffered (2 x 57 KB: one workgroup per CU).
// N > 304 (H = 600): two column blocks of 19 tiles (blockIdx.y), each streaming its own tiles of the image.
#ifndef RR_EPI_MODE
#define RR_EPI_MODE 1
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SK = 32;                    // k per step

__host__ __device__ constexpr int r32(int k) { return (k + 31) & ~31; }

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// exact three-term split of two floats (packed bf16 pairs, low half = x)
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = cvt_pk_bf16(x, y);
  float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
  p1 = cvt_pk_bf16(rx, ry);
  rx -= __uint_as_float(p1 << 16);
  ry -= __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt_pk_bf16(rx, ry);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// Two-term f16 form (w_packed = 3): S x = h + l with h = f16(S x), l = f16(S x - h) (round to nearest even; the remainder
// is exact in f32), 22 significant bits of every operand, and  x w = h_x h_w + (h_x l_w + l_x h_w) + terms below
// 2^-22 |x w|: three v_mfma_f32_16x16x32_f16 per k-step instead of six bf16 ones.  S is a power of two that puts the
// tensor's largest magnitude below 2^15 (f16 has 5 exponent bits: the caller supplies the bound).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// the power of two S with 2^14 <= S * bound < 2^15 (bounds outside 2^+-110, zero included, are clamped: nothing to protect
// below, garbage in above), and its inverse
__host__ __device__ __forceinline__ int rr_f16_exp(float bound) {
  uint32_t u;
  __builtin_memcpy(&u, &bound, 4);
  int e = static_cast<int>((u >> 23) & 0xffu) - 127;     // bound < 2^(e+1)
  return e < -110 ? -110 : (e > 110 ? 110 : e);
}
// |v| folded into a running maximum; a wave's maximum into a device float (one atomic per wave, and only while it can
// still raise the slot: a stale read costs an atomic, never a result)
__device__ __forceinline__ float rr_amax4(float m, f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// A wave's maximum into a word of LDS (the workgroup's running maximum; the device float gets ONE atomic per workgroup at
// the end of the kernel).  Nothing here touches global memory: a load of the slot at this point would be waited for with
// vmcnt behind every store the epilogue has just issued - loads and stores retire in issue order - and drain the store
// queue at each row block (measured: the GEMM twice as slow); an atomic per wave and block is 4,464 atomics on one address.
__device__ __forceinline__ void rr_amax_commit_wave(float m, unsigned int* lds_word) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(lds_word, __float_as_uint(m));
}
__device__ __forceinline__ float rr_pow2(int e) { return __uint_as_float(static_cast<uint32_t>(127 + e) << 23); }
__device__ __forceinline__ void split_pair_h(float x, float y, float S, uint32_t& p0, uint32_t& p1) {
  const f32x2 v = {x * S, y * S};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(uint32_t, h);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

// NTP: column tiles of the packed weight image; NT: column tiles of ONE workgroup (blockIdx.y picks tiles y * NT ...);
// WAVES: 16-row groups per workgroup.  Instantiated: <19, 19, 12 waves> - one workgroup per CU (114 KB of LDS) covers all
// columns of 192 rows - and <10, 10, 8> / <4, 4, 8> for narrow layers.  Cutting the 19 tiles into 10 + 9 (<19, 10, 8>:
// 60 KB, <= 128 registers, TWO workgroups per CU whose store epilogues and MFMA loops overlap) was measured and lost:
// both halves load and split the operand rows, 320 vs 236 us on the masked dX GEMM (profiles/r02_experiments.txt).
// EPI: 0 = accumulator-layout epilogue, 1 = row-contiguous epilogue through LDS (12-wave geometry; chosen per launch,
// see launch_split_one: separate instantiations keep each epilogue's registers out of the other's kernel)
template <int NTP, int NT, int MODE, int WAVES, int EPI = 0, bool F16 = false>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 4 : 3) linear_split_kernel(const LinearParams P) {
  constexpr int BN = 16 * NT;
  constexpr int TERMS = F16 ? 2 : 3;                   // operand terms: three bf16 or two f16
  constexpr int PANEL = NT * TERMS * 1024;             // bytes of one k-step's weight image in LDS
  constexpr int SRC_PANEL = NTP * TERMS * 1024;            // ... and in the packed weights
  constexpr int S_THREADS = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS layout.  k-loop: two weight images [0, 2 * PANEL).  Epilogue of the 12-wave geometry (RS_EPI): the images' space
  // becomes twelve wave-private transposition regions of 8 rows x 77 float4 (the 77th is padding: rows 1232 bytes apart
  // keep the eight lanes of a ds_write_b128 group on different banks), followed by the column-sum / sign-bit staging and
  // the bias slice.  The 8-wave geometries keep bias and staging where they were.
  constexpr bool RS_EPI = WAVES == 12;
  constexpr int RS = 77, REGION = 8 * RS * 16;         // bytes per wave and pass
  constexpr int CS_OFF = RS_EPI ? (WAVES * REGION > 2 * PANEL ? WAVES * REGION : 2 * PANEL) : 0;
  constexpr int BIAS_OFF = RS_EPI ? CS_OFF + WAVES * BN * 4 : 2 * PANEL;
  float* const bias_s = reinterpret_cast<float*>(smem + BIAS_OFF);
  constexpr int PF_OFF = BIAS_OFF + BN * 4;            // persistent form: 2 KiB per wave for the next block's step-0 operand chunks
  // Two column blocks (N > 304: <38, 19, ...>): a 1-D grid in which ids i and i + 8 are the two column blocks of ONE row block.
  // Workgroup ids are dealt to the 8 XCDs round-robin, so the pair lands on one XCD within a few dispatches of each other and the
  // second one finds the operand rows in that XCD's L2 (a (rows, 2) grid ran all first column blocks before any second one: every
  // operand row left HBM twice).
  constexpr bool PAIR = NTP == 2 * NT;
  const unsigned int bx = PAIR ? (((blockIdx.x >> 4) << 3) + (blockIdx.x & 7u)) : blockIdx.x;
  const unsigned int by = PAIR ? ((blockIdx.x >> 3) & 1u) : blockIdx.y;
  if (PAIR && static_cast<int64_t>(bx) * (16 * WAVES) >= P.a.M) return;   // (uniform: the grid is padded to whole groups of 16 ids)
  const int t0 = by * NT;                              // first column tile of this workgroup
  const int nth = NTP - t0 < NT ? NTP - t0 : NT;       // its column tiles (the last workgroup of a row block may have fewer)
  const bool full = nth == NT;
  const int n0 = t0 * 16;

  const rr_linear_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fkq = lane >> 4;
  // two-term f16 form: the operand scale from the caller's bounds (uniform: scalar loads), the weight's from its image
  float xs = 1.f, ixs = 1.f, iws = 1.f;
  if (F16) {
    float b1 = (a.a1_amax ? rr_amax_read(a.a1_amax) : 0.f) + (a.a1_sub_amax ? rr_amax_read(a.a1_sub_amax) : 0.f);
    if (MODE == 2 || MODE == 3) b1 *= fabsf(a.mask_scale);
    const float b2 = a.a2_amax ? rr_amax_read(a.a2_amax) : 0.f;
    const float bound = fmaxf(b1, b2);
    const int e = rr_f16_exp(bound);
    xs = bound < 2.5e33f ? rr_pow2(14 - e) : __builtin_nanf("");   // an infinite / > 2^110 element: no scale fits, every output is NaN
    ixs = rr_pow2(e - 14);
    iws = 1.0f / a.w[static_cast<int64_t>(P.t1 + P.t2) * (SRC_PANEL / 4)];   // (a power of two: exact)
  }
  RR_STAMP(0);
#ifdef RR_TRACE
  if (rr_trace_buf && threadIdx.x == 0 && by == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rr_trace_buf[static_cast<size_t>(bx) * 8 + 4] = (static_cast<unsigned long long>(xcc) << 32) | hw;
  }
#endif
#ifdef RR_SPLIT_STAGGER
  // experiment: de-phase the workgroups of the first round (equal work + simultaneous start = every CU in its store
  // epilogue at the same time); later workgroups start when an earlier one retires and inherit the offsets
  if (WAVES == 12 && gridDim.x > 256 && blockIdx.x < 256) {
    const int ph = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < ph * RR_SPLIT_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // Persistent form (plain-operand GEMMs of the 12-wave geometry, chosen by the host: P.persist): gridDim.x = one workgroup
  // per CU, each owning a CONTIGUOUS range of rows - an equal share of the 64-row units, so every CU finishes at the same
  // time - and walking it in blocks of up to 12 waves x 16 rows; the last block of a range has 4 or 8 active waves (one or
  // two per SIMD instead of three: it takes 1/3 or 2/3 of a full block's time, where the one-block-per-workgroup launch
  // pays a whole second round for it).  The k-loop's software pipeline (operand chunks two steps ahead, weight image one
  // step ahead) continues across the block boundary: the last step of a block issues the NEXT block's first weight image
  // (the weights do not depend on the rows), the step before it fetches the next block's step-0 operand chunks, so a
  // block's ~5 us of dependent loads before its first MFMA run under the previous block's last MFMA blocks.  Same values,
  // same order per element: bit-identical to the one-block form.  Rows are counted in groups of 16 (one wave's rows).
  constexpr bool CAN_PERSIST = MODE == 0 && WAVES == 12 && NT == NTP && EPI == 0;
  const bool persist = CAN_PERSIST && P.persist != 0;
  // the workgroup's running maxima of |C| / |dz_out| (rr_linear_args.c_amax_out / dz_amax_out): two words behind everything else
  unsigned int* const amx = reinterpret_cast<unsigned int*>(smem + PF_OFF + (CAN_PERSIST ? WAVES * 2048 : 0));
  if (F16 && tid == 0) { amx[0] = 0u; amx[1] = 0u; }    // (the prologue's barrier orders this before any use)
  int64_t g_cur = static_cast<int64_t>(bx) * WAVES;                    // first 16-row group of the current block
  int64_t g_end = g_cur + WAVES;                                        // end of this workgroup's range
  if (persist) {
    const int64_t units = (a.M + 63) / 64, G = gridDim.x;
    const int64_t base = units / G, rem = units % G, p = blockIdx.x;
    g_cur = 4 * (p * base + (p < rem ? p : rem));
    g_end = g_cur + 4 * (base + (p < rem ? 1 : 0));
  }
  int nw = g_end - g_cur < WAVES ? static_cast<int>(g_end - g_cur) : WAVES;   // active waves of the current block (uniform)
  int64_t m0 = g_cur * 16;
  int64_t m = m0 + wave * 16 + fr;
  bool row_ok = wave < nw && m < a.M;
  int64_t mc = row_ok ? m : a.M - 1;
  const float* const dummy = a.w;                      // any valid, 16-byte aligned GLOBAL address (keeps the loads global_load)
  const float* rowp1 = nullptr;
  const float* rowp2 = nullptr;
  const float* subp = nullptr;                         // MODE 1: subtract source, MODE 2: mask source
  const uint8_t* bitrow = nullptr;                     // MODE 3: the mask as one bit per element (rr_linear_args.a_mask_bits)
  {
    const bool g1 = a.k1 > 0 && a.a1_idx != nullptr;
    const bool g2 = MODE == 1 && a.k1 > 0 && a.a1_sub != nullptr && a.a1_sub_idx != nullptr;
    const int32_t j1 = g1 ? ldgi(a.a1_idx + mc) : 0;
    const int32_t j2 = g2 ? ldgi(a.a1_sub_idx + mc) : 0;
    if (row_ok) {
      if (a.k1 > 0) {
        if (g1) {
          if (j1 >= 0) rowp1 = a.a1 + static_cast<int64_t>(j1) * a.lda1;
        } else {
          rowp1 = a.a1 + m * a.lda1;
        }
        if (MODE == 1 && a.a1_sub) {
          if (g2) {
            if (j2 >= 0) subp = a.a1_sub + static_cast<int64_t>(j2) * a.lda1_sub;
          } else {
            subp = a.a1_sub + m * a.lda1_sub;
          }
        }
        if (MODE == 2) subp = a.a_mask + m * a.ld_mask;
        if (MODE == 3) bitrow = a.a_mask_bits + m * mask_bits_row(a.k1);
      }
      if (a.k2 > 0) rowp2 = a.a2 + m * a.lda2;
    }
  }
  float dz_am = 0.f;                                   // largest |dz_out| this lane stored (rr_linear_args.dz_amax_out)
  float* dzrow = nullptr;                              // MODE 2 side output: dz_out (+)= masked operand
  if ((MODE == 2 || MODE == 3) && a.dz_out && row_ok && by == 0) dzrow = a.dz_out + m * a.ld_dz;

  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const float* const wlane = a.w + t0 * (TERMS * 256) + lane * 4;   // this workgroup's tiles of a step; 16 B per lane inside a 1 KiB block
  const uint32_t lds0 = rr_lds_addr(reinterpret_cast<const float*>(smem));

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4(0.f);

  const int nk = P.t1 + P.t2;
  // operand chunks in flight: two k-steps (slot = step & 1), so a load has two MFMA blocks to land; the weight image one
  f32x4 ra[WAVES == 12 ? 2 : 1][2], rs[WAVES == 12 ? 2 : 1][2];
  uint32_t rb[2] = {0u, 0u};                           // MODE 3: the 8 mask bits of a step's chunk pair
  u32x4 x0, x1, x2;                                    // the three bf16 terms of the current step's operand

  // Interior k-steps (all 32 columns of the step inside the segment) of MODE 0 / 1 take a leaner path: the lane's row
  // pointers are resolved ONCE (a missing row points at a row of zeros), a step adds its wave-uniform column offset, and
  // fixup() needs no per-element selects.  ~27 of the ~90 vector instructions of a k-step; same loaded values, same
  // arithmetic.  The last step of a segment (partial: K = 300 ends inside it) keeps the select form below.
#ifndef RR_SPLIT_NO_FASTX
  constexpr bool FASTX = (MODE == 0 || MODE == 1);
#else
  constexpr bool FASTX = false;
#endif
  // an interior step reads columns [s*SK, (s+1)*SK) <= k of its row - or of rr_zero_row when the row is missing: the
  // segment (plus one step of slack for the prefetch) must fit inside that array
  static_assert(RR_ZERO_ROW % SK == 0 && RR_ZERO_ROW >= 2 * SK, "rr_zero_row must hold whole k-steps");
  // (the EPI 0 / 1 instantiations of the 12-wave geometry are only launched with segments that fit - launch_split_one sends
  // longer ones to their twins EPI 2 / 3, which keep the generic loader - so their select-per-element loader is dead code:
  // fewer live scalars and pointers in kernels that have none to spare.  Measured on
  // the persistent form <19,19,0,12,0>: 12 -> 2 spilled registers, -3 ... -5 % per launch at 71k rows, -0.7 % on the step)
  constexpr bool LEAN_ONLY = FASTX && WAVES == 12 && EPI < 2;
  const bool fastx_ok = FASTX && (LEAN_ONLY || (a.k1 + SK <= RR_ZERO_ROW && a.k2 + SK <= RR_ZERO_ROW));
  const float* xb1 = (rowp1 != nullptr ? rowp1 : rr_zero_row) + fkq * 8;
  const float* xb2 = (rowp2 != nullptr ? rowp2 : rr_zero_row) + fkq * 8;
  const float* const sb1 = (subp != nullptr ? subp : rr_zero_row) + fkq * 8;
  const float* xb1n = xb1;                             // persistent form: the NEXT row block's operand rows
  const float* xb2n = xb2;
  bool has_next = false, wrapped = false;              // wrapped: this block was entered from the previous block's pipeline
  auto next_rows = [&](int64_t g, int nwn) {           // MODE 0 only: plain or index-gathered segment 1, plain segment 2
    const int64_t mm = (g + wave) * 16 + fr;
    const bool ok = wave < nwn && mm < a.M;
    const int64_t mmc = ok ? mm : a.M - 1;
    const bool g1 = a.k1 > 0 && a.a1_idx != nullptr;
    const int32_t j1 = g1 ? ldgi(a.a1_idx + mmc) : 0;
    const float* r1 = nullptr;
    const float* r2 = nullptr;
    if (ok) {
      if (a.k1 > 0) {
        if (g1) {
          if (j1 >= 0) r1 = a.a1 + static_cast<int64_t>(j1) * a.lda1;
        } else {
          r1 = a.a1 + mm * a.lda1;
        }
      }
      if (a.k2 > 0) r2 = a.a2 + mm * a.lda2;
    }
    xb1n = (r1 != nullptr ? r1 : rr_zero_row) + fkq * 8;
    xb2n = (r2 != nullptr ? r2 : rr_zero_row) + fkq * 8;
  };
  auto interior = [&](int s) -> bool {                 // (wave-uniform)
    if (!fastx_ok) return false;
    return s < P.t1 ? (s + 1) * SK <= a.k1 : (s - P.t1 + 1) * SK <= a.k2;
  };
  auto issue_x = [&](int s, int slot, bool nextblk = false) {   // pure loads (unconditional, from a selected address)
    if (FASTX && fastx_ok) {
      const bool s1 = s < P.t1;
      const int off = (s1 ? s : s - P.t1) * SK;          // wave-uniform
      const float* p = (s1 ? (nextblk ? xb1n : xb1) : (nextblk ? xb2n : xb2)) + off;
      const float* q = (MODE == 1 && s1) ? sb1 + off : rr_zero_row;   // (segment 2 has no subtract source: zeros, the count of loads per step stays NX)
      if (interior(s)) {
        ra[slot][0] = ldg4(p);
        ra[slot][1] = ldg4(p + 4);
        if (MODE == 1) {
          rs[slot][0] = ldg4(q);
          rs[slot][1] = ldg4(q + 4);
        }
      } else {                                         // last step of a segment: chunks past its end read zeros
        int fq = fkq;                                  // (opaque: these selects are per-lane loop invariants - left alone they are
        if (CAN_PERSIST) asm volatile("" : "+v"(fq));  // hoisted out of the block loop, kept live across the k-loop and spilled)
        const int kl = off + fq * 8, ks = s1 ? a.k1 : a.k2;
        ra[slot][0] = ldg4(kl < ks ? p : rr_zero_row);
        ra[slot][1] = ldg4(kl + 4 < ks ? p + 4 : rr_zero_row);
        if (MODE == 1) {
          rs[slot][0] = ldg4(kl < ks ? q : rr_zero_row);
          rs[slot][1] = ldg4(kl + 4 < ks ? q + 4 : rr_zero_row);
        }
      }
      return;
    }
    const bool seg1 = s < P.t1;
    const int kl = (seg1 ? s : s - P.t1) * SK + fkq * 8;
    const float* p = seg1 ? rowp1 : rowp2;
    const int ks = seg1 ? a.k1 : a.k2;
    ra[slot][0] = ldg4((p != nullptr && kl < ks) ? p + kl : dummy);
    ra[slot][1] = ldg4((p != nullptr && kl + 4 < ks) ? p + kl + 4 : dummy);
    if (MODE == 1 || MODE == 2) {
      const bool oks = seg1 && subp != nullptr;
      rs[slot][0] = ldg4((oks && kl < ks) ? subp + kl : dummy);
      rs[slot][1] = ldg4((oks && kl + 4 < ks) ? subp + kl + 4 : dummy);
    }
    if (MODE == 3) {                                   // byte (column block, half tile, tile) holds k = kl .. kl+7, bit e <-> kl + e
      const int tc = 2 * s + (fkq >> 1);
      const int y = tc / 19;
      const uint8_t* q = bitrow + y * 40 + (fkq & 1) * 20 + (tc - 19 * y);
      rb[slot] = ldgb((seg1 && bitrow != nullptr && kl < ks) ? q : reinterpret_cast<const uint8_t*>(dummy));
    }
  };
  auto issue_w = [&](int s) {                          // weight image of step s: NT * 3 LDS-DMA blocks of 1 KiB over the waves
    const float* src = wlane + static_cast<int64_t>(s) * (SRC_PANEL / 4);
    const uint32_t dst = lds0 + (s & 1) * PANEL;
#pragma unroll
    for (int b0 = 0; b0 < NT * TERMS; b0 += WAVES) {
      const int b = b0 + uwave;
      if (b < nth * TERMS) rr_glds16(src + b * 256, dst + b * 1024);
    }
  };
  auto split8 = [&](const f32x4& v0, const f32x4& v1) {
    uint32_t t0, t1, t2;
    if (F16) {
      split_pair_h(v0.x, v0.y, xs, t0, t1); x0.x = t0; x1.x = t1;
      split_pair_h(v0.z, v0.w, xs, t0, t1); x0.y = t0; x1.y = t1;
      split_pair_h(v1.x, v1.y, xs, t0, t1); x0.z = t0; x1.z = t1;
      split_pair_h(v1.z, v1.w, xs, t0, t1); x0.w = t0; x1.w = t1;
      return;
    }
    split_pair(v0.x, v0.y, t0, t1, t2); x0.x = t0; x1.x = t1; x2.x = t2;
    split_pair(v0.z, v0.w, t0, t1, t2); x0.y = t0; x1.y = t1; x2.y = t2;
    split_pair(v1.x, v1.y, t0, t1, t2); x0.z = t0; x1.z = t1; x2.z = t2;
    split_pair(v1.z, v1.w, t0, t1, t2); x0.w = t0; x1.w = t1; x2.w = t2;
  };
  auto fixup = [&](int s, int slot) {                  // first use of the loads: selects, mask / subtract, split
    if (CAN_PERSIST && uwave >= nw) return;             // (its loads are still issued: every wave keeps the same vmcnt sequence)
    if (FASTX && fastx_ok) {
      f32x4 v0 = ra[slot][0], v1 = ra[slot][1];
      f32x4 u0 = f32x4(0.f), u1 = f32x4(0.f);
      if (MODE == 1) { u0 = rs[slot][0]; u1 = rs[slot][1]; }
      const bool s1 = s < P.t1;
      const int ks = s1 ? a.k1 : a.k2;
      if (!interior(s) && (ks & 3) != 0) {             // a 16-byte chunk that straddles the segment's end: per element
        const int kl = (s1 ? s : s - P.t1) * SK + fkq * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = kl + e < ks ? v0[e] : 0.f;
          v1[e] = kl + 4 + e < ks ? v1[e] : 0.f;
          if (MODE == 1) {
            u0[e] = kl + e < ks ? u0[e] : 0.f;
            u1[e] = kl + 4 + e < ks ? u1[e] : 0.f;
          }
        }
      }
      if (MODE == 1) {
        v0 = v0 - u0;
        v1 = v1 - u1;
      }
      split8(v0, v1);
      return;
    }
    const bool seg1 = s < P.t1;
    const int kl = (seg1 ? s : s - P.t1) * SK + fkq * 8;
    const float* p = seg1 ? rowp1 : rowp2;
    const int ks = seg1 ? a.k1 : a.k2;
    const bool ok = (p != nullptr);
    f32x4 v0, v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = (ok && kl + e < ks) ? ra[slot][0][e] : 0.f;
      v1[e] = (ok && kl + 4 + e < ks) ? ra[slot][1][e] : 0.f;
    }
    if (MODE == 1) {
      const bool oks = seg1 && (subp != nullptr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] -= (oks && kl + e < ks) ? rs[slot][0][e] : 0.f;
        v1[e] -= (oks && kl + 4 + e < ks) ? rs[slot][1][e] : 0.f;
      }
    }
    if (MODE == 2) {
      const bool oks = seg1 && (subp != nullptr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = (oks && kl + e < ks && rs[slot][0][e] > 0.f) ? v0[e] * a.mask_scale : 0.f;
        v1[e] = (oks && kl + 4 + e < ks && rs[slot][1][e] > 0.f) ? v1[e] * a.mask_scale : 0.f;
      }
    }
    if (MODE == 3) {
      const bool oks = seg1 && (bitrow != nullptr);
      const uint32_t bits = rb[slot];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = (oks && kl + e < ks && ((bits >> e) & 1u)) ? v0[e] * a.mask_scale : 0.f;
        v1[e] = (oks && kl + 4 + e < ks && ((bits >> (4 + e)) & 1u)) ? v1[e] * a.mask_scale : 0.f;
      }
    }
    split8(v0, v1);
    if (MODE == 2 || MODE == 3) {                      // side output (k1 % 4 == 0: chunks are whole).  Stored HERE, after the
      if (dzrow != nullptr && seg1) {                  // step's load wait: the store then has the whole next MFMA block to retire
        if (kl < ks) { *reinterpret_cast<f32x4*>(dzrow + kl) = v0; if (F16) dz_am = rr_amax4(dz_am, v0); }
        if (kl + 4 < ks) { *reinterpret_cast<f32x4*>(dzrow + kl + 4) = v1; if (F16) dz_am = rr_amax4(dz_am, v1); }
      }
    }
  };
  auto mfma_block = [&](int s) {
    if (CAN_PERSIST && uwave >= nw) return;             // (uniform) a wave without rows in a short last block: no MFMAs, no LDS reads
    const u32x4* Ws = reinterpret_cast<const u32x4*>(smem + (s & 1) * PANEL) + lane;
    if constexpr (F16) {
      const f16x8 h0 = as_f16x8(x0), h1 = as_f16x8(x1);
      u32x4 wa = Ws[0], wb = Ws[64];
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f16x8 w0 = as_f16x8(wa), w1 = as_f16x8(wb);
        if (j + 1 < NT) {
          wa = Ws[((j + 1) * 2 + 0) * 64];
          wb = Ws[((j + 1) * 2 + 1) * 64];
        }
        f32x4 c = acc[j];
        if (j + 1 == NT && NT != NTP && !full) continue;
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, h0, c, 0, 0, 0);   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h0, c, 0, 0, 0);
        acc[j] = c;
        if (j + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
      return;
    }
    const bf16x8 b0 = as_bf16x8(x0), b1 = as_bf16x8(x1), b2 = as_bf16x8(x2);
    // the three weight terms of tile j+1 are read while the six MFMAs of tile j run (pinned with sched_group_barrier:
    // left alone, the scheduler issues each ds_read right in front of its first use and waits for it)
    u32x4 wa = Ws[0], wb = Ws[64], wc = Ws[128];
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8 w0 = as_bf16x8(wa), w1 = as_bf16x8(wb), w2 = as_bf16x8(wc);
      if (j + 1 < NT) {
        wa = Ws[((j + 1) * 3 + 0) * 64];
        wb = Ws[((j + 1) * 3 + 1) * 64];
        wc = Ws[((j + 1) * 3 + 2) * 64];
      }
      f32x4 c = acc[j];
      if (j + 1 == NT && NT != NTP && !full) continue;  // (uniform) the narrower last column block has no tile NT-1
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, b0, c, 0, 0, 0);   // smallest terms first
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, b2, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, b0, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, b0, c, 0, 0, 0);
      acc[j] = c;
      if (j + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
  };
  // one k-step with compile-time slots.  vmcnt retires in issue order: the operand loads of step s+2 are issued AFTER the
  // weight image of step s+1, so "all but the youngest NX" = image landed, step s+1's chunks landed, step s+2's in flight.
  constexpr int NX = MODE == 0 ? 2 : (MODE == 3 ? 3 : 4);   // vector-memory instructions of one issue_x
  constexpr bool DEEP = WAVES == 12;                   // the 8-wave geometries (several workgroups per CU, <= 128 registers): one step ahead
  // Waves of the second half ("late") split their operand at the START of the step that consumes it, the first half at
  // the END of the step before: between two barriers every wave runs the same program, so without this all three waves of
  // a SIMD finish their MFMA blocks together and then run their ~150 VALU instructions of fixup() together, matrix pipe
  // idle (measured: 79 k shader cycles per 10 k-steps against 54.7 k of MFMA issue).  De-phased, one half's VALU runs
  // beside the other half's MFMAs.  Same values in the same order: only WHEN a wave converts its operand changes.
#ifdef RR_SPLIT_NO_LATE
  const bool late = false;
#else
  const bool late = DEEP && uwave >= WAVES / 2;
#endif
  // persistent form: step 0 of the NEXT row block, fetched by LDS-DMA into this wave's 2 KiB (lane-linear 16-byte slots:
  // chunk pair A | B) - no register crosses the store epilogue for it.  Same addresses as issue_x(0, .) would read.
  auto prefetch_next0 = [&]() {
    next_rows(g_cur + nw, g_end - (g_cur + nw) < WAVES ? static_cast<int>(g_end - (g_cur + nw)) : WAVES);   // (recomputed at the block switch: not live across the k-loop)
    const bool s1 = 0 < P.t1;
    const float* p = s1 ? xb1n : xb2n;
    const uint32_t dst = lds0 + PF_OFF + uwave * 2048;
    if (interior(0)) {
      rr_glds16(p, dst);
      rr_glds16(p + 4, dst + 1024);
    } else {
      int fq = fkq;
      asm volatile("" : "+v"(fq));                     // (see issue_x)
      const int kl = fq * 8, ks = s1 ? a.k1 : a.k2;
      rr_glds16(kl < ks ? p : rr_zero_row, dst);
      rr_glds16(kl + 4 < ks ? p + 4 : rr_zero_row, dst + 1024);
    }
  };
  auto step = [&](int s, int slot) {
    const bool more = s + 1 < nk, more2 = s + 2 < nk;
    const bool wrap = CAN_PERSIST && has_next;         // (uniform) the pipeline runs on into the next row block; nk is even
    // (step 0 of a block entered through the wrap below finds its operand already split by EVERY wave: one register
    // set - x0..x2 - crosses the store epilogue instead of two)
    if (late && !(CAN_PERSIST && s == 0 && wrapped)) fixup(s, DEEP ? slot : 0);
    if (more) issue_w(s + 1);
    else if (wrap) issue_w(0);                         // next block's first image: buffer 0, last read in step nk - 2
    if (DEEP) {
      if (more2) issue_x(s + 2, slot);
      else if (wrap && more) prefetch_next0();         // s = nk - 2: the next block's step-0 chunks go to LDS, not to registers
    } else {
      if (more) issue_x(s + 1, 0);
    }
    mfma_block(s);
    if (DEEP && (more2 || (wrap && more))) {           // (the two prefetch DMAs stand in for an issue_x: same count)
      if (NX == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (NX == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      rr_wait_vm0();
    }
    if (more && !late) fixup(s + 1, DEEP ? slot ^ 1 : 0);
    __syncthreads();
  };

  if (tid < BN / 4) {
    const int n = n0 + tid * 4;
    *reinterpret_cast<f32x4*>(bias_s + tid * 4) = ldg4((a.bias && n < a.N) ? a.bias + n : dummy);
  }
  issue_w(0);
  issue_x(0, 0);
  if (DEEP && nk > 1) issue_x(1, 1);
  rr_wait_vm0();
  if (!late) fixup(0, 0);
  __syncthreads();
  RR_STAMP(1);

  for (;;) {                                            // one pass per row block (a single pass unless persistent)
  if (CAN_PERSIST) {
    has_next = persist && g_cur + nw < g_end;
  }
  for (int s = 0; s < nk; s += 2) {
    step(s, 0);
    if (s + 1 < nk) step(s + 1, 1);
  }
  RR_STAMP(2);

  // ---- epilogue: the accumulator layout is that of linear_fast_kernel (a lane holds 4 consecutive columns of one row)
  // (persistent form: the lane id goes through an opaque asm per row block, so the epilogue's lane-derived offsets and
  // addresses are recomputed here - a few VALU instructions - instead of being hoisted out of the block loop, kept live
  // across the k-loop and spilled: 70 spilled registers / +60 MB of scratch writes per launch without this)
  if (F16) {                                           // back from the scaled operands: two exact powers of two (one product
#pragma unroll                                          // of them could leave the f32 exponent range where the result does not)
    for (int i = 0; i < NT; ++i) acc[i] = (acc[i] * ixs) * iws;
  }
  int lane_o = threadIdx.x;
#ifndef RR_PERSIST_HOIST
  if (CAN_PERSIST) asm volatile("" : "+v"(lane_o));
#endif
  const int tid = lane_o, lane = lane_o & 63, wave = lane_o >> 6;
  const int fr = lane & 15, fkq = lane >> 4;
  // this block's rows, from the (uniform) block index: nothing row-specific is carried through the k-loop in registers
  const int64_t m0 = g_cur * 16;
  const int64_t m = m0 + wave * 16 + fr;
  const bool row_ok = wave < nw && m < a.M;
  const int64_t mc = row_ok ? m : a.M - 1;
  const int nq = fkq * 4;
  float* crow = a.c + mc * a.ldc;
  const float* rrow = nullptr;
  if (a.residual) {
    const int64_t rr = a.residual_idx ? static_cast<int64_t>(a.residual_idx[mc]) : mc;
    if (rr >= 0) rrow = a.residual + rr * a.ldr;
  }
  const bool has_bias = a.bias != nullptr;
  const bool relu = a.act == RR_ACT_RELU;
  float c_am = 0.f;                                    // largest |C| this lane stored (rr_linear_args.c_amax_out)
  auto finish = [&](f32x4 v, int n) -> f32x4 {
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (P.drop_thr != 0u) {
      const uint64_t base = static_cast<uint64_t>(m) * static_cast<uint64_t>(a.N) + static_cast<uint64_t>(n);
      const uint32_t w = rr_hash_group(a.drop_seed, base >> 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rr_hash_lane(w, e) >= P.drop_thr ? v[e] * P.keep_scale : 0.f;
    }
    if (row_ok && n < a.N) {
      *reinterpret_cast<f32x4*>(crow + n) = v;
      if (F16) c_am = rr_amax4(c_am, v);
    }
    return v;
  };
  float* prow = a.c_pre ? a.c_pre + mc * a.ld_pre : nullptr;
  // optional fourth output: sign bits of what was stored (the mask a later dX GEMM needs: 1 bit instead of 4 bytes).
  // A lane holds 4 columns of a tile; lanes fkq and fkq^1 (16 lanes apart) make a byte = 8 consecutive columns, the
  // even one collects its 19 bytes in 5 registers and stores them once.
  const bool mb_on = a.mask_bits_out != nullptr;
  uint32_t mb[5] = {0u, 0u, 0u, 0u, 0u};
  const bool cs_on = a.colsum_partial != nullptr;
  const float wrow = (cs_on && row_ok) ? a.colsum_w[mc] : 0.f;
  float* const cs_lds = reinterpret_cast<float*>(smem + CS_OFF);    // [waves][BN] (8-wave geometries: the now idle weight image)
  // ---- row-contiguous epilogue (12-wave geometry).  In the accumulator layout a 16-lane group holds 16 ROWS x 16 bytes:
  // every global_load / global_store of the epilogue touches 64 different 128-byte lines for 1 KiB of payload, and the
  // address coalescer - not HBM - sets its time (measured: 10.8 us per 192-row block for the plain store epilogue, 21 us
  // with the residual read; a de-phased start of the workgroups changed nothing).  So the tile goes through LDS once: the
  // accumulators are written in their own layout, read back lane-linear (a lane = 4 consecutive columns of a row, 64 lanes
  // = 1 KiB of consecutive memory where ldc == N) and everything after the GEMM - bias, residual, ReLU, dropout, second
  // output, sign bits, the store - happens in that layout with fully coalesced accesses.  Same operations in the same
  // order per element, so the stored values are those of the accumulator-layout epilogue bit for bit.  The weighted
  // column sums (dX GEMMs: no bias / residual / activation) are taken from the accumulators before the transposition;
  // the rare combination of column sums WITH epilogue arithmetic keeps the accumulator-layout code below.
  const bool plain_epi = !has_bias && a.residual == nullptr && !relu && P.drop_thr == 0u && prow == nullptr && !mb_on;
  bool done = false;
  if (RS_EPI && (EPI & 1) == 1 && (!cs_on || plain_epi)) {
    done = true;
    if (cs_on) {
#pragma unroll
      for (int tc = 0; tc < NT; ++tc) {
        f32x4 t = acc[tc] * wrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = t[e];
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));   // row_shr:1
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));   // row_shr:2
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));   // row_shr:4
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));   // row_shr:8
          t[e] = x;
        }
        if (fr == 15) *reinterpret_cast<f32x4*>(cs_lds + wave * BN + tc * 16 + nq) = t;
      }
    }
    unsigned char* const region = smem + uwave * REGION;
    unsigned char* const bits_s = smem + CS_OFF + uwave * 640;       // [16 rows][40 bytes] (never together with column sums)
    if (mb_on) {                                                     // the two pad bytes of a row stay zero
#pragma unroll
      for (int d = lane; d < 160; d += 64) reinterpret_cast<uint32_t*>(bits_s)[d] = 0u;
    }
    const int nqv = (a.N - n0) / 4 < 4 * NT ? (a.N - n0) / 4 : 4 * NT;   // valid float4 columns of this column block
    const int64_t mw = m0 + uwave * 16;
    const bool res_on = a.residual != nullptr;
    constexpr int NI = (8 * 76 + 63) / 64;                           // lane-linear float4 reads per pass (76 per row, 8 rows)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if ((fr >> 3) == pass) {
#pragma unroll
        for (int tc = 0; tc < NT; ++tc)
          *reinterpret_cast<f32x4*>(region + (((fr & 7) * RS + tc * 4 + fkq) << 4)) = acc[tc];
      }
      // (wave-private region: a wave's LDS operations execute in order, no barrier)
      // lane-linear reads in groups of RG: the group's residual chunks are in flight together, then the group is finished
      // (all NI at once would need 40 registers next to the 76 accumulators that stay live until pass 1 is written)
      constexpr int RG = 2;
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += RG) {
        asm volatile("" ::: "memory");
        f32x4 rres[RG];
#pragma unroll
        for (int u = 0; u < RG; ++u) {
          const int i = i0 + u;
          if (i < NI && res_on) {
            const int q = (64 * i) / 76, rem = (64 * i) % 76;
            const bool wrap = rem + lane >= 76;
            const int r = q + (wrap ? 1 : 0), c4 = rem + lane - (wrap ? 76 : 0);
            const int64_t mm = mw + pass * 8 + r;
            const bool ok = (64 * i + lane < 8 * 76) && c4 < nqv && mm < a.M;
            const int64_t mmc = ok ? mm : 0;
            const int64_t rr = a.residual_idx ? static_cast<int64_t>(ldgi(a.residual_idx + mmc)) : mmc;
            rres[u] = ldg4((ok && rr >= 0) ? a.residual + rr * a.ldr + n0 + 4 * c4 : rr_zero_chunk);
          }
        }
#pragma unroll
        for (int u = 0; u < RG; ++u) {
          const int i = i0 + u;
          if (i >= NI) continue;
          const int q = (64 * i) / 76, rem = (64 * i) % 76;
          const bool wrap = rem + lane >= 76;
          const int r = q + (wrap ? 1 : 0), c4 = rem + lane - (wrap ? 76 : 0);
          const int64_t mm = mw + pass * 8 + r;
          const bool ok = (64 * i + lane < 8 * 76) && c4 < nqv && mm < a.M;
          const int n = n0 + 4 * c4;
          f32x4 v = *reinterpret_cast<const f32x4*>(region + ((r * RS + c4) << 4));
          if (has_bias) v = v + *reinterpret_cast<const f32x4*>(bias_s + 4 * (c4 < 4 * NT ? c4 : 0));
          if (res_on) v = v + rres[u];
          if (prow != nullptr && ok) *reinterpret_cast<f32x4*>(a.c_pre + mm * a.ld_pre + n) = v;
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (P.drop_thr != 0u) {
            const uint64_t base = static_cast<uint64_t>(mm) * static_cast<uint64_t>(a.N) + static_cast<uint64_t>(n);
            const uint32_t w = rr_hash_group(a.drop_seed, base >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rr_hash_lane(w, e) >= P.drop_thr ? v[e] * P.keep_scale : 0.f;
          }
          if (ok) {
            *reinterpret_cast<f32x4*>(a.c + mm * a.ldc + n) = v;
            if (F16) c_am = rr_amax4(c_am, v);
          }
          if (mb_on) {                                               // lanes l, l^1 hold the two halves of 8 consecutive columns
            uint32_t nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            if (c4 >= nqv) nib = 0u;                                  // columns past N: zero bits
            const uint32_t other = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, static_cast<int>(nib)));
            const int j = c4 >> 1;                                   // columns 8j .. 8j+7: tile j/2, half j&1
            if ((c4 & 1) == 0 && 64 * i + lane < 8 * 76 && c4 < 4 * NT)
              bits_s[(pass * 8 + r) * 40 + (j & 1) * 20 + (j >> 1)] = static_cast<uint8_t>(nib | (other << 4));
          }
        }
      }
    }
    if (mb_on) {                                                     // 16 rows x 10 dwords, coalesced
      const int64_t rowb = mask_bits_row(a.N);
#pragma unroll
      for (int d = lane; d < 160; d += 64) {
        const int r = d / 10, w = d - 10 * r;
        if (mw + r < a.M)
          *reinterpret_cast<uint32_t*>(a.mask_bits_out + (mw + r) * rowb + by * 40 + 4 * w) = reinterpret_cast<const uint32_t*>(bits_s)[d];
      }
    }
  }
  if (!done) {
    // (with RR_EPI_MODE 1 the 12-wave forward forms that carry a residual take the row-contiguous epilogue above: this
    // instantiation then never sees one, and its register ring is not needed)
    constexpr bool MAY_RES = !(RR_EPI_MODE == 1 && WAVES == 12 && (MODE == 0 || MODE == 1) && (EPI & 1) == 0);
    const bool res_ok = MAY_RES && rrow != nullptr;
    const float* rbase = res_ok ? rrow : dummy;
    constexpr int D = MAY_RES ? 4 : 0;
    f32x4 ring[D + 1];
    auto ldres = [&](int tc) {
      const int n = n0 + tc * 16 + nq;
      return ldg4(rbase + ((res_ok && n < a.N) ? n : 0));
    };
#pragma unroll
    for (int t = 0; t < D; ++t)
      if (t < NT) ring[t] = ldres(t);
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
      const int n = n0 + tc * 16 + nq;
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + tc * 16 + nq);
      f32x4 v = acc[tc];
      const f32x4 vb = v + b;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = has_bias ? vb[e] : v[e];
      if (MAY_RES) {
        const f32x4 vr = v + ring[tc % (D + 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = res_ok ? vr[e] : v[e];
        if (tc + D < NT) ring[(tc + D) % (D + 1)] = ldres(tc + D);
      }
      if (prow != nullptr && row_ok && n < a.N) *reinterpret_cast<f32x4*>(prow + n) = v;
      const f32x4 stored = finish(v, n);
      if (mb_on) {
        const uint32_t nib = (stored.x > 0.f ? 1u : 0u) | (stored.y > 0.f ? 2u : 0u) | (stored.z > 0.f ? 4u : 0u) | (stored.w > 0.f ? 8u : 0u);
        const uint32_t other = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, static_cast<int>(nib)));
        if (NT == 19) {
          mb[tc >> 2] |= (nib | (other << 4)) << (8 * (tc & 3));
        } else if (row_ok && (fkq & 1) == 0 && (full || tc + 1 < NT)) {       // narrow column blocks: one byte store per tile
          const int tg = t0 + tc;
          a.mask_bits_out[m * mask_bits_row(a.N) + (tg / 19) * 40 + (fkq >> 1) * 20 + (tg % 19)] =
              static_cast<uint8_t>(nib | (other << 4));
        }
      }
      if (cs_on) {
        f32x4 t = stored * wrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = t[e];
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));   // row_shr:1
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));   // row_shr:2
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));   // row_shr:4
          x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));   // row_shr:8
          t[e] = x;
        }
        if (fr == 15) *reinterpret_cast<f32x4*>(cs_lds + wave * BN + tc * 16 + nq) = t;
      }
    }
  }
  if (!done && NT == 19 && mb_on && row_ok && (fkq & 1) == 0) {
    uint32_t* d = reinterpret_cast<uint32_t*>(a.mask_bits_out + m * mask_bits_row(a.N) + by * 40 + (fkq >> 1) * 20);
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = mb[i];
  }
  // (the magnitude outputs exist in the two-f16-term instantiations only: the three-term kernels keep their registers)
  if (F16 && a.c_amax_out != nullptr) rr_amax_commit_wave(c_am, amx);
  if (F16 && (MODE == 2 || MODE == 3) && a.dz_amax_out != nullptr && by == 0) {
    rr_amax_commit_wave(dz_am, amx + 1);
    dz_am = 0.f;
  }
  if (cs_on) {                                         // one partial row per 64 rows (rr_linear_colsum_rows) = per 4 waves
    __syncthreads();
    static_assert(WAVES % 4 == 0 && (WAVES / 4) * (BN / 4) <= S_THREADS, "colsum slices");
    if (tid < (WAVES / 4) * (BN / 4)) {
      const int h = tid / (BN / 4);
      const int q = tid - h * (BN / 4);
      const int n = n0 + q * 4;
      const float* base = cs_lds + h * 4 * BN + q * 4;
      const f32x4 s01 = ld4(base) + ld4(base + BN);
      const f32x4 s23 = ld4(base + 2 * BN) + ld4(base + 3 * BN);
      if (n < a.N && h * 4 < nw && m0 + h * 64 < a.M)    // (one partial row per 64-row unit; a short last block owns fewer)
        *reinterpret_cast<f32x4*>(a.colsum_partial + (g_cur / 4 + h) * a.ld_partial + n) = s01 + s23;
    }
  }
  if (!(CAN_PERSIST && has_next)) break;
  // next row block: its step-0 operand is split (early waves) or loaded (late waves), its step-1 chunks are in flight, its
  // first weight image is in LDS buffer 0 - the state the prologue leaves behind
  if (cs_on) __syncthreads();                          // the column-sum staging of this block has been read
  wrapped = true;
  g_cur += nw;
  nw = g_end - g_cur < WAVES ? static_cast<int>(g_end - g_cur) : WAVES;
  next_rows(g_cur, nw);
  xb1 = xb1n;
  xb2 = xb2n;
  if (nk > 1) issue_x(1, 1);                           // step 1: one MFMA block to land instead of two
  {
    const unsigned char* pf = smem + PF_OFF + uwave * 2048 + lane * 16;
    ra[0][0] = *reinterpret_cast<const f32x4*>(pf);
    ra[0][1] = *reinterpret_cast<const f32x4*>(pf + 1024);
  }
  fixup(0, 0);                                         // every wave (step 0 of a wrapped block skips the late split)
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4(0.f);
  }
  if (F16 && (a.c_amax_out != nullptr || a.dz_amax_out != nullptr)) {  // (uniform) one atomic per workgroup and output
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int vc = amx[0], vd = amx[1];
      if (a.c_amax_out != nullptr) rr_amax_put(a.c_amax_out, __uint_as_float(vc));
      if (a.dz_amax_out != nullptr) rr_amax_put(a.dz_amax_out, __uint_as_float(vd));
    }
  }
#ifdef RR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RR_STAMP(3);
#endif
}

constexpr int PACK_SCALE_BLOCKS = 16;
// weight terms of the split path: dst = [k-step][column tile][term 0..2][lane 0..63][8 bf16], the LDS image of a k-step
__device__ __forceinline__ void split_one(float x, uint16_t& t0, uint16_t& t1, uint16_t& t2) {
  uint32_t p0, p1, p2;
  split_pair(x, 0.f, p0, p1, p2);
  t0 = static_cast<uint16_t>(p0 & 0xffffu);
  t1 = static_cast<uint16_t>(p1 & 0xffffu);
  t2 = static_cast<uint16_t>(p2 & 0xffffu);
}
__host__ __device__ constexpr int split_nt(int N) { return N <= 64 ? 4 : (N <= 160 ? 10 : (N <= 304 ? 19 : 38)); }

__device__ __forceinline__ void pack_split_elem(const rr_pack_desc& q, int64_t e, float S) {
  const int nt = split_nt(q.rows);
  const int t1 = r32(q.k1) / SK;
  const int el = static_cast<int>(e & 7), lane = static_cast<int>((e >> 3) & 63);
  const int64_t blk = e >> 9;                          // (s * nt + j)
  const int j = static_cast<int>(blk % nt), s = static_cast<int>(blk / nt);
  const int n = j * 16 + (lane & 15);
  const int kk = (s < t1 ? s : s - t1) * SK + (lane >> 4) * 8 + el;
  int lc = -1;
  if (s < t1) {
    if (kk < q.k1) lc = kk;
  } else if (kk < q.k2) {
    lc = q.k1 + kk;
  }
  float v = 0.f;
  if (lc >= 0 && n < q.rows)
    v = q.transpose ? q.src[static_cast<int64_t>(lc) * q.ld_src + q.c0 + n] : q.src[static_cast<int64_t>(n) * q.ld_src + q.c0 + lc];
  if (q.split == 2) {                                  // two f16 terms of S * L (S from pack_scale_kernel's partial maxima)
    v *= S;
    _Float16* d = reinterpret_cast<_Float16*>(q.dst) + blk * 2 * 512 + lane * 8 + el;
    const _Float16 h = static_cast<_Float16>(v);
    d[0] = h;
    d[512] = static_cast<_Float16>(v - static_cast<float>(h));
    return;
  }
  uint16_t* d = reinterpret_cast<uint16_t*>(q.dst) + blk * 3 * 512 + lane * 8 + el;
  split_one(v, d[0], d[512], d[1024]);
}

// split = 2: largest magnitude of each weight, as PACK_SCALE_BLOCKS partial maxima behind its last image (floats 4 .. of the
// trailer; pack_split_kernel folds them into S = the power of two with 2^14 <= S max|L| < 2^15 and stores S at float 0)
__global__ void __launch_bounds__(1024) pack_scale_kernel(const PackMany P) {
  const rr_pack_desc& q = P.d[blockIdx.y];
  if (q.split != 2) return;
  __shared__ float part[16];
  const int K = q.k1 + q.k2;
  const int64_t total = static_cast<int64_t>(q.rows) * K;
  auto val = [&](int64_t e) -> float {                  // consecutive threads read consecutive memory in either orientation
    const int n = q.transpose ? static_cast<int>(e % q.rows) : static_cast<int>(e / K);
    const int lc = q.transpose ? static_cast<int>(e / q.rows) : static_cast<int>(e % K);
    return fabsf(q.transpose ? q.src[static_cast<int64_t>(lc) * q.ld_src + q.c0 + n] : q.src[static_cast<int64_t>(n) * q.ld_src + q.c0 + lc]);
  };
  float m4[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int64_t ST = 1024 * PACK_SCALE_BLOCKS;
  int64_t e = static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x;
  for (; e + 3 * ST < total; e += 4 * ST) {
#pragma unroll
    for (int u = 0; u < 4; ++u) m4[u] = fmaxf(m4[u], val(e + u * ST));
  }
  for (; e < total; e += ST) m4[0] = fmaxf(m4[0], val(e));
  float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, part[i]);
    const int64_t nblk = static_cast<int64_t>((r32(q.k1) + r32(q.k2)) / SK) * split_nt(q.rows);
    q.dst[nblk * 512 + 4 + blockIdx.x] = m;
  }
}

// (one launch packs EVERY weight of a pass: the f32 panels of the FFN head as well - blockIdx.y picks the weight, its
// `split` the layout; a second launch for the plain ones cost a ~7 us kernel + a launch boundary in front of every forward)
__global__ void __launch_bounds__(256) pack_split_kernel(const PackMany P) {
  const rr_pack_desc& q = P.d[blockIdx.y];
  if (!q.split) {
    pack_plain_desc(q);
    return;
  }
  const int64_t total = static_cast<int64_t>((r32(q.k1) + r32(q.k2)) / SK) * split_nt(q.rows) * 512;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float S = 1.f;
  if (q.split == 2) {                                  // fold pack_scale_kernel's partial maxima; thread 0 leaves S for the GEMM
    const int64_t nblk = total / 512;
    const float* part = q.dst + nblk * 512 + 4;
    float m = part[0];
#pragma unroll
    for (int i = 1; i < PACK_SCALE_BLOCKS; ++i) m = fmaxf(m, part[i]);
    S = rr_pow2(14 - rr_f16_exp(m));
    if (blockIdx.x == 0 && threadIdx.x == 0) q.dst[nblk * 512] = S;
  }
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) pack_split_elem(q, e, S);
}

// ======================================================================== weight gradient
constexpr int WT = 5;               // 5x5 MFMA tiles (80 x 80) per wave
constexpr int WBN = 160;            // workgroup output tile: 160 (n) x 160 (k), 2x2 waves
constexpr int WLD = 176;            // LDS row stride (== 16 mod 32 -> ds_read_b32 conflict-free)
constexpr int WMT = 16;             // rows of M per staged tile

struct WgradParams {
  rr_wgrad_args a;
  int k1p;              // segment-2 start column in the extended X (k1 rounded up to 4)
  int kext;             // k1p + k2 + 1 (last column = ones -> dbias)
  int nblk_n, nblk_k;   // output tiles
  int64_t rows_per_chunk;
  int nchunks;
  int flags;            // F_A1_VEC (x1), F_A2_VEC (x2), F_SUB_VEC, F_MASK_VEC (mask), F_EPI_VEC (dy)
  int64_t slab;         // floats per partial slab = N*(k1+k2) + N
};

__global__ void __launch_bounds__(THREADS) wgrad_kernel(const WgradParams P) {
  __shared__ __attribute__((aligned(16))) float lds[2][2 * WMT * WLD];
  const rr_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (id % 8 share an L2), so the
  // nblk_n * nblk_k output tiles of ONE M-chunk get consecutive slots of one XCD: the second reader
  // of every dZ / X row hits that XCD's L2 instead of HBM (speed only; any placement is correct).
  const int nt = P.nblk_n * P.nblk_k;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % nt, chunk = (slot / nt) * 8 + xcd;
  if (chunk >= P.nchunks) return;
  const int bn = tile / P.nblk_k, bk = tile % P.nblk_k;
  const int nb = bn * WBN, kb = bk * WBN;
  const int64_t mbeg = static_cast<int64_t>(chunk) * P.rows_per_chunk;
  int64_t mend = mbeg + P.rows_per_chunk;
  if (mend > a.M) mend = a.M;
  const int flags = P.flags;
  const int K = a.k1 + a.k2;

  // staging: 2 tiles x 16 rows x 40 float4 = 1280 chunks, 5 per thread
  auto load_chunk_z = [&](int64_t mrow, int col) -> f32x4 {       // dZ[mrow][nb+col .. +3]
    f32x4 v = f32x4(0.f);
    const int n = nb + col;
    if (mrow >= mend || n >= a.N) return v;
    v = load_chunk(a.dy + mrow * a.ld_dy, n, a.N, flags & F_EPI_VEC);
    if (a.mask) v = apply_mask(v, load_chunk(a.mask + mrow * a.ld_mask, n, a.N, flags & F_MASK_VEC), a.mask_scale);
    return v;
  };
  auto load_chunk_x = [&](int64_t mrow, int col) -> f32x4 {       // X_ext[mrow][kb+col .. +3]
    f32x4 v = f32x4(0.f);
    const int k = kb + col;
    if (mrow >= mend || k >= P.kext) return v;
    if (k < P.k1p) {
      const float* p = nullptr;
      if (a.x1_idx) {
        const int32_t j = a.x1_idx[mrow];
        if (j >= 0) p = a.x1 + static_cast<int64_t>(j) * a.ldx1;
      } else {
        p = a.x1 + mrow * a.ldx1;
      }
      v = load_chunk(p, k, a.k1, flags & F_A1_VEC);
      if (a.x1_sub) {
        const float* sp = nullptr;
        if (a.x1_sub_idx) {
          const int32_t j = a.x1_sub_idx[mrow];
          if (j >= 0) sp = a.x1_sub + static_cast<int64_t>(j) * a.ldx1_sub;
        } else {
          sp = a.x1_sub + mrow * a.ldx1_sub;
        }
        v = v - load_chunk(sp, k, a.k1, flags & F_SUB_VEC);
      }
    } else {
      const int k2 = k - P.k1p;
      if (a.k2 > 0) v = load_chunk(a.x2 + mrow * a.ldx2, k2, a.k2, flags & F_A2_VEC);
      // the ones column (bias gradient) sits right after segment 2
      const int one = a.k2 - k2;                 // position of the ones column inside this chunk
      if (one >= 0 && one < 4) v[one] = 1.0f;
    }
    return v;
  };
  constexpr int CH = 5;
  auto load_tiles = [&](int64_t mt, f32x4 (&r)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int f = i * THREADS + tid;           // 0 .. 1279
      const int which = f >= 640 ? 1 : 0;
      const int g = f - which * 640;
      const int row = g / 40, col = (g - row * 40) * 4;
      r[i] = which ? load_chunk_x(mt + row, col) : load_chunk_z(mt + row, col);
    }
  };
  auto store_tiles = [&](int buf, const f32x4 (&r)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int f = i * THREADS + tid;
      const int which = f >= 640 ? 1 : 0;
      const int g = f - which * 640;
      const int row = g / 40, col = (g - row * 40) * 4;
      *reinterpret_cast<f32x4*>(&lds[buf][which * WMT * WLD + row * WLD + col]) = r[i];
    }
  };

  f32x4 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = f32x4(0.f);

  const int wn = (wave >> 1) * (WT * 16), wk = (wave & 1) * (WT * 16);   // wave's corner inside the 160x160 tile
  const int fr = lane & 15, fq = lane >> 4;

  f32x4 r[CH];
  const int64_t ntiles = (mend > mbeg) ? (mend - mbeg + WMT - 1) / WMT : 0;
  if (ntiles > 0) {
    load_tiles(mbeg, r);
    store_tiles(0, r);
  }
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = static_cast<int>(t & 1);
    const bool more = t + 1 < ntiles;
    if (more) load_tiles(mbeg + (t + 1) * WMT, r);
    const float* Zs = lds[cur];
    const float* Xs = lds[cur] + WMT * WLD;
#pragma unroll
    for (int kk = 0; kk < WMT / 4; ++kk) {
      float zf[WT], xf[WT];
      const int row = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        zf[i] = Zs[row * WLD + wn + i * 16 + fr];
        xf[i] = Xs[row * WLD + wk + i * 16 + fr];
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1, r);
    __syncthreads();
  }

  // partial slab [chunk][ N*K (dw, row-major) | N (dbias) ]; D[i = n][j = k]: lane -> n = .. + fq*4 + e, k = .. + fr
  float* slab = static_cast<float*>(a.workspace) + static_cast<int64_t>(chunk) * P.slab;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int kx = kb + wk + j * 16 + fr;       // extended column
      if (kx >= P.kext) continue;
      int kreal = -1;                             // -1: dead pad column, -2: ones column
      if (kx < a.k1) kreal = kx;
      else if (kx >= P.k1p && kx < P.k1p + a.k2) kreal = a.k1 + (kx - P.k1p);
      else if (kx == P.k1p + a.k2) kreal = -2;
      if (kreal == -1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nb + wn + i * 16 + fq * 4 + e;
        if (n >= a.N) continue;
        if (kreal >= 0) slab[static_cast<int64_t>(n) * K + kreal] = acc[i][j][e];
        else slab[static_cast<int64_t>(a.N) * K + n] = acc[i][j][e];
      }
    }
  }
}


// ------------------------------------------------------------------------ wgrad fast path
// Hot case: every operand 16-byte addressable, N % 4 == 0.  A 16-row tile is 640 dZ chunks +
// 16 * (k-block width / 4) X chunks of 16 bytes; thread t owns dZ chunks {t, t+256, t+512 (t<128)}
// and up to three X chunks, so every slot's role is known before the loop.
//
// The loop is VALU-bound if the loader is written naively (per-tile 64-bit address products,
// per-element bounds selects: ~500 VALU + ~300 SALU instructions per tile against 100 MFMAs, measured
// 61 % matrix-pipe duty).  So all per-tile work that can be hoisted is hoisted:
//  * every streamed operand is a per-slot POINTER that advances by 16 rows per tile (one 64-bit add);
//  * a slot that must read as zero (column block tail, k-block tail, unused slot) points at a zero
//    chunk with stride 0 — no validity select in the loop; the ones column (bias gradient) is a
//    constant {1,0,0,0} chunk when it starts a chunk (k2 % 4 == 0);
//  * gathered rows cost one v_mad_u64_u32 (index x row pitch + column pointer) and one select
//    (index < 0 -> zero chunk); indices are fetched one tile ahead from an advancing pointer;
//  * rows past the end of the M-chunk exist only in its last tile, which takes a separate
//    instantiation of the issue code (selects to the zero chunk); the steady state has none;
//  * partial 16-byte chunks (k1 % 4 or k2 % 4 != 0) are patched per element only under a uniform flag.
// What remains per tile: the loads, the pointer bumps, the ReLU-mask select and the subtraction.
__device__ __attribute__((aligned(16))) const float rr_one_chunk[4] = {1.f, 0.f, 0.f, 0.f};

template <bool HAS_MASK, bool HAS_SUB, int WTK>
__global__ void __launch_bounds__(THREADS, 2) wgrad_fast_kernel(const WgradParams P) {
  constexpr int KB = 32 * WTK;                          // columns per k-block (160 / 128 / 96)
  constexpr int XC = KB / 4;                            // 16-byte X chunks per row
  constexpr int S = 3;                                  // slots per operand per thread
  __shared__ __attribute__((aligned(16))) float lds[2][2 * WMT * WLD + 4];   // + one dump chunk for unused staging slots
  const rr_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (id % 8 share an L2), so the
  // nblk_n * nblk_k output tiles of ONE M-chunk get consecutive slots of one XCD: the second reader
  // of every dZ / X row hits that XCD's L2 instead of HBM (speed only; any placement is correct).
  const int nt = P.nblk_n * P.nblk_k;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % nt, chunk = (slot / nt) * 8 + xcd;
  if (chunk >= P.nchunks) return;
  RR_STAMP(0);
  const int bn = tile / P.nblk_k, bk = tile % P.nblk_k;
  const int nb = bn * WBN, kb = bk * KB;
  const int64_t mbeg = static_cast<int64_t>(chunk) * P.rows_per_chunk;
  int64_t mend = mbeg + P.rows_per_chunk;
  if (mend > a.M) mend = a.M;
  const int K = a.k1 + a.k2;
  const int nrows = static_cast<int>(mend - mbeg);      // rows of this M-chunk (> 0: chunk < nchunks)
  const int ntiles = (nrows + WMT - 1) / WMT;
  const float* const zero = rr_zero_chunk;
  const bool partial = (a.k1 & 3) != 0 || (a.k2 & 3) != 0;

  // ---- slot roles (loop invariant)
  enum : int { X_NONE = 0, X_DIRECT = 1, X_GATHER = 2, X_ONES = 3 };
  int zrow[S], zoff[S], xrow[S], xoff[S], xkind[S], nval[S], onee[S];
  const float* pz[S];                                   // dZ chunk of this slot in the current tile (advances)
  const float* pm[S];                                   // ReLU-mask chunk
  const float* px[S];                                   // X chunk (direct) or column pointer into row 0 (gather)
  const float* ps[S];                                   // subtract source, same two forms
  const int32_t* pi[S];                                 // gather index of the slot's row, one tile ahead
  const int32_t* pj[S];
  int gi[S], gj[S];                                     // -1 for gathered slots (index offset mask), 0 otherwise
  uint32_t zstep[S], mstep[S], xstep[S], sstep[S];      // bytes per tile (0 for constant chunks); 16 rows * pitch < 4 GiB
  bool sgather[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int g = tid + i * THREADS;
    // dZ
    const bool zuse = g < WMT * 40;
    zrow[i] = zuse ? g / 40 : 0;
    const int zcol = zuse ? (g - zrow[i] * 40) * 4 : 0;
    zoff[i] = (zuse ? zrow[i] * WLD + zcol : 2 * WMT * WLD) / 4;      // in 16-byte units: the store is a ds_write_b128
    const bool zok = zuse && (nb + zcol < a.N);
    pz[i] = zok ? a.dy + (mbeg + zrow[i]) * a.ld_dy + nb + zcol : zero;
    zstep[i] = zok ? static_cast<uint32_t>(WMT * 4 * a.ld_dy) : 0u;
    pm[i] = zero;
    mstep[i] = 0;
    if (HAS_MASK && zok) {
      pm[i] = a.mask + (mbeg + zrow[i]) * a.ld_mask + nb + zcol;
      mstep[i] = static_cast<uint32_t>(WMT * 4 * a.ld_mask);
    }
    // X
    const bool xuse = g < WMT * XC;
    xrow[i] = xuse ? g / XC : 0;
    const int xc0 = xuse ? (g - xrow[i] * XC) * 4 : 0;
    xoff[i] = (xuse ? WMT * WLD + xrow[i] * WLD + xc0 : 2 * WMT * WLD) / 4;
    const int kx = kb + xc0;                            // extended column of the chunk
    xkind[i] = X_NONE;
    px[i] = zero; ps[i] = zero; xstep[i] = 0; sstep[i] = 0; sgather[i] = false;
    pi[i] = reinterpret_cast<const int32_t*>(rr_zero_chunk); pj[i] = pi[i];   // non-gather slots read index 0
    gi[i] = 0; gj[i] = 0;
    nval[i] = 4; onee[i] = -1;
    if (xuse && kx < a.k1) {                            // segment 1
      nval[i] = min(4, a.k1 - kx);
      if (a.x1_idx) {
        xkind[i] = X_GATHER;
        px[i] = a.x1 + kx;
        pi[i] = a.x1_idx + mbeg + xrow[i];
        gi[i] = -1;
      } else {
        xkind[i] = X_DIRECT;
        px[i] = a.x1 + (mbeg + xrow[i]) * a.ldx1 + kx;
        xstep[i] = static_cast<uint32_t>(WMT * 4 * a.ldx1);
      }
      if (HAS_SUB) {
        if (a.x1_sub_idx) {
          sgather[i] = true;
          ps[i] = a.x1_sub + kx;
          pj[i] = a.x1_sub_idx + mbeg + xrow[i];
          gj[i] = -1;
        } else {
          ps[i] = a.x1_sub + (mbeg + xrow[i]) * a.ldx1_sub + kx;
          sstep[i] = static_cast<uint32_t>(WMT * 4 * a.ldx1_sub);
        }
      }
    } else if (xuse && kx >= P.k1p && kx < P.kext) {    // segment 2 and / or the ones column
      const int c2 = kx - P.k1p;
      if (c2 < a.k2) {
        xkind[i] = X_DIRECT;
        px[i] = a.x2 + (mbeg + xrow[i]) * a.ldx2 + c2;
        xstep[i] = static_cast<uint32_t>(WMT * 4 * a.ldx2);
        nval[i] = min(4, a.k2 - c2);
        if (a.k2 - c2 < 4) onee[i] = a.k2 - c2;         // ones column shares this chunk (k2 % 4 != 0)
      } else {                                          // c2 == k2 (k2 % 4 == 0): the chunk is {1, 0, 0, 0}
        xkind[i] = X_ONES;
        px[i] = rr_one_chunk;
        nval[i] = 0; onee[i] = 0;
      }
    }
  }

  f32x4 zv[S], zm[S], xv[S], xs[S];
  bool xrv[S];                                          // the X slot's row is inside the M-chunk (latched at issue)
  int32_t ia[S], is[S];
#pragma unroll
  for (int i = 0; i < S; ++i) { ia[i] = 0; is[i] = 0; xrv[i] = true; }

  // indices for the tile whose first row is `r0` (relative to mbeg); rows past the chunk read the last row's index
  auto fetch_idx = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const int over = r0 + xrow[i] - (nrows - 1);      // > 0: past the end -> step back to the last row
      const int off = r0 - (over > 0 ? over : 0);
      ia[i] = ldgi(pi[i] + (off & gi[i]));              // direct slots always read index 0 (>= 0, adds 0 rows)
      if (HAS_SUB) is[i] = ldgi(pj[i] + (off & gj[i]));
    }
  };
  // rows_left < 16 only in the last tile of the M-chunk: those rows read the zero chunk.  No branches: every
  // load is issued, the pointer is what gets selected (a load under a per-lane branch makes hipcc wait at the join)
  auto issue = [&](int rows_left) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const bool zr = zrow[i] < rows_left;
      zv[i] = ldg4(zr ? pz[i] : zero);
      if (HAS_MASK) zm[i] = ldg4(zr ? pm[i] : zero);
      pz[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pz[i]) + zstep[i]);
      if (HAS_MASK) pm[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pm[i]) + mstep[i]);
      const bool xr = xrow[i] < rows_left;
      xrv[i] = xr;
      const float* p = px[i] + static_cast<uint64_t>(static_cast<uint32_t>(ia[i])) * static_cast<uint32_t>(a.ldx1);
      xv[i] = ldg4((xr && ia[i] >= 0) ? p : zero);
      px[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(px[i]) + xstep[i]);
      if (HAS_SUB) {
        const float* q = ps[i] + static_cast<uint64_t>(static_cast<uint32_t>(is[i])) * static_cast<uint32_t>(a.ldx1_sub);
        xs[i] = ldg4((xr && is[i] >= 0) ? q : zero);
        ps[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ps[i]) + sstep[i]);
      }
    }
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < S; ++i) {
      {
        f32x4 z = zv[i];
        if (HAS_MASK) {
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = zm[i][e] > 0.f ? zv[i][e] * a.mask_scale : 0.f;
        }
        reinterpret_cast<f32x4*>(lds[buf])[zoff[i]] = z;
      }
      {
        f32x4 x = xv[i];
        if (HAS_SUB) x = xv[i] - xs[i];
        if (partial) {                                  // k1 % 4 or k2 % 4 != 0: patch the chunk per element
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = e < nval[i] ? xv[i][e] : 0.f;
            if (HAS_SUB) u -= e < nval[i] ? xs[i][e] : 0.f;
            if (e == onee[i]) u = xrv[i] ? 1.0f : 0.f;
            x[e] = u;
          }
        }
        reinterpret_cast<f32x4*>(lds[buf])[xoff[i]] = x;
      }
    }
  };

  f32x4 acc[WT][WTK];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WTK; ++j) acc[i][j] = f32x4(0.f);

  const int wn = (wave >> 1) * (WT * 16), wk = (wave & 1) * (WTK * 16);
  const int fr = lane & 15, fq = lane >> 4;
  // tile 0
  fetch_idx(0);
  issue(nrows);
  fetch_idx(WMT);
  commit(0);
  __syncthreads();
  RR_STAMP(1);
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < ntiles;
#ifdef RR_TRACE_LOOP
#define RR_LSTAMP(q)                                                                                      \
    do {                                                                                                   \
      if (rr_trace_buf && (threadIdx.x & 63) == 0 && blockIdx.x < 2 && t < 48)                             \
        rr_trace_buf[4096 * 8 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 48 + t) * 8 + (q)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RR_LSTAMP(q)
#endif
    RR_LSTAMP(0);
    if (more) {
      const int left = nrows - (t + 1) * WMT;           // rows of tile t+1 (uses the indices fetched one tile ago)
      issue(left);
      fetch_idx((t + 2) * WMT);
    }
    RR_LSTAMP(1);
    const float* Zs = lds[cur];
    const float* Xs = lds[cur] + WMT * WLD;
#pragma unroll
    for (int kk = 0; kk < WMT / 4; ++kk) {
      float zf[WT], xf[WTK];
      const int row = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < WT; ++i) zf[i] = Zs[row * WLD + wn + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < WTK; ++j) xf[j] = Xs[row * WLD + wk + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WTK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    RR_LSTAMP(2);
#ifdef RR_TRACE_LOOP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RR_LSTAMP(3);
#endif
    if (more) commit(cur ^ 1);
    RR_LSTAMP(4);
    __syncthreads();
    RR_LSTAMP(5);
  }

  RR_STAMP(2);
  float* slab = static_cast<float*>(a.workspace) + static_cast<int64_t>(chunk) * P.slab;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int j = 0; j < WTK; ++j) {
      const int kx = kb + wk + j * 16 + fr;
      if (kx >= P.kext) continue;
      int kreal = -1;
      if (kx < a.k1) kreal = kx;
      else if (kx >= P.k1p && kx < P.k1p + a.k2) kreal = a.k1 + (kx - P.k1p);
      else if (kx == P.k1p + a.k2) kreal = -2;
      if (kreal == -1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nb + wn + i * 16 + fq * 4 + e;
        if (n >= a.N) continue;
        if (kreal >= 0) slab[static_cast<int64_t>(n) * K + kreal] = acc[i][j][e];
        else slab[static_cast<int64_t>(a.N) * K + n] = acc[i][j][e];
      }
    }
  }
#ifdef RR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RR_STAMP(3);
#endif
}

// ------------------------------------------------------------------------ weight gradient, split path
// dW = dZ^T X on the bf16 matrix core with the three-term operand split of linear_split_kernel: here BOTH operands are
// activations, so each staged element is split once per workgroup when it is written to LDS (row-major bf16 term
// images [term][32 rows of M][columns]), and the MFMA operands - 8 consecutive rows of M for one column - come out of
// ds_read_b64_tr_b16, the transposing LDS read (a 16-lane group reads 4 rows x 16 columns and each lane receives one
// column).  Output tile, M-chunking, slab layout and the fixed-order reduction are those of wgrad_fast_kernel.
// Loader: thread t owns row t/8 of the 32-row tile and the 16-byte chunks (t%8) + 8 i of that row: one row pointer
// per operand, one gather index per tile.  One LDS stage of 60 KB (two workgroups per CU overlap each other's
// staging with MFMA); image rows are 64 (mod 128) bytes apart so the 8-byte term stores are conflict-free, and the
// 32-byte column blocks of rows 8-15 / 24-31 are swapped pairwise (XOR 32) so that the two row quads a 32-lane half
// reads in one transposed read (rows r..r+3 and r+8..r+11) fall on different banks.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* rr_lds_s16x4;

__device__ __forceinline__ u32x4 tr_read8(const unsigned char* p, int rowbytes) {     // rows r..r+3 and r+4..r+7
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rr_lds_s16x4)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rr_lds_s16x4)(p + 4 * rowbytes));
  u32x4 r;
  r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
  r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
  return r;
}

constexpr int SMT = 32;             // rows of M per staged tile on the split path

template <bool HAS_MASK, bool HAS_SUB, int WTK, bool F16 = false>
__global__ void __launch_bounds__(THREADS, 2) wgrad_split_kernel(const WgradParams P) {
  constexpr int KB = 32 * WTK;                          // columns per k-block (160 / 128 / 96)
  constexpr int ZRB = 320;                              // bytes per row of a dZ term image (160 bf16)
  constexpr int XRB = WTK == 3 ? 192 : 320;             // X term image (WTK 4: 256 bytes of data + 64 of pad)
  constexpr int ZIMG = SMT * ZRB, XIMG = SMT * XRB;
  constexpr int TERMS = F16 ? 2 : 3;                    // three bf16 terms, or two f16 terms of the scaled operands (rr_wgrad_args.split = 2)
  __shared__ __attribute__((aligned(16))) unsigned char lds[TERMS * ZIMG + TERMS * XIMG];
  const rr_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float zs = 1.f, xsc = 1.f, izs = 1.f, ixs = 1.f;     // F16: operand scales from the caller's bounds (uniform) and their inverses
  if (F16) {
    const float bz = (a.dy_amax ? rr_amax_read(a.dy_amax) : 0.f) * (HAS_MASK ? fabsf(a.mask_scale) : 1.f);
    const int ez = rr_f16_exp(bz);
    const float bx = fmaxf((a.x1_amax ? rr_amax_read(a.x1_amax) : 0.f) + (a.x1_sub_amax ? rr_amax_read(a.x1_sub_amax) : 0.f),
                           a.x2_amax ? rr_amax_read(a.x2_amax) : 0.f);
    const int ex = rr_f16_exp(fmaxf(bx, 1.0f));         // (the ones column of the extended X)
    zs = rr_pow2(14 - ez); izs = rr_pow2(ez - 14);
    xsc = rr_pow2(14 - ex); ixs = rr_pow2(ex - 14);
    if (!(bz < 2.5e33f)) zs = __builtin_nanf("");       // an infinite / > 2^110 element: no scale fits, the gradient is NaN
    if (!(bx < 2.5e33f)) xsc = __builtin_nanf("");
  }
  const int nt = P.nblk_n * P.nblk_k;                   // XCD-aware mapping, see wgrad_fast_kernel
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % nt, chunk = (slot / nt) * 8 + xcd;
  if (chunk >= P.nchunks) return;
  const int bn = tile / P.nblk_k, bk = tile % P.nblk_k;
  const int nb = bn * WBN, kb = bk * KB;
  const int64_t mbeg = static_cast<int64_t>(chunk) * P.rows_per_chunk;
  int64_t mend = mbeg + P.rows_per_chunk;
  if (mend > a.M) mend = a.M;
  const int K = a.k1 + a.k2;
  const int nrows = static_cast<int>(mend - mbeg);
  const int ntiles = (nrows + SMT - 1) / SMT;
  const float* const zero = rr_zero_chunk;

  // ---- loader role: row r of the tile, chunks g + 8 i
  const int r = tid >> 3, g = tid & 7;
  const int xr = ((r >> 3) & 1) << 5;                   // column-block swap of this row in the images
  enum : int { X_NONE = 0, X_SEG1 = 1, X_SEG2 = 2, X_ONES = 3 };
  bool zok[5];
  int xcode[WTK];                                       // per chunk: column | kind << 16 | valid elements << 18 | (ones element + 1) << 21
  bool partial = false;
#pragma unroll
  for (int i = 0; i < 5; ++i) zok[i] = nb + 4 * (g + 8 * i) < a.N;
#pragma unroll
  for (int i = 0; i < WTK; ++i) {
    const int kx = kb + 4 * (g + 8 * i);                // extended column of the chunk
    int kind = X_NONE, col = 0, nv = 4, one = -1;
    if (kx < a.k1) {
      kind = X_SEG1; col = kx; nv = min(4, a.k1 - kx);
    } else if (kx >= P.k1p && kx < P.kext) {
      const int c2 = kx - P.k1p;
      if (c2 < a.k2) {
        kind = X_SEG2; col = c2; nv = min(4, a.k2 - c2);
        if (a.k2 - c2 < 4) one = a.k2 - c2;             // the ones column shares this chunk (k2 % 4 != 0)
      } else {
        kind = X_ONES; nv = 0; one = 0;                 // c2 == k2: the chunk is {1, 0, 0, 0}
      }
    }
    if (nv != 4 || one >= 0) partial = true;
    xcode[i] = col | (kind << 16) | (nv << 18) | ((one + 1) << 21);
  }
  partial = __any(partial);

  f32x4 zv[5], zm[5], xv[WTK], xs[WTK];
  bool m_ok = false;                                    // this thread's row of the tile in flight is inside the M-chunk
  int32_t ia = 0, is = 0;                               // gather indices of the NEXT tile's row
  auto fetch_idx = [&](int t) {                         // rows past the chunk read the last row's index (never used)
    int rr = t * SMT + r;
    if (rr > nrows - 1) rr = nrows - 1;
    ia = a.x1_idx ? ldgi(a.x1_idx + mbeg + rr) : 0;
    if (HAS_SUB) is = a.x1_sub_idx ? ldgi(a.x1_sub_idx + mbeg + rr) : 0;
  };
  auto issue = [&](int t) {                             // every load is issued; the address is what gets selected
    const int rr = t * SMT + r;
    m_ok = rr < nrows;
    const int64_t m = mbeg + (m_ok ? rr : 0);
    const float* zp = a.dy + m * a.ld_dy + nb + 4 * g;
    const float* mp = HAS_MASK ? a.mask + m * a.ld_mask + nb + 4 * g : zero;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      zv[i] = ldg4((m_ok && zok[i]) ? zp + 32 * i : zero);
      if (HAS_MASK) zm[i] = ldg4((m_ok && zok[i]) ? mp + 32 * i : zero);
    }
    const float* p1 = nullptr;
    const float* ps = nullptr;
    if (a.k1 > 0) {
      if (a.x1_idx) {
        if (ia >= 0) p1 = a.x1 + static_cast<int64_t>(ia) * a.ldx1;
      } else {
        p1 = a.x1 + m * a.ldx1;
      }
      if (HAS_SUB) {
        if (a.x1_sub_idx) {
          if (is >= 0) ps = a.x1_sub + static_cast<int64_t>(is) * a.ldx1_sub;
        } else {
          ps = a.x1_sub + m * a.ldx1_sub;
        }
      }
    }
    const float* p2 = a.k2 > 0 ? a.x2 + m * a.ldx2 : nullptr;
#pragma unroll
    for (int i = 0; i < WTK; ++i) {
      const int kind = (xcode[i] >> 16) & 3, col = xcode[i] & 0xffff;
      const float* src = kind == X_SEG1 ? p1 : (kind == X_SEG2 ? p2 : nullptr);
      xv[i] = ldg4((m_ok && src != nullptr) ? src + col : zero);
      if (HAS_SUB) xs[i] = ldg4((m_ok && ps != nullptr && kind == X_SEG1) ? ps + col : zero);
    }
  };
  auto put = [&](unsigned char* img, int imgbytes, int rowbytes, int chunk8, f32x4 v, float sc) {   // split + three 8-byte stores
    uint32_t a0, a1, a2, b0, b1, b2;
    if (F16) {
      split_pair_h(v.x, v.y, sc, a0, a1);
      split_pair_h(v.z, v.w, sc, b0, b1);
      unsigned char* d = img + r * rowbytes + ((chunk8 * 8) ^ xr);
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + imgbytes) = make_uint2(a1, b1);
      return;
    }
    split_pair(v.x, v.y, a0, a1, a2);
    split_pair(v.z, v.w, b0, b1, b2);
    unsigned char* d = img + r * rowbytes + ((chunk8 * 8) ^ xr);
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(d + imgbytes) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + 2 * imgbytes) = make_uint2(a2, b2);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      f32x4 z = zv[i];
      if (HAS_MASK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = zm[i][e] > 0.f ? zv[i][e] * a.mask_scale : 0.f;
      }
      put(lds, ZIMG, ZRB, g + 8 * i, z, zs);
    }
#pragma unroll
    for (int i = 0; i < WTK; ++i) {
      f32x4 x = xv[i];
      if (HAS_SUB) x = xv[i] - xs[i];
      if (partial) {                                    // k1 % 4 or k2 % 4 != 0, or the ones column: patch per element
        const int nv = (xcode[i] >> 18) & 7, one = ((xcode[i] >> 21) & 7) - 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = e < nv ? xv[i][e] : 0.f;
          if (HAS_SUB) u -= e < nv ? xs[i][e] : 0.f;
          if (e == one) u = m_ok ? 1.0f : 0.f;
          x[e] = u;
        }
      }
      put(lds + TERMS * ZIMG, XIMG, XRB, g + 8 * i, x, xsc);
    }
  };

  f32x4 acc[WT][WTK];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WTK; ++j) acc[i][j] = f32x4(0.f);

  const int wn = (wave >> 1) * (WT * 16), wk = (wave & 1) * (WTK * 16);
  const int fr = lane & 15, fq = lane >> 4;
  // transposed-read role: lane 4q+p of 16-lane group fq supplies row 8 fq + q, columns 4p..4p+3 of the 16-column block
  const int tq = fr >> 2, tp = fr & 3;
  const int trow = 8 * fq + tq;
  const int txr = (fq & 1) << 5;
  const unsigned char* const zbase = lds + trow * ZRB + 8 * tp;
  const unsigned char* const xbase = lds + TERMS * ZIMG + trow * XRB + 8 * tp;

  fetch_idx(0);
  issue(0);
  fetch_idx(1);
  for (int t = 0; t < ntiles; ++t) {
    commit();                                           // waits for the tile's loads; splits; writes the term images
    __syncthreads();
    if (t + 1 < ntiles) {
      issue(t + 1);
      fetch_idx(t + 2);
    }
    // The X terms of a group of k-tiles stay in registers across the five n-tiles (36 / 24 registers); holding all
    // WTK at once (what common-subexpression elimination makes of the plain double loop) spills next to the 100
    // accumulators and the next tile's chunks in flight.
    constexpr int JH = WTK <= 3 ? WTK : 3;
#pragma unroll
    for (int j0 = 0; j0 < WTK; j0 += JH) {
      asm volatile("" ::: "memory");                    // the second group RE-READS the dZ terms (no CSE across groups)
      if constexpr (F16) {
        f16x8 h0[JH], h1[JH];
#pragma unroll
        for (int jj = 0; jj < JH; ++jj) {
          if (j0 + jj < WTK) {
            const int xc = ((wk + 16 * (j0 + jj)) * 2) ^ txr;
            h0[jj] = as_f16x8(tr_read8(xbase + xc, XRB));
            h1[jj] = as_f16x8(tr_read8(xbase + XIMG + xc, XRB));
          }
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          const int zc = ((wn + 16 * i) * 2) ^ txr;
          const f16x8 a0 = as_f16x8(tr_read8(zbase + zc, ZRB));
          const f16x8 a1 = as_f16x8(tr_read8(zbase + ZIMG + zc, ZRB));
#pragma unroll
          for (int jj = 0; jj < JH; ++jj) {
            if (j0 + jj < WTK) {
              f32x4 c = acc[i][j0 + jj];
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, h0[jj], c, 0, 0, 0);   // smallest terms first
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, h1[jj], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, h0[jj], c, 0, 0, 0);
              acc[i][j0 + jj] = c;
            }
          }
        }
        continue;
      }
      bf16x8 b0[JH], b1[JH], b2[JH];
#pragma unroll
      for (int jj = 0; jj < JH; ++jj) {
        if (j0 + jj < WTK) {
          const int xc = ((wk + 16 * (j0 + jj)) * 2) ^ txr;
          b0[jj] = as_bf16x8(tr_read8(xbase + xc, XRB));
          b1[jj] = as_bf16x8(tr_read8(xbase + XIMG + xc, XRB));
          b2[jj] = as_bf16x8(tr_read8(xbase + 2 * XIMG + xc, XRB));
        }
      }
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        const int zc = ((wn + 16 * i) * 2) ^ txr;
        const bf16x8 a0 = as_bf16x8(tr_read8(zbase + zc, ZRB));
        const bf16x8 a1 = as_bf16x8(tr_read8(zbase + ZIMG + zc, ZRB));
        const bf16x8 a2 = as_bf16x8(tr_read8(zbase + 2 * ZIMG + zc, ZRB));
#pragma unroll
        for (int jj = 0; jj < JH; ++jj) {
          if (j0 + jj < WTK) {
            f32x4 c = acc[i][j0 + jj];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0[jj], c, 0, 0, 0);   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[jj], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2[jj], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0[jj], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1[jj], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[jj], c, 0, 0, 0);
            acc[i][j0 + jj] = c;
          }
        }
      }
    }
    __syncthreads();                                    // every wave is done with the images before the next commit
  }

  float* slab = static_cast<float*>(a.workspace) + static_cast<int64_t>(chunk) * P.slab;
  if (F16) {                                            // back from the scaled operands (two exact powers of two)
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WTK; ++j) acc[i][j] = (acc[i][j] * izs) * ixs;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int j = 0; j < WTK; ++j) {
      const int kx = kb + wk + j * 16 + fr;
      if (kx >= P.kext) continue;
      int kreal = -1;
      if (kx < a.k1) kreal = kx;
      else if (kx >= P.k1p && kx < P.k1p + a.k2) kreal = a.k1 + (kx - P.k1p);
      else if (kx == P.k1p + a.k2) kreal = -2;
      if (kreal == -1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nb + wn + i * 16 + fq * 4 + e;
        if (n >= a.N) continue;
        if (kreal >= 0) slab[static_cast<int64_t>(n) * K + kreal] = acc[i][j][e];
        else slab[static_cast<int64_t>(a.N) * K + n] = acc[i][j][e];
      }
    }
  }
}

// fixed-order sum of the chunk slabs into dw / dbias.  64 elements per workgroup; the chunk range is cut in four
// quarters (one per wave) of 8-deep independent loads - a thread walking all ~128 slabs alone keeps too few bytes in flight
// (32 us for 46 MB) - and the quarters are added in a fixed order through LDS: deterministic, no atomics.
__global__ void __launch_bounds__(THREADS) wgrad_reduce_kernel(const float* __restrict__ ws, int nchunks, int64_t slab,
                                                               int N, int K, float* __restrict__ dw, int64_t ld_dw,
                                                               float* __restrict__ dbias, int accumulate) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + el;
  const int per = (nchunks + 3) / 4;
  const int c0 = q * per;
  int c1 = c0 + per;
  if (c1 > nchunks) c1 = nchunks;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < slab) {
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += ws[static_cast<int64_t>(c + u) * slab + e];
    }
    for (; c < c1; ++c) s[0] += ws[static_cast<int64_t>(c) * slab + e];
  }
  part[q][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (q != 0 || e >= slab) return;
  const float t = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
  const int64_t nk = static_cast<int64_t>(N) * K;
  if (e < nk) {
    const int64_t n = e / K, k = e - n * K;
    float* d = dw + n * ld_dw + k;
    *d = accumulate ? *d + t : t;
  } else if (dbias) {
    float* d = dbias + (e - nk);
    *d = accumulate ? *d + t : t;
  }
}

int64_t wgrad_want_chunks(int64_t M, int N, int kext) {
  const int tiles = ((N + WBN - 1) / WBN) * ((kext + WBN - 1) / WBN);
  // One round of workgroups (2 fit per CU: a 513th would wait a whole round).  Chunk c runs on XCD c % 8
  // (XCD-aware mapping), so the chunk count is a multiple of 8: otherwise some XCDs get one more chunk than their
  // 64 slots hold.  Round 1 ran 384 here (the masked / subtracting loader held ~250 VGPRs and starved the dX chain on
  // the main stream); with the mask applied by the dX GEMM (dZ side output) the kernels are leaner and the full
  // 512 wins: step -1.1 % (same-box A/B: 448 -0.6 %, 320 +2.4 %), this kernel alone -20...-29 %.
#ifndef RR_WGRAD_WGS
#define RR_WGRAD_WGS 512
#endif
  int64_t want = RR_WGRAD_WGS / tiles;
  if (want >= 8) want -= want % 8;
  const int64_t maxc = (M + 63) / 64;
  if (want > maxc) want = maxc;
  if (want < 1) want = 1;
  return want;
}

void wgrad_plan(int64_t M, int N, int k1, int k2, WgradParams* P, int mt = WMT) {
  P->k1p = (k1 + 3) & ~3;                         // segment 2 (and the ones column) start 16-byte aligned
  P->kext = P->k1p + k2 + 1;
  P->nblk_n = (N + WBN - 1) / WBN;
  P->nblk_k = (P->kext + WBN - 1) / WBN;
  const int64_t want = wgrad_want_chunks(M, N, P->kext);
  int64_t rpc = (M + want - 1) / want;
  rpc = (rpc + mt - 1) / mt * mt;
  if (rpc < mt) rpc = mt;
  P->rows_per_chunk = rpc;
  P->nchunks = static_cast<int>((M + rpc - 1) / rpc);
  if (P->nchunks < 1) P->nchunks = 1;
  P->slab = static_cast<int64_t>(N) * (k1 + k2) + N;
}

template <int NT>
int launch_linear(const LinearParams& P, hipStream_t s, bool fast) {
  const rr_linear_args& a = P.a;
  dim3 grid(static_cast<unsigned>((a.M + BM - 1) / BM), static_cast<unsigned>((a.N + 16 * NT - 1) / (16 * NT)));
  if (fast) {
    if (a.a_mask) linear_fast_kernel<NT, 2><<<grid, THREADS, 0, s>>>(P);
    else if (a.a1_sub) linear_fast_kernel<NT, 1><<<grid, THREADS, 0, s>>>(P);
    else linear_fast_kernel<NT, 0><<<grid, THREADS, 0, s>>>(P);
  } else {
    if (a.a_mask) linear_kernel<NT, 2><<<grid, THREADS, 0, s>>>(P);
    else if (a.a1_sub) linear_kernel<NT, 1><<<grid, THREADS, 0, s>>>(P);
    else linear_kernel<NT, 0><<<grid, THREADS, 0, s>>>(P);
  }
  return rr_launch_status();
}

inline bool vec_ok(const float* p, int64_t ld) { return p && rr_aligned16(p) && (ld % 4 == 0); }

// C[m, n] = bias[n] + sum_k A[m, k] * W[n, k] for a handful of output columns (the FFN's last layer: N = task_num <= 8,
// models/base_model.py:57): 16 lanes per row, the row in registers, a shuffle tree per output.  The MFMA kernels spend a
// 64-column tile (and ~20 us of fixed cost at one row per molecule) on these one or two columns.
constexpr int RD_MAXK = 1024;                 // 16 lanes x 4 floats x 16 chunks
__global__ void __launch_bounds__(256) linear_rowdot_kernel(const float* __restrict__ a, int64_t lda, int k,
                                                            const float* __restrict__ w, int64_t ldw,
                                                            const float* __restrict__ bias, int64_t M, int N,
                                                            float* __restrict__ c, int64_t ldc) {
  const int l16 = threadIdx.x & 15;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
  const bool live = row < M;
  const float* ar = a + (live ? row : 0) * lda;
  f32x4 x[RD_MAXK / 64];
  const int nch = (k + 63) / 64;
#pragma unroll
  for (int i = 0; i < RD_MAXK / 64; ++i) {
    const int kk = i * 64 + l16 * 4;
    x[i] = (i < nch && kk < k) ? *reinterpret_cast<const f32x4*>(ar + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int n = 0; n < N; ++n) {
    const float* wr = w + static_cast<int64_t>(n) * ldw;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < RD_MAXK / 64; ++i) {
      const int kk = i * 64 + l16 * 4;
      if (i < nch && kk < k) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + kk);
        acc += x[i][0] * wv[0];
        acc += x[i][1] * wv[1];
        acc += x[i][2] * wv[2];
        acc += x[i][3] * wv[3];
      }
    }
    acc += __shfl_xor(acc, 8, 16);
    acc += __shfl_xor(acc, 4, 16);
    acc += __shfl_xor(acc, 2, 16);
    acc += __shfl_xor(acc, 1, 16);
    if (live && l16 == 0) c[row * ldc + n] = acc + (bias ? bias[n] : 0.f);
  }
}


// largest magnitude of a [rows, per_row (x 4 when VEC)] block, maxed into the magnitude slot `out` (non-negative floats order
// like their bit patterns; a NaN fails every comparison and is skipped)
template <bool VEC>
__global__ void __launch_bounds__(256) amax_kernel(const float* __restrict__ x, int64_t total, int per_row, int64_t ld, int tail,
                                                   float* __restrict__ out) {
  __shared__ float part[4];
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  const bool dense = VEC ? ld == 4 * static_cast<int64_t>(per_row) : ld == per_row;
  auto at = [&](int64_t e) -> int64_t { return dense ? e * (VEC ? 4 : 1) : (e / per_row) * ld + (e % per_row) * (VEC ? 4 : 1); };
  // VEC with cols % 4 != 0 (rows are 16-byte aligned, the last chunk of a row holds padding): its tail elements do not count
  auto ldv = [&](int64_t e) -> f32x4 {
    f32x4 v = ldg4(x + at(e));
    if (tail != 0 && (e % per_row) == per_row - 1) {
      if (tail < 2) v.y = 0.f;
      if (tail < 3) v.z = 0.f;
      v.w = 0.f;
    }
    return v;
  };
  int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; e + 3 * stride < total; e += 4 * stride) {      // four independent loads in flight per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (VEC) {
        const f32x4 v = ldv(e + u * stride);
        m[u] = fmaxf(fmaxf(m[u], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      } else {
        m[u] = fmaxf(m[u], fabsf(x[at(e + u * stride)]));
      }
    }
  }
  for (; e < total; e += stride) {
    if (VEC) {
      const f32x4 v = ldv(e);
      m[0] = fmaxf(fmaxf(m[0], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      m[0] = fmaxf(m[0], fabsf(x[at(e)]));
    }
  }
  float r = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    rr_amax_put(out, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
  }
}

// RR_EPI_MODE (A/B knob): 0 = accumulator-layout epilogue everywhere, 1 = row-contiguous where the epilogue READS (a
// residual), 2 = row-contiguous everywhere it applies.  Measured (profiles/r03_experiments.txt): with a residual read
// 223 -> 196 us per isolated 139k-row launch; store-only epilogues do not gain and pay the LDS round trip; inside a
// training step (kernels of three streams interleaved on the chip) the difference is within the noise.
template <int NTP, int NT, int MODE, int WAVES, int EPI, bool F16>
int launch_split_epi(const LinearParams& P, hipStream_t s) {
  // k-loop: two weight images + the bias slice; the 12-wave geometry's epilogue needs 12 transposition regions of
  // 8 x 77 float4, the column-sum / sign-bit staging and the bias slice (linear_split_kernel, "LDS layout")
  constexpr int panel2 = 2 * NT * (F16 ? 2 : 3) * 1024, bn4 = 16 * NT * 4;
  constexpr bool can_persist = MODE == 0 && WAVES == 12 && NT == NTP && EPI == 0;
  constexpr int smem = (WAVES == 12 ? ((12 * 8 * 77 * 16 > panel2 ? 12 * 8 * 77 * 16 : panel2) + 13 * bn4) : panel2 + bn4) +
                       (can_persist ? WAVES * 2048 : 0) + 16;  // + the persistent form's operand prefetch slots + the two magnitude words
  // > 64 KiB of LDS has to be asked for once per kernel AND per device (the attribute lives with the device's code
  // object); atomics because two host threads may launch the same instantiation at once (setting it twice is harmless)
  static std::atomic<uint64_t> configured{0};          // bit d: done on device d (devices >= 64 set it every launch)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RR_ERR_LAUNCH;
  if (dev < 0 || dev >= 64 || !((configured.load(std::memory_order_acquire) >> dev) & 1u)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_split_kernel<NTP, NT, MODE, WAVES, EPI, F16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return RR_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) configured.fetch_or(uint64_t(1) << dev, std::memory_order_release);
  }
  const int64_t nblk = (P.a.M + 16 * WAVES - 1) / (16 * WAVES);
  // Persistent form (see the kernel): plain-operand GEMMs of the one-workgroup-per-CU geometry with more row blocks than
  // CUs, an even number of k-steps (the pipeline's two slots / two image buffers keep their parity across the block
  // boundary) and interior steps on the lean loader.  RR_NO_PERSIST (A/B knob) keeps one workgroup per row block.
  if (MODE == 0 && WAVES == 12 && NT == NTP && EPI == 0) {
    static int n_cu[64] = {0};
    int cus = (dev >= 0 && dev < 64) ? n_cu[dev] : 0;
    if (cus == 0) {
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1 << 30;
      if (dev >= 0 && dev < 64) n_cu[dev] = cus;
    }
    const int nk = P.t1 + P.t2;
    const bool lean = P.a.k1 + SK <= RR_ZERO_ROW && P.a.k2 + SK <= RR_ZERO_ROW;
    if (nblk > cus && nk >= 2 && nk % 2 == 0 && lean && !getenv("RR_NO_PERSIST")) {
      LinearParams Q = P;
      Q.persist = 1;
      // every CU an equal share of the 64-row units (the granularity of the column-sum partials)
      const dim3 grid(static_cast<unsigned>(cus), 1);
      linear_split_kernel<NTP, NT, MODE, WAVES, EPI, F16><<<grid, 64 * WAVES, smem, s>>>(Q);
      return rr_launch_status();
    }
  }
  // (two column blocks: ids i, i + 8 of a 1-D grid are one row block's pair - see the kernel)
  const dim3 grid = NTP == 2 * NT ? dim3(static_cast<unsigned>((nblk + 7) / 8 * 16), 1)
                                  : dim3(static_cast<unsigned>(nblk), static_cast<unsigned>((NTP + NT - 1) / NT));
  linear_split_kernel<NTP, NT, MODE, WAVES, EPI, F16><<<grid, 64 * WAVES, smem, s>>>(P);
  return rr_launch_status();
}
template <int NTP, int NT, int MODE, int WAVES, bool F16>
int launch_split_one(const LinearParams& P, hipStream_t s) {
  if (WAVES == 12 && (MODE == 0 || MODE == 1)) {       // (the dX forms, MODE 2 / 3, never carry a residual)
    const bool rs = RR_EPI_MODE == 2 || (RR_EPI_MODE == 1 && P.a.residual != nullptr);
    // EPI 0 / 1: the epilogue (accumulator layout / row-contiguous) with the lean loader only; segments past the zero row
    // (K > 992: no configuration of the model) go to the twins EPI 2 / 3, which keep the generic loader
    const bool lean = P.a.k1 + SK <= RR_ZERO_ROW && P.a.k2 + SK <= RR_ZERO_ROW;
    constexpr int E1 = (WAVES == 12 && (MODE == 0 || MODE == 1)) ? 1 : 0, G = (WAVES == 12 && (MODE == 0 || MODE == 1)) ? 2 : 0;
    if (!lean) return rs ? launch_split_epi<NTP, NT, MODE, WAVES, G + E1, F16>(P, s) : launch_split_epi<NTP, NT, MODE, WAVES, G, F16>(P, s);
    if (rs) return launch_split_epi<NTP, NT, MODE, WAVES, E1, F16>(P, s);
  }
  return launch_split_epi<NTP, NT, MODE, WAVES, 0, F16>(P, s);
}
template <int NTP, int NT, int WAVES, bool F16 = false>
int launch_split(const LinearParams& P, hipStream_t s) {
  if (P.a.a_mask_bits) return launch_split_one<NTP, NT, 3, WAVES, F16>(P, s);
  if (P.a.a_mask) return launch_split_one<NTP, NT, 2, WAVES, F16>(P, s);
  if (P.a.a1_sub) return launch_split_one<NTP, NT, 1, WAVES, F16>(P, s);
  return launch_split_one<NTP, NT, 0, WAVES, F16>(P, s);
}
}  // namespace

extern "C" {

#ifdef RR_TRACE
int rr_debug_set_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rr_trace_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

int rr_linear_f32(const rr_linear_args* args, rr_stream_t stream) {
  RR_CHECK_ARG(args);
  const rr_linear_args& a = *args;
  RR_CHECK_ARG(a.M >= 0 && a.N >= 1 && a.k1 >= 0 && a.k2 >= 0 && a.k1 + a.k2 >= 1);
  RR_CHECK_ARG(a.w && a.c && a.ldc >= a.N);
  RR_CHECK_ARG(a.w_packed >= 0 && a.w_packed <= 3);
  RR_CHECK_ARG(a.w_packed >= 2 ? rr_aligned16(a.w)
                               : (a.w_packed ? (a.ldw == r16(a.k1) + r16(a.k2) && rr_aligned16(a.w)) : (a.ldw >= a.k1 + a.k2)));
  RR_CHECK_ARG(a.k1 == 0 || (a.a1 && a.lda1 >= a.k1));
  RR_CHECK_ARG(a.k2 == 0 || (a.a2 && a.lda2 >= a.k2));
  RR_CHECK_ARG(!a.a1_sub || (a.k1 > 0 && a.lda1_sub >= a.k1));
  RR_CHECK_ARG(!a.a_mask || (a.k2 == 0 && !a.a1_sub && !a.a1_idx && a.ld_mask >= a.k1));
  RR_CHECK_ARG(!a.residual || a.ldr >= a.N);
  RR_CHECK_ARG(!a.c_pre || a.ld_pre >= a.N);
  RR_CHECK_ARG(!a.dz_out || ((a.a_mask || a.a_mask_bits) && a.ld_dz >= a.k1));
  RR_CHECK_ARG(!a.a_mask_bits || (a.w_packed >= 2 && a.k2 == 0 && !a.a1_sub && !a.a1_idx && a.k1 % 4 == 0));
  RR_CHECK_ARG(!a.mask_bits_out || (a.w_packed >= 2 && a.N % 4 == 0));
  RR_CHECK_ARG(a.act == RR_ACT_NONE || a.act == RR_ACT_RELU);
  RR_CHECK_ARG((!a.c_amax_out && !a.dz_amax_out) || a.w_packed == 3);      // (the two-f16-term kernels' epilogues only)
  RR_CHECK_ARG(!a.dz_amax_out || a.dz_out);
  RR_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f);
  RR_CHECK_ARG(a.M < (int64_t(1) << 31) * BM);
  if (a.M == 0) return RR_OK;

  LinearParams P;
  P.a = a;
  P.t1 = (a.k1 + BK - 1) / BK;
  P.t2 = (a.k2 + BK - 1) / BK;
  P.flags = 0;
  P.persist = 0;
  if (a.k1 > 0 && vec_ok(a.a1, a.lda1)) P.flags |= F_A1_VEC;
  if (a.k2 > 0 && vec_ok(a.a2, a.lda2)) P.flags |= F_A2_VEC;
  if (a.a1_sub && vec_ok(a.a1_sub, a.lda1_sub)) P.flags |= F_SUB_VEC;
  if (a.a_mask && vec_ok(a.a_mask, a.ld_mask)) P.flags |= F_MASK_VEC;
  if (vec_ok(a.w, a.ldw)) {
    P.flags |= F_W1_VEC;
    if (a.k1 % 4 == 0) P.flags |= F_W2_VEC;
  }
  if (a.N % 4 == 0 && vec_ok(a.c, a.ldc) && (!a.bias || rr_aligned16(a.bias)) &&
      (!a.residual || vec_ok(a.residual, a.ldr)))
    P.flags |= F_EPI_VEC;
  if (a.c_pre && vec_ok(a.c_pre, a.ld_pre)) P.flags |= F_PRE_VEC;
  P.drop_thr = rr_drop_threshold(a.drop_p);
  P.keep_scale = 1.0f / (1.0f - a.drop_p);

  hipStream_t s = static_cast<hipStream_t>(stream);
  P.w_k1_off = a.w_packed ? r16(a.k1) : a.k1;         // (packed rows are r16(k1) + r16(k2) floats: with k2 = 0 this is the row pitch)
  // a handful of output columns off one plain operand, nothing fused (the FFN's last layer): the row-dot kernel
  if (a.N <= 8 && a.w_packed == 1 && a.k2 == 0 && a.k1 % 4 == 0 && a.k1 <= RD_MAXK && (P.flags & F_A1_VEC) && !a.a1_idx && !a.a1_sub &&
      !a.a_mask && !a.a_mask_bits && !a.residual && a.act == RR_ACT_NONE && a.drop_p == 0.f && !a.c_pre && !a.dz_out &&
      !a.colsum_partial && !a.mask_bits_out && !getenv("RR_NO_ROWDOT")) {
    linear_rowdot_kernel<<<static_cast<unsigned>((a.M + 15) / 16), 256, 0, s>>>(a.a1, a.lda1, a.k1, a.w, P.w_k1_off, a.bias, a.M,
                                                                               a.N, a.c, a.ldc);
    return rr_launch_status();
  }
  // fast path: packed W, every present A source 16-byte addressable, vector epilogue
  bool fast = a.w_packed && (P.flags & F_EPI_VEC) && (!a.c_pre || (P.flags & F_PRE_VEC));
  if (a.k1 > 0 && !(P.flags & F_A1_VEC)) fast = false;
  if (a.k2 > 0 && !(P.flags & F_A2_VEC)) fast = false;
  if (a.a1_sub && !(P.flags & F_SUB_VEC)) fast = false;
  if (a.a_mask && !(P.flags & F_MASK_VEC)) fast = false;
  // the generic kernel reads packed weights too: segment 2 simply starts at column r16(k1)
  P.w_k1_off = a.w_packed ? r16(a.k1) : a.k1;
  if (a.w_packed) P.flags |= F_W1_VEC | F_W2_VEC;
  if (a.dz_out) {                                     // side output only exists on the straight-line path
    if (!fast || a.k1 % 4 != 0 || !vec_ok(a.dz_out, a.ld_dz)) return RR_ERR_ALIGN;
  }
  if (a.colsum_partial) {                             // likewise the weighted column-sum side output
    RR_CHECK_ARG(a.colsum_w && a.ld_partial >= a.N);
    if (!fast || !vec_ok(a.colsum_partial, a.ld_partial)) return RR_ERR_ALIGN;
  }
  if (a.w_packed == 3) {                              // two f16 terms: the same geometries
    if (!fast || a.N > 608 || a.M >= (int64_t(1) << 31) * 128) return RR_ERR_ALIGN;
    if (a.dz_accumulate) return RR_ERR_UNSUPPORTED;
    RR_CHECK_ARG((a.k1 == 0 || a.a1_amax) && (a.k2 == 0 || a.a2_amax) && (!a.a1_sub || a.a1_sub_amax));
    P.t1 = r32(a.k1) / SK;
    P.t2 = r32(a.k2) / SK;
    if (a.N <= 64) return launch_split<4, 4, 8, true>(P, s);
    if (a.N <= 160) return launch_split<10, 10, 8, true>(P, s);
    if (a.N <= 304 && a.M <= 8192) return launch_split<19, 5, 8, true>(P, s);
    if (a.N <= 304) return launch_split<19, 19, 12, true>(P, s);
    return launch_split<38, 19, 12, true>(P, s);
  }
  if (a.w_packed == 2) {                              // split terms only exist in the straight-line geometry
    if (!fast || a.N > 608 || a.M >= (int64_t(1) << 31) * 128) return RR_ERR_ALIGN;
    if (a.dz_accumulate) return RR_ERR_UNSUPPORTED;
    P.t1 = r32(a.k1) / SK;
    P.t2 = r32(a.k2) / SK;
    if (a.N <= 64) return launch_split<4, 4, 8>(P, s);
    if (a.N <= 160) return launch_split<10, 10, 8>(P, s);
    // few rows (the distinct reactants of a shared-prefix step: ~2 k bonds): 192-row workgroups would leave most CUs idle,
    // so the 19 column tiles are cut into blocks of 5 (5 + 5 + 5 + 4) as well - same weight image, same k order
    if (a.N <= 304 && a.M <= 8192) return launch_split<19, 5, 8>(P, s);
    if (a.N <= 304) return launch_split<19, 19, 12>(P, s);
    return launch_split<38, 19, 12>(P, s);               // two column blocks of 19 tiles (H = 600)
  }
  // Few rows (the FFN head: one row per molecule, 64 row blocks): cut the columns into 64-wide blocks as well, so the
  // launch covers the chip (5 x 64 workgroups at N = 300 instead of 64) - each element's k-order, hence its value, is
  // the same in every geometry.
  if (a.N <= 64 || a.M <= 8192) return launch_linear<4>(P, s, fast);
  if (a.N <= 160) return launch_linear<10>(P, s, fast);
  return launch_linear<19>(P, s, fast);
}

int rr_pack_weight_f32(const float* src, int64_t ld_src, int transpose, int rows, int c0, int k1, int k2, float* dst,
                       rr_stream_t stream) {
  RR_CHECK_ARG(src && dst && rows >= 1 && c0 >= 0 && k1 >= 0 && k2 >= 0 && k1 + k2 >= 1 && ld_src >= 1);
  const int64_t total = static_cast<int64_t>(rows) * (r16(k1) + r16(k2));
  pack_weight_kernel<<<rr_grid_for(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(src, ld_src, transpose, rows,
                                                                                           c0, k1, k2, dst);
  return rr_launch_status();
}

int rr_pack_weights_f32(const rr_pack_desc* descs, int n, rr_stream_t stream) {
  RR_CHECK_ARG(descs && n >= 0 && n <= RR_MAX_PACK);
  if (n == 0) return RR_OK;
  PackMany P;
  int64_t biggest = 0, biggest_split = 0;
  for (int i = 0; i < n; ++i) {
    const rr_pack_desc& q = descs[i];
    RR_CHECK_ARG(q.src && q.dst && q.rows >= 1 && q.c0 >= 0 && q.k1 >= 0 && q.k2 >= 0 && q.k1 + q.k2 >= 1 && q.ld_src >= 1);
    RR_CHECK_ARG(q.split == 0 || ((q.split == 1 || q.split == 2) && q.rows <= 608 && rr_aligned16(q.dst)));
    P.d[i] = q;
    const int64_t total = q.split ? static_cast<int64_t>((r32(q.k1) + r32(q.k2)) / SK) * split_nt(q.rows) * 512
                                  : static_cast<int64_t>(q.rows) * (r16(q.k1) + r16(q.k2));
    if (total > (q.split ? biggest_split : biggest)) (q.split ? biggest_split : biggest) = total;
  }
  for (int i = n; i < RR_MAX_PACK; ++i) P.d[i] = descs[0];
  if (biggest_split > 0) {                             // split weights present: ONE launch packs both layouts
    bool any_f16 = false;
    for (int i = 0; i < n; ++i) any_f16 = any_f16 || descs[i].split == 2;
    if (any_f16) pack_scale_kernel<<<dim3(PACK_SCALE_BLOCKS, static_cast<unsigned>(n)), 1024, 0, static_cast<hipStream_t>(stream)>>>(P);
    const int64_t work = biggest_split > biggest ? biggest_split : biggest;
    dim3 grid(static_cast<unsigned>(rr_grid_for(work, 256, 64)), static_cast<unsigned>(n));
    pack_split_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(P);
  } else if (biggest > 0) {
    dim3 grid(static_cast<unsigned>(rr_grid_for(biggest, 256, 64)), static_cast<unsigned>(n));
    pack_weights_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(P);
  }
  return rr_launch_status();
}

int rr_amax_f32(const float* x, int64_t rows, int cols, int64_t ld, float* amax, rr_stream_t stream) {
  RR_CHECK_ARG(x && amax && rows >= 0 && cols >= 1 && ld >= cols);
  if (rows == 0) return RR_OK;
  const bool vec = vec_ok(x, ld) && (cols + 3) / 4 * 4 <= ld;
  const int64_t per_row = vec ? (cols + 3) / 4 : cols;
  const int64_t total = rows * per_row;
  const unsigned grid = static_cast<unsigned>(rr_grid_for((total + 3) / 4, 256, 1024));
  if (vec) amax_kernel<true><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(x, total, static_cast<int>(per_row), ld, cols % 4, amax);
  else amax_kernel<false><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(x, total, static_cast<int>(per_row), ld, 0, amax);
  return rr_launch_status();
}

int64_t rr_packed_weight_ld(int k1, int k2) { return r16(k1) + r16(k2); }

int64_t rr_mask_bits_row_bytes(int N) { return N < 1 ? 0 : mask_bits_row(N); }

size_t rr_split_weight_bytes(int rows, int k1, int k2) {
  if (rows < 1 || rows > 608 || k1 < 0 || k2 < 0 || k1 + k2 < 1) return 0;
  return static_cast<size_t>((r32(k1) + r32(k2)) / SK) * split_nt(rows) * 3 * 1024;
}

int64_t rr_linear_colsum_rows(int64_t M) { return M <= 0 ? 0 : (M + BM - 1) / BM; }

size_t rr_linear_wgrad_workspace_bytes(int64_t M, int N, int K) {
  if (M < 0 || N < 1 || K < 1) return 0;
  // upper bound over every [k1|k2] split of K: the chunk count is largest for the narrowest extended K
  const int64_t nc = wgrad_want_chunks(M, N, K + 1);
  return static_cast<size_t>(nc) * (static_cast<size_t>(N) * K + N) * sizeof(float);
}

int rr_linear_wgrad_f32(const rr_wgrad_args* args, rr_stream_t stream) {
  RR_CHECK_ARG(args);
  const rr_wgrad_args& a = *args;
  RR_CHECK_ARG(a.M >= 0 && a.N >= 1 && a.k1 >= 0 && a.k2 >= 0 && a.k1 + a.k2 >= 1);
  RR_CHECK_ARG(a.dy && a.dw && a.workspace && a.ld_dy >= a.N && a.ld_dw >= a.k1 + a.k2);
  RR_CHECK_ARG(a.k1 == 0 || (a.x1 && a.ldx1 >= a.k1));
  RR_CHECK_ARG(a.k2 == 0 || (a.x2 && a.ldx2 >= a.k2));
  RR_CHECK_ARG(!a.x1_sub || (a.k1 > 0 && a.ldx1_sub >= a.k1));
  RR_CHECK_ARG(!a.mask || a.ld_mask >= a.N);
  const int K = a.k1 + a.k2;
  WgradParams P;
  P.a = a;
  RR_CHECK_ARG(a.split >= 0 && a.split <= 2);
  RR_CHECK_ARG(a.split != 2 || (a.dy_amax && (a.k1 == 0 || a.x1_amax) && (a.k2 == 0 || a.x2_amax) && (!a.x1_sub || a.x1_sub_amax)));
  wgrad_plan(a.M, a.N, a.k1, a.k2, &P, a.split ? SMT : WMT);
  if (a.workspace_bytes < static_cast<size_t>(P.nchunks) * static_cast<size_t>(P.slab) * sizeof(float))
    return RR_ERR_WORKSPACE;
  P.flags = 0;
  if (a.k1 > 0 && vec_ok(a.x1, a.ldx1)) P.flags |= F_A1_VEC;
  if (a.k2 > 0 && vec_ok(a.x2, a.ldx2)) P.flags |= F_A2_VEC;
  if (a.x1_sub && vec_ok(a.x1_sub, a.ldx1_sub)) P.flags |= F_SUB_VEC;
  if (a.mask && vec_ok(a.mask, a.ld_mask)) P.flags |= F_MASK_VEC;
  if (vec_ok(a.dy, a.ld_dy)) P.flags |= F_EPI_VEC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(static_cast<unsigned>(P.nblk_n * P.nblk_k * ((P.nchunks + 7) / 8) * 8));
  bool fast = (P.flags & F_EPI_VEC) != 0 && (a.N % 4 == 0) && a.M < (int64_t(1) << 31);
  const int64_t max_pitch = int64_t(1) << 25;          // 16 rows * pitch * 4 bytes must fit the kernel's 32-bit pointer steps
  if (a.ld_dy >= max_pitch || a.ld_mask >= max_pitch || a.ldx1 >= max_pitch || a.ldx1_sub >= max_pitch || a.ldx2 >= max_pitch)
    fast = false;
  if (a.mask && !(P.flags & F_MASK_VEC)) fast = false;
  if (a.k1 > 0 && !(P.flags & F_A1_VEC)) fast = false;
  if (a.k2 > 0 && !(P.flags & F_A2_VEC)) fast = false;
  if (a.x1_sub && !(P.flags & F_SUB_VEC)) fast = false;
  if (fast) {
    // narrowest k-block (96 / 128 / 160 columns) that still covers kext with nblk_k blocks
    const int per_blk = (P.kext + P.nblk_k - 1) / P.nblk_k;
    const int wtk = per_blk <= 96 ? 3 : (per_blk <= 128 ? 4 : 5);
    if (a.split) {                                      // (a request: the scalar-load geometry below stays on f32)
#define RR_WSPLIT_LAUNCH(MASK, SUB, F16)                                                   \
    do {                                                                                   \
      if (wtk == 3) wgrad_split_kernel<MASK, SUB, 3, F16><<<grid, THREADS, 0, s>>>(P);     \
      else if (wtk == 4) wgrad_split_kernel<MASK, SUB, 4, F16><<<grid, THREADS, 0, s>>>(P); \
      else wgrad_split_kernel<MASK, SUB, 5, F16><<<grid, THREADS, 0, s>>>(P);              \
    } while (0)
      if (a.split == 2) {
        if (a.mask && a.x1_sub) RR_WSPLIT_LAUNCH(true, true, true);
        else if (a.mask) RR_WSPLIT_LAUNCH(true, false, true);
        else if (a.x1_sub) RR_WSPLIT_LAUNCH(false, true, true);
        else RR_WSPLIT_LAUNCH(false, false, true);
      } else {
        if (a.mask && a.x1_sub) RR_WSPLIT_LAUNCH(true, true, false);
        else if (a.mask) RR_WSPLIT_LAUNCH(true, false, false);
        else if (a.x1_sub) RR_WSPLIT_LAUNCH(false, true, false);
        else RR_WSPLIT_LAUNCH(false, false, false);
      }
#undef RR_WSPLIT_LAUNCH
    } else {
#define RR_WGRAD_LAUNCH(MASK, SUB)                                                         \
    do {                                                                                   \
      if (wtk == 3) wgrad_fast_kernel<MASK, SUB, 3><<<grid, THREADS, 0, s>>>(P);           \
      else if (wtk == 4) wgrad_fast_kernel<MASK, SUB, 4><<<grid, THREADS, 0, s>>>(P);      \
      else wgrad_fast_kernel<MASK, SUB, 5><<<grid, THREADS, 0, s>>>(P);                    \
    } while (0)
    if (a.mask && a.x1_sub) RR_WGRAD_LAUNCH(true, true);
    else if (a.mask) RR_WGRAD_LAUNCH(true, false);
    else if (a.x1_sub) RR_WGRAD_LAUNCH(false, true);
    else RR_WGRAD_LAUNCH(false, false);
#undef RR_WGRAD_LAUNCH
    }
  } else {
    wgrad_kernel<<<grid, THREADS, 0, s>>>(P);
  }
  const int64_t total = P.slab;
  wgrad_reduce_kernel<<<static_cast<unsigned>((total + 63) / 64), THREADS, 0, s>>>(
      static_cast<const float*>(a.workspace), P.nchunks, P.slab, a.N, K, a.dw, a.ld_dw, a.dbias, a.accumulate);
  return rr_launch_status();
}

}  // extern "C"
