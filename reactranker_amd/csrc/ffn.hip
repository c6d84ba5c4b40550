// The FFN head of the scorer (reference models/base_model.py:32-60: Dropout, then ffn_depth x [Linear, ReLU, Dropout], the
// last Linear bare) and its input-gradient chain as ONE launch each: rr_ffn_chain_f32.
//
// Why: at one row per molecule (4,096 rows per 64 x 64 step) every layer of the head is a ~20 us launch of which ~5 us is
// matrix work, and the layers depend on each other: forward 3 launches, backward 3, ~12 launch boundaries of 6-7 us with the
// chip idle - ~140 us of a 5.2 ms training step, ~60 us of a 1.6 ms evaluation step (profiles/r04_bf16x3_step_timeline.txt).
// Here a workgroup owns 16 rows (one MFMA row tile) and walks ALL layers: the activations stay in LDS between layers, every
// wave owns a few 16-column tiles of the current layer and streams its weight fragments L2 -> registers directly (no other
// wave needs them; a weight element is used once per workgroup), three k-tiles ahead.  256 workgroups of 8 waves at 4,096
// rows: one per CU.
//
// Same numbers as the per-layer launches, bit for bit: the products of an output element enter v_mfma_f32_16x16x4_f32 in the
// order linear_fast_kernel / linear_kernel feed them (k-tiles of 16 ascending; inside a tile instruction j takes
// k = 4 q + j, q = 0..3), bias / ReLU / dropout / mask are the same operations on the same values, and the last layer of the
// forward chain is linear_rowdot_kernel's 16-lane dot product with its shuffle tree.  tests/test_gpu_ffn.py holds it to
// torch.equal against the per-layer path.
#include "rr_common.h"
#include <type_traits>

namespace {

constexpr int FROWS = 16;                       // rows per workgroup (one MFMA row tile)
constexpr int FPF = 3;                          // k-tiles of weight fragments in flight per wave

__host__ __device__ constexpr int f_r16(int k) { return (k + 15) & ~15; }
// LDS row pitch (floats): == 4 (mod 32), so the 8 lanes of a ds_read_b128 phase (8 rows, same k) hit 8 different 4-bank groups
__host__ __device__ constexpr int f_pitch(int kmax) { return ((kmax + 31) & ~31) + 4; }

typedef const __attribute__((address_space(1))) f32x4* f_gptr4;
__device__ __forceinline__ f32x4 f_ldg4(const float* p) { return *(f_gptr4)(p); }

struct FfnParams {
  rr_ffn_chain_args a;
  int pitch;                                    // LDS row pitch of the two activation buffers
  int x_vec;                                    // the input rows can be read in 16-byte chunks
  uint32_t drop_thr;
  float keep_scale;
};

// NTW: column tiles per wave (wave w owns tiles w, w + NW, ...); NW: waves per workgroup
template <int NW, int NTW>
__global__ void __launch_bounds__(64 * NW) ffn_chain_kernel(const FfnParams P) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const rr_ffn_chain_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fkq = lane >> 4;
  const int pitch = P.pitch;
  float* cur = fsm;                             // [FROWS][pitch] input of the stage being computed
  float* nxt = fsm + FROWS * pitch;             // ... its output = the next stage's input
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * FROWS;

  // ---- stage 0 input: rows of x, columns >= n_in read as zero up to the k-tile boundary
  {
    const int kin = a.stage[0].n_in, kp = f_r16(kin);
    const int per_row = kp / 4;
    for (int c = tid; c < FROWS * per_row; c += 64 * NW) {
      const int r = c / per_row, k = (c - r * per_row) * 4;
      const int64_t m = m0 + r;
      f32x4 v = f32x4(0.f);
      if (m < a.M && k < kin) {
        const float* xp = a.x + m * a.ldx + k;
        if (P.x_vec) {                                                // rows 16-byte addressable
          v = f_ldg4(xp);
          if (k + 1 >= kin) v.y = 0.f;
          if (k + 2 >= kin) v.z = 0.f;
          if (k + 3 >= kin) v.w = 0.f;
        } else {                                                      // e.g. the [M, 1] gradient of the scores
          v.x = xp[0];
          if (k + 1 < kin) v.y = xp[1];
          if (k + 2 < kin) v.z = xp[2];
          if (k + 3 < kin) v.w = xp[3];
        }
      }
      *reinterpret_cast<f32x4*>(cur + r * pitch + k) = v;
    }
  }
  __syncthreads();

  const int64_t m = m0 + fr;                     // this lane's row in every MFMA stage
  const bool row_ok = m < a.M;
  for (int s = 0; s < a.n_stages; ++s) {
    const rr_ffn_stage& S = a.stage[s];
    if (S.rowdot) {
      // ---- linear_rowdot_kernel's arithmetic on the staged rows: 16 lanes per row, lane l16 owns the chunks k = 64 i + 4 l16
      if (tid < 256) {
        const int l16 = tid & 15, r = tid >> 4;
        const int64_t mr = m0 + r;
        const float* xr = cur + r * pitch;
        const int k = S.n_in, nch = (k + 63) / 64;
        for (int n = 0; n < S.n_out; ++n) {
          const float* wr = S.w + static_cast<int64_t>(n) * S.ldw;
          float acc = 0.f;
          for (int i = 0; i < nch; ++i) {
            const int kk = i * 64 + l16 * 4;
            if (kk < k) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + kk);
              const f32x4 wv = f_ldg4(wr + kk);
              acc += xv[0] * wv[0];
              acc += xv[1] * wv[1];
              acc += xv[2] * wv[2];
              acc += xv[3] * wv[3];
            }
          }
          acc += __shfl_xor(acc, 8, 16);
          acc += __shfl_xor(acc, 4, 16);
          acc += __shfl_xor(acc, 2, 16);
          acc += __shfl_xor(acc, 1, 16);
          if (mr < a.M && l16 == 0) S.out[mr * S.ld_out + n] = acc + (S.bias ? S.bias[n] : 0.f);
        }
      }
      continue;                                  // (always the last stage: checked on the host)
    }
    const int N = S.n_out, K = S.n_in;
    const int T = (N + 15) / 16, nk = f_r16(K) / 16;
    // this wave's tiles: w, w + NW, ... below T (uniform count ntw: the first T % NW waves own one more than the others)
    int ntw = 0;
#pragma unroll
    for (int g = 0; g < NTW; ++g) ntw += (wave + g * NW < T) ? 1 : 0;
    f32x4 acc[NTW];
#pragma unroll
    for (int g = 0; g < NTW; ++g) acc[g] = f32x4(0.f);
    const float* arow = cur + fr * pitch + 4 * fkq;
    // The k-loop, compiled once per tile count NA so that it is straight-line code: weight fragments FPF k-tiles ahead in
    // registers, every load UNCONDITIONAL (a k-tile index past the end is clamped: a harmless re-read of the last one) - a load
    // under a branch, even a uniform one, is waited for with vmcnt(0) at the join, which puts a full L2 round trip in front of
    // every k-tile's MFMAs (the first version: 35 us per launch).
    auto gemm = [&](auto na_c) {
      constexpr int NA = decltype(na_c)::value;
      const float* wp[NA];
#pragma unroll
      for (int g = 0; g < NA; ++g) {
        int n = (wave + g * NW) * 16 + fr;
        n = n < N ? n : N - 1;                   // (the packed weight has exactly N rows)
        wp[g] = S.w + static_cast<int64_t>(n) * S.ldw + 4 * fkq;
      }
      f32x4 wf[FPF][NA];
#pragma unroll
      for (int p = 0; p < FPF; ++p) {
        const int kl = p < nk ? p : nk - 1;
#pragma unroll
        for (int g = 0; g < NA; ++g) wf[p][g] = f_ldg4(wp[g] + 16 * kl);
      }
      auto ktile = [&](int kt, const f32x4 (&w_now)[NA]) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(arow + 16 * kt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_now[g][j], af[j], acc[g], 0, 0, 0);
        }
      };
      int kt0 = 0;
      for (; kt0 + FPF <= nk; kt0 += FPF) {      // whole groups of FPF k-tiles
#pragma unroll
        for (int p = 0; p < FPF; ++p) {
          const int kt = kt0 + p;
          f32x4 w_now[NA];
#pragma unroll
          for (int g = 0; g < NA; ++g) w_now[g] = wf[p][g];
          const int kl = kt + FPF < nk ? kt + FPF : nk - 1;
#pragma unroll
          for (int g = 0; g < NA; ++g) wf[p][g] = f_ldg4(wp[g] + 16 * kl);
          asm volatile("" ::: "memory");         // the refill is ISSUED here, FPF k-tiles ahead of its use (left alone the
          ktile(kt, w_now);                      // scheduler sinks it to just in front of that use to save registers)
        }
      }
#pragma unroll
      for (int p = 0; p < FPF - 1; ++p) {        // the last nk % FPF k-tiles: their fragments are in flight or here, no loads left
        if (kt0 + p < nk) {
          f32x4 w_now[NA];
#pragma unroll
          for (int g = 0; g < NA; ++g) w_now[g] = wf[p][g];
          ktile(kt0 + p, w_now);
        }
      }
    };
    if (ntw == NTW) gemm(std::integral_constant<int, NTW>{});
    else if (NTW >= 2 && ntw == NTW - 1) gemm(std::integral_constant<int, (NTW >= 2 ? NTW - 1 : 1)>{});
    else if (NTW >= 3 && ntw == NTW - 2) gemm(std::integral_constant<int, (NTW >= 3 ? NTW - 2 : 1)>{});
    else if (NTW >= 4 && ntw == NTW - 3) gemm(std::integral_constant<int, (NTW >= 4 ? NTW - 3 : 1)>{});
    else if (NTW >= 5 && ntw == NTW - 4) gemm(std::integral_constant<int, (NTW >= 5 ? NTW - 4 : 1)>{});
    // ---- epilogue: a lane holds columns n .. n+3 (n = 16 t + 4 fkq) of row fr per tile
    const bool last = s + 1 == a.n_stages;
    const int kp_next = last ? 0 : f_r16(a.stage[s + 1].n_in);   // the next stage reads [0, kp_next) of its input rows
    const bool relu = S.relu != 0;
    const float* yrow = (S.post_mask != nullptr && row_ok) ? S.post_mask + m * S.ld_mask : nullptr;
#pragma unroll
    for (int g = 0; g < NTW; ++g) {
      const int t = wave + g * NW;
      if (t >= T) continue;                      // (uniform)
      const int n = t * 16 + 4 * fkq;
      f32x4 v = acc[g];
      if (S.bias != nullptr && n < N) v = v + f_ldg4(S.bias + n);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (S.dropout && P.drop_thr != 0u) {
        const uint64_t base = static_cast<uint64_t>(m) * static_cast<uint64_t>(N) + static_cast<uint64_t>(n);
        const uint32_t w = rr_hash_group(S.drop_seed, base >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rr_hash_lane(w, e) >= P.drop_thr ? v[e] * P.keep_scale : 0.f;
      }
      if (n >= N) v = f32x4(0.f);                // columns past the layer's width: zeros for the next stage's k padding
      if (S.out != nullptr && row_ok && n < N) *reinterpret_cast<f32x4*>(S.out + m * S.ld_out + n) = v;
      if (!last) {
        if (S.post_mask != nullptr) {            // the next stage's operand: ReLU / dropout backward of the layer below
          f32x4 y = f32x4(0.f);
          if (yrow != nullptr && n < N) y = f_ldg4(yrow + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = y[e] > 0.f ? v[e] * a.mask_scale : 0.f;
        }
        if (n < kp_next) *reinterpret_cast<f32x4*>(nxt + fr * pitch + n) = v;
      }
    }
    if (!last) {
      // columns [16 T, kp_next) exist only when the next stage reads FEWER columns than this one wrote - nothing to zero:
      // kp_next <= 16 T always (the next stage's n_in <= this stage's n_out, checked on the host)
      __syncthreads();
      float* t2 = cur;
      cur = nxt;
      nxt = t2;
    }
  }
}

template <int NW, int NTW>
int launch_chain(const FfnParams& P, size_t lds, hipStream_t s) {
  static bool configured[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RR_ERR_LAUNCH;
  if (lds > 65536 && (dev < 0 || dev >= 64 || !configured[dev])) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_chain_kernel<NW, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return RR_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  const unsigned grid = static_cast<unsigned>((P.a.M + FROWS - 1) / FROWS);
  ffn_chain_kernel<NW, NTW><<<grid, 64 * NW, lds, s>>>(P);
  return rr_launch_status();
}

inline bool f_vec_ok(const float* p, int64_t ld) { return p && rr_aligned16(p) && (ld % 4 == 0); }

}  // namespace

extern "C" {

size_t rr_abi_ffn_chain_size(void) { return sizeof(rr_ffn_chain_args); }

// RR_ERR_UNSUPPORTED = a shape this kernel does not take: the caller issues the per-layer entry points instead
int rr_ffn_chain_f32(const rr_ffn_chain_args* args, rr_stream_t stream) {
  RR_CHECK_ARG(args);
  const rr_ffn_chain_args& a = *args;
  RR_CHECK_ARG(a.M >= 0 && a.n_stages >= 1 && a.n_stages <= RR_MAX_FFN && a.x && a.drop_p >= 0.f && a.drop_p < 1.f);
  if (a.M == 0) return RR_OK;
  RR_CHECK_ARG(a.ldx >= a.stage[0].n_in);
  int kmax = 0, tmax = 0;
  for (int s = 0; s < a.n_stages; ++s) {
    const rr_ffn_stage& S = a.stage[s];
    RR_CHECK_ARG(S.w && S.n_out >= 1 && S.n_in >= 1 && S.ldw >= S.n_in);
    RR_CHECK_ARG(s == 0 || S.n_in <= a.stage[s - 1].n_out);                  // a stage reads a prefix of what the one before wrote
    if (!rr_aligned16(S.w) || S.ldw % 4 != 0) return RR_ERR_UNSUPPORTED;
    if (S.n_in > kmax) kmax = S.n_in;
    if (S.rowdot) {
      RR_CHECK_ARG(s + 1 == a.n_stages && S.out && S.ld_out >= S.n_out);
      if (S.n_out > 8 || S.n_in % 4 != 0 || S.relu || S.dropout || S.post_mask) return RR_ERR_UNSUPPORTED;
      continue;
    }
    if (S.ldw != f_r16(S.n_in)) return RR_ERR_UNSUPPORTED;                    // packed, zero-padded rows (rr_pack_weights_f32)
    if (S.n_out % 4 != 0) return RR_ERR_UNSUPPORTED;
    if (S.bias && !rr_aligned16(S.bias)) return RR_ERR_UNSUPPORTED;
    if (S.out && (!f_vec_ok(S.out, S.ld_out) || S.ld_out < S.n_out)) return RR_ERR_UNSUPPORTED;
    if (S.post_mask && (!f_vec_ok(S.post_mask, S.ld_mask) || S.ld_mask < S.n_out)) return RR_ERR_UNSUPPORTED;
    RR_CHECK_ARG(s + 1 < a.n_stages || S.out);                               // the last stage must write somewhere
    const int T = (S.n_out + 15) / 16;
    if (T > tmax) tmax = T;
    if (f_r16(S.n_out) > kmax) kmax = f_r16(S.n_out);
  }
  if (kmax > 1024 || tmax > 40) return RR_ERR_UNSUPPORTED;
  FfnParams P;
  P.a = a;
  P.pitch = f_pitch(f_r16(kmax));
  P.x_vec = f_vec_ok(a.x, a.ldx) ? 1 : 0;
  P.drop_thr = rr_drop_threshold(a.drop_p);
  P.keep_scale = 1.0f / (1.0f - a.drop_p);
  const size_t lds = static_cast<size_t>(2) * FROWS * P.pitch * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tmax <= 8) return launch_chain<8, 1>(P, lds, s);
  if (tmax <= 24) return launch_chain<8, 3>(P, lds, s);
  return launch_chain<8, 5>(P, lds, s);
}

}  // extern "C"
