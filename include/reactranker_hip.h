/*
 * reactranker_hip.h — C-ABI of the MI355X-native ReactRanker hot path (gfx950 / CDNA4).
 *
 * The reference (IannLiu/ReactRanker) has no FFI layer: its boundary is Python call
 * signatures over stock ATen ops.  Each entry point below replaces one ATen op *site* of
 * the reference's D-MPNN encoder / ranking-loss path (cited per function, paths relative
 * to the reference root).  The host-side mirror (the reactranker_amd package) binds these with
 * ctypes and re-exposes the reference's own names (build_model, MPN, MPNDiff, FFN,
 * MLEloss, ListnetLoss, evidential_ranking, index_select_ND, LogCumsumExp, BatchMolGraph).
 *
 * Conventions
 *   - plain pointers + sizes only; every `float*`/`int32_t*` is DEVICE memory unless the
 *     parameter is documented as host; pointers are borrowed, never retained.
 *   - all floating point is fp32; all indices are int32 (the reference uses int64).
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised.
 *   - return value: RR_OK (0) or a negative rr_status; nothing is thrown; re-entrant across streams.
 *     rr_strerror() names a status.  State the library keeps between calls, all of it created lazily and
 *     mutex-guarded: the two non-blocking HIP streams per device that the step plans run their side chains on
 *     (csrc/plan.hip), the one-time shared-memory attribute of the large-LDS kernels, and the RCCL entry points
 *     resolved by the first rr_comm_* call.  No results, tensors or workspaces are retained.
 *   - a row index < 0 in any gather table means "skip" (contributes zero); the reference's
 *     padding index 0 is an ordinary row (row 0 = the padding row of BatchMolGraph,
 *     features/featurization.py:255-264) and is gathered like any other.
 *   - reductions are deterministic: sums run in a fixed order without float atomics; the only atomics are integer
 *     maxima of non-negative floats' bit patterns (the magnitude slots of the two-f16-term GEMMs, rr_amax_f32), which are
 *     exact and order-independent.  Results are run-to-run bit-identical.
 */
#ifndef REACTRANKER_HIP_H
#define REACTRANKER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rr_stream_t;

typedef enum rr_status {
  RR_OK = 0,
  RR_ERR_ARG = -1,          /* null pointer / negative size / inconsistent shapes */
  RR_ERR_ALIGN = -2,        /* pointer or leading dimension not aligned as required */
  RR_ERR_LAUNCH = -3,       /* hipLaunch / hipGetLastError failed */
  RR_ERR_UNSUPPORTED = -4,  /* size outside the supported range (documented per call) */
  RR_ERR_WORKSPACE = -5     /* workspace too small */
} rr_status;

const char* rr_strerror(int status);
int rr_version(void);               /* ABI version, bumped on any signature change */
/* sizeof(rr_linear_args) / sizeof(rr_wgrad_args) as compiled, so a binding can verify its struct layout. */
void rr_abi_struct_sizes(size_t* linear_args, size_t* wgrad_args);
size_t rr_abi_gather_epi_size(void);   /* sizeof(rr_gather_epi) as compiled */

/* ------------------------------------------------------------------ gathers -------- */

/* out[r, 0:H] = sum_{k<K, idx[r*K+k] >= 0} src[idx[r*K+k], 0:H]
 * Replaces index_select_ND(...).sum(dim=1)  (utils.py:176-193 + models/mpn.py:89-90,
 * 101-102, 201-209, 215-216) without materialising the [n_out, K, H] tensor.  The same
 * kernel run on the transposed tables is the backward of every gather-sum. */
int rr_gather_sum_f32(const float* src, int64_t n_src, int64_t ld_src,
                      const int32_t* idx, int64_t n_out, int K, int H,
                      float* out, int64_t ld_out, rr_stream_t stream);

/* rr_gather_sum_f32 whose output row 0 is instead the fixed-order sum of `n_partial` partial rows
 * (row0_partial[i, 0:H], i ascending): the padding row's adjoint delivered by rr_linear_args.colsum_partial.
 * The one wavefront that owns row 0 does the sum while the others gather - no extra pass, no extra launch. */
int rr_gather_sum_padrow_f32(const float* src, int64_t n_src, int64_t ld_src,
                             const int32_t* idx, int64_t n_out, int K, int H,
                             const float* row0_partial, int64_t n_partial, int64_t ld_partial,
                             float* out, int64_t ld_out, rr_stream_t stream);
/* Both of the above in one entry point (row0_partial may be NULL), plus an optional magnitude output:
 * max |out| maxed into the magnitude slot amax_out - see rr_gather_epi.amax_out. */
int rr_gather_sum_amax_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K,
                           int H, const float* row0_partial, int64_t n_partial, int64_t ld_partial, float* out,
                           int64_t ld_out, float* amax_out, rr_stream_t stream);

/* out[r, 0:H] = sum_{j in [offsets[r], offsets[r+1])} src[idx[j], 0:H]   (CSR form, offsets has n_out+1 entries)
 * The adjoint of index_select_ND (utils.py:176-193) for a GENERIC index tensor: autograd's index_select
 * backward is a scatter-add; here the sources of every destination row are listed (sorted by destination, stable)
 * and summed in that fixed order - no atomics, and no pad width to learn with a device->host sync. */
/* out[r, 0:H] = sum_k (mask[j, :] > 0 ? src[j, :] * scale : 0), j = idx[r*K+k] >= 0 (src and mask share n_src / ld_src):
 * rr_relu_bwd_f32 followed by rr_gather_sum_f32 in one pass, bit-identical to the sequence.  The shared-prefix reactant
 * pass (train_listwise.py:188: every candidate carries its query's reactant) sums each distinct bond's gradient over
 * its copies this way. */
int rr_gather_sum_masked_f32(const float* src, const float* mask, int64_t n_src, int64_t ld_src,
                             const int32_t* idx, int64_t n_out, int K, int H, float scale,
                             float* out, int64_t ld_out, rr_stream_t stream);
/* The same sum when the mask is dropout_j(y[r]) - the copies of a shared row, each with its own dropout mask, as
 * rr_gather_dropout_f32(y, ..., drop_p, drop_seed) produced them:
 *   out[r, c] = sum_k ( kept(drop_seed, j * H + c) && y[r, c] > 0 ? src[j, c] * scale : 0 ),  j = idx[r*K+k] >= 0
 * Equal to rr_gather_sum_masked_f32 on the materialised copies bit for bit, without reading them (ABI revision 5).
 * Needs H % 4 == 0 and 16-byte aligned rows (RR_ERR_ALIGN otherwise). */
int rr_gather_sum_dropmask_f32(const float* src, int64_t n_src, int64_t ld_src, const float* y, int64_t ld_y,
                               const int32_t* idx, int64_t n_out, int K, int H, float drop_p, uint64_t drop_seed,
                               float scale, float* out, int64_t ld_out, rr_stream_t stream);

/* The sum over idx of a tensor that is itself a sum of n_srcs tensors of one shape (ABI revision 8):
 *   out[r, :] = sum_{k<K, j = idx[r*K+k] >= 0} ( ((srcs[0][j, :] + srcs[1][j, :]) + srcs[2][j, :]) + ... )
 * In the shared-prefix reactant backward a copy's d input is the sum of the dZ of every per-copy layer (models/mpn.py:94-97)
 * and only its sum over the copies is needed: this replaces n_srcs - 1 rr_axpby_f32 passes over [n_src, H] plus a one-source
 * gather by one pass that reads every addend once.  Same additions, same order: bit-identical to that sequence.
 * 1 <= n_srcs <= RR_MAX_GATHER_SRCS; `srcs` is a HOST array of device pointers, all [n_src, ld_src].  H % 4 == 0 and 16-byte
 * aligned rows (RR_ERR_ALIGN otherwise - callers then run the sequence above). */
#define RR_MAX_GATHER_SRCS 4
int rr_gather_sum_multi_f32(const float* const* srcs, int n_srcs, int64_t n_src, int64_t ld_src,
                            const int32_t* idx, int64_t n_out, int K, int H,
                            float* out, int64_t ld_out, rr_stream_t stream);

/* Gather-sum with a fused epilogue - the form the backward chain uses (ABI revision 4):
 *   out[r, :] = M_r( sum_{k<K, idx[r*K+k] >= 0} src[idx[r*K+k], :] )  +  sum_{j<n_adds} adds[j][r, :]
 *   M_r(g)[c] = g[c]                                   without a mask
 *             = (mask[r, c] > 0) ? g[c] * mask_scale : 0   with `mask` ([n_out, ld_mask] f32) or `mask_bits` (the sign-bit
 *               image rr_linear_args.mask_bits_out wrote for that activation: rr_mask_bits_row_bytes(H) bytes per row;
 *               preferred when both are given - 1/32 of the bytes)
 * The adjoint of a message-passing layer never needs the gathered gradient as such: d message is consumed masked by the
 * ReLU / dropout pattern of the layer below (dZ = d message * (y > 0) / (1 - p), models/mpn.py:94-97) and the gradient of
 * the residual `input` (:94) is the sum of every iteration's dZ.  With the mask and the sum applied here, by the HBM-bound
 * kernel that produces the gradient, the dX GEMM downstream reads dZ as a plain operand (no mask, no dZ side output
 * written from inside its k-loop) and rr_relu_bwd_sum_f32's extra pass disappears.  The addends are summed in order
 * starting from zero and the masked gather is added last - the order of rr_relu_bwd_sum_f32 - so the result equals the
 * separate-kernel sequence bit for bit.  row0_partial (may be NULL) as in rr_gather_sum_padrow_f32; the epilogue applies
 * to row 0 as well.  Requires H % 4 == 0 and 16-byte aligned rows everywhere (RR_ERR_ALIGN otherwise); `epi` is a HOST
 * struct of device pointers. */
#define RR_MAX_GATHER_ADDS 15
typedef struct rr_gather_epi {
  const float* mask;        int64_t ld_mask;
  const uint8_t* mask_bits;
  float mask_scale;
  int n_adds;               int64_t ld_add;      /* every addend is [n_out, ld_add] */
  const float* adds[RR_MAX_GATHER_ADDS];
  float* amax_out;          /* optional: max |out| maxed into this magnitude slot (RR_AMAX_FLOATS floats, see rr_amax_f32) - the
                               bound a two-f16-term GEMM needs of this result (rr_linear_args.a1_amax) without a pass over it */
} rr_gather_epi;
int rr_gather_sum_epi_f32(const float* src, int64_t n_src, int64_t ld_src,
                          const int32_t* idx, int64_t n_out, int K, int H,
                          const float* row0_partial, int64_t n_partial, int64_t ld_partial,
                          const rr_gather_epi* epi, float* out, int64_t ld_out, rr_stream_t stream);

int rr_gather_sum_csr_f32(const float* src, int64_t n_src, int64_t ld_src,
                          const int32_t* offsets, const int32_t* idx, int64_t n_out, int H,
                          float* out, int64_t ld_out, rr_stream_t stream);

/* f_bonds[b, :] = [ f_atoms[b2a[b], 0:atom_fdim] | fbond[b, 0:bond_fdim] | zeros up to ld_out ]
 * The reference stores a directed bond's feature row as its source atom's features followed by the bond's own
 * (features/featurization.py:198-199), i.e. 61 of the 83 columns repeat f_atoms.  A packed step therefore ships only the
 * bond columns (reactranker_amd/shards.py) and this kernel rebuilds the [n_bonds, ld_out] array in HBM.  b2a values must
 * lie in [0, n_atoms). */
int rr_build_fbonds_f32(const float* f_atoms, int64_t n_atoms, int64_t ld_fa, int atom_fdim, const int32_t* b2a,
                        const float* fbond, int64_t ld_fbb, int bond_fdim, int64_t n_bonds,
                        float* out, int64_t ld_out, rr_stream_t stream);

/* out[r] = (ia[r] >= 0 ? a[ia[r]] : 0) - (im[r] >= 0 ? m[im[r]] : 0)
 * Replaces  message = a_message[b2a] - message[b2revb]  (models/mpn.py:91-92); with the
 * transposed tables (b2t, b2revb) it is that line's backward. */
int rr_gather_diff_f32(const float* a, int64_t n_a, int64_t ld_a, const int32_t* ia,
                       const float* m, int64_t n_m, int64_t ld_m, const int32_t* im,
                       int64_t n_out, int H, float* out, int64_t ld_out, rr_stream_t stream);

/* out[r, c] = keep(seed, r*H + c) ? src[idx[r], c] / (1 - p) : 0     (idx[r] < 0 -> 0)
 * Expands rows shared by several destinations and applies each destination's own dropout mask: the
 * message after the first W_h of a de-duplicated reactant (same mask stream as rr_linear_f32's epilogue). */
int rr_gather_dropout_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int H,
                          float drop_p, uint64_t drop_seed, float* out, int64_t ld_out, rr_stream_t stream);
/* The same with a magnitude output (see rr_gather_epi.amax_out): max |out| maxed into the slot amax_out (may be NULL). */
int rr_gather_dropout_amax_f32(const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int H,
                               float drop_p, uint64_t drop_seed, float* out, int64_t ld_out, float* amax_out,
                               rr_stream_t stream);

/* out[0:H] (+)= sum_r w[r] * x[r, 0:H]   (w == NULL -> all ones).
 * Backward of the padding row: the reference gathers row 0 (K - deg(a)) times per atom
 * (features/featurization.py:286), so d_src[0] = sum_a npad[a] * d_out[a].  Also used for
 * bias gradients.  workspace: rr_colsum_workspace_bytes(n, H) bytes. */
size_t rr_colsum_workspace_bytes(int64_t n, int H);
int rr_weighted_colsum_f32(const float* x, int64_t n, int64_t ld, const float* w, int H,
                           float* out, int accumulate, void* workspace, size_t workspace_bytes,
                           rr_stream_t stream);

/* ------------------------------------------------------------------ dense layers --- */

typedef enum rr_act { RR_ACT_NONE = 0, RR_ACT_RELU = 1 } rr_act;

/* C[m, n] = dropout(act(residual[m,n] + bias[n] + sum_k A[m,k] * W[n,k]))
 *
 * A is the concatenation [A1 | A2] (k1 + k2 columns, either may be 0 wide):
 *   A1 row m = (a1_idx ? a1[a1_idx[m]] : a1[m])  -  (a1_sub ? a1_sub[a1_sub_idx ? a1_sub_idx[m] : m] : 0)
 *   A2 row m = a2[m]
 * and, if a_mask != NULL, every A element is multiplied by (a_mask[m,k] > 0) * mask_scale
 * (ReLU/dropout backward fused into the operand load; requires k2 == 0).
 *
 * One call replaces, at the reference's sites:
 *   W_i(f_bonds) + relu                         models/mpn.py:80-81, 194-195
 *   a_message[b2a] - message[b2revb]; W_h; input + .; relu; dropout   models/mpn.py:91-97
 *   cat([f_atoms, a_message]); W_o; relu; dropout                     models/mpn.py:103-105, 217-219
 *   cat(nei_a_message, nei_f_bonds).sum; W_h; ...                     models/mpn.py:208-213
 *   FFN Linear/ReLU/Dropout chain                                     models/base_model.py:32-60
 * and their input-gradient GEMMs (dX = dZ * W  ==  same call with w = W^T).
 *
 * Arithmetic: fp32 operands and results, fp32 accumulation.  Two kernels serve the call: with w_packed = 1 every
 * product runs on the f32 MFMA (v_mfma_f32_16x16x4_f32, bit for bit an fmaf chain); with w_packed = 2 (weights
 * packed as three exact bf16 terms by rr_pack_weights_f32 / rr_pack_weight_f32 with split = 1) each f32 operand is
 * written EXACTLY as three bf16 terms and the six products down to 2^-18 |x w| are accumulated in f32 by
 * v_mfma_f32_16x16x32_bf16 - no operand bit is dropped, what is omitted lies below 2^-26 |x w| per product, and the
 * measured error against f64 is at or below the f32-MFMA kernel's.  Differences in kind on
 * that path: an infinite operand, or a finite |x| >= 3.3962e38, turns its output row into NaN (the f32 MFMA gives
 * +-inf).  With w_packed = 3 each operand is scaled by a power of two S that puts its tensor's largest magnitude (the
 * caller's a*_amax bounds; the weight's own, found by the packer) just below 2^15 and written as TWO f16 terms,
 * S x = h + l, h = f16(S x), l = f16(S x - h) (round to nearest even; the remainder is exact): 22 significant bits of
 * every element that is within 2^-18 of its tensor's largest, an absolute error below 2^-40 of that largest otherwise;
 * the three products h h, h l, l h are accumulated in f32 by v_mfma_f32_16x16x32_f16 and the result is scaled back
 * (exactly).  Omitted: l l, below 2^-22 |x w|.  Measured against f64 on N(0,1) operands, K = 300: mean error
 * 2.6e-8 of sum |x||w| (three bf16 terms 1.6e-8, the f32 MFMA chain 2.0e-8), maximum below the f32 chain's
 * (tools/f16_gemm_bench.py).  Dropout keep-mask = rr_dropout_keep(seed, m*N + n).
 * `residual` may alias `c` (in-place accumulate).  Vector loads need 16-byte aligned base
 * pointers and leading dimensions that are multiples of 4; otherwise a scalar path runs. */
typedef struct rr_linear_args {
  int64_t M;
  int N;
  const float* a1;       int64_t lda1;      int k1;   const int32_t* a1_idx;
  const float* a1_sub;   int64_t lda1_sub;            const int32_t* a1_sub_idx;
  const float* a2;       int64_t lda2;      int k2;
  const float* a_mask;   int64_t ld_mask;   float mask_scale;
  float* dz_out;         int64_t ld_dz;     int dz_accumulate;  /* with a_mask: also materialise the masked operand,
                                                       dz_out[m,k] (+)= A[m,k] — e.g. d_input += dZ of every
                                                       message-passing step (the residual `input +` of mpn.py:95).
                                                       Needs k1 % 4 == 0 and 16-byte aligned rows. */
  const float* w;        int64_t ldw;               /* [N, k1+k2] row-major (nn.Linear.weight), or */
  int w_packed;                                     /* 1: the zero-padded layout of rr_pack_weight_f32;
                                                       2: the three-bf16-term images of rr_pack_weights_f32
                                                          (rr_pack_desc.split = 1): f32 result on the bf16 matrix core;
                                                       3: its two-f16-term images (rr_pack_desc.split = 2): 22 significant
                                                          bits per operand, half the matrix instructions (see below) */
  const float* bias;                                /* [N] or NULL */
  const float* residual; int64_t ldr;               /* [M, N] or NULL */
  const int32_t* residual_idx;                      /* optional: row m adds residual[residual_idx[m]] (shared `input`
                                                       of a de-duplicated reactant, models/mpn.py:95) */
  int act;                                          /* rr_act */
  float drop_p;          uint64_t drop_seed;        /* drop_p == 0 -> no dropout */
  float* c;              int64_t ldc;
  float* c_pre;          int64_t ld_pre;            /* optional second output: the pre-activation value
                                                       (input = W_i(f_bonds) next to message = relu(input),
                                                       models/mpn.py:80-81) */
  const float* colsum_w;                            /* optional third output (fast path only): per-row weights w[M] ... */
  float* colsum_partial; int64_t ld_partial;        /* ... and partial[rr_linear_colsum_rows(M), ld_partial >= N]:
                                                       partial[i, n] = sum over the 64 rows of row block i of w[m]*C[m,n].
                                                       The padding row's adjoint (a2b is right-padded with bond 0,
                                                       features/featurization.py:286, so row 0 of a gathered source
                                                       receives sum_a npad[a] * d_out[a]) falls out of the GEMM that
                                                       produces d_out instead of a separate pass over it; the partial
                                                       rows are summed, in order, by rr_gather_sum_padrow_f32. */
  uint8_t* mask_bits_out;                           /* optional fourth output (w_packed = 2, N % 4 == 0): the sign of every
                                                       stored element, 1 bit each, rr_mask_bits_row_bytes(N) bytes per row:
                                                       per block of up to 304 columns 2 x 20 bytes - byte 20*h + t holds
                                                       columns 16*t + 8*h .. + 7 (bit e = column + e).  What a later
                                                       dX GEMM needs of this activation (a_mask > 0), at 1/32 of the bytes */
  const uint8_t* a_mask_bits;                       /* alternative to a_mask (w_packed >= 2, k2 = 0, k1 % 4 == 0): such a bit
                                                       image over the k1 columns of A; a_mask is then not read */
  const float* a1_amax;                             /* w_packed = 3 only (required there for every operand present): DEVICE */
  const float* a1_sub_amax;                         /* magnitude slots (RR_AMAX_FLOATS floats each, see rr_amax_f32) whose maximum */
  const float* a2_amax;                             /* is >= max |x| over the elements of a1 / a1_sub / a2 this call can read (a
                                                       bound that is too large costs low-end precision, one that is too small
                                                       overflows f16) */
  float* c_amax_out;                                /* optional outputs (w_packed = 3): max |C| stored, and the same for dz_out, */
  float* dz_amax_out;                               /* maxed into a magnitude slot - the bound the NEXT GEMM needs of this one's
                                                       result without a pass over it (zero the slot first) */
} rr_linear_args;

int rr_linear_f32(const rr_linear_args* args, rr_stream_t stream);
int64_t rr_mask_bits_row_bytes(int N);
/* Row blocks (= partial rows written through colsum_partial) of an M-row call. */
int64_t rr_linear_colsum_rows(int64_t M);

/* Packed weight layout for the straight-line fast path of rr_linear_f32:
 *   dst[r, 0:k1] = L[r, 0:k1];  dst[r, r16(k1) : r16(k1)+k2] = L[r, k1:k1+k2];  zeros elsewhere;
 *   row stride rr_packed_weight_ld(k1,k2) = r16(k1) + r16(k2), r16 = round up to 16;
 *   L[r, c] = src[r*ld_src + c0 + c]            (transpose = 0: a column slice of the weight)
 *           = src[c*ld_src + c0 + r]            (transpose = 1: its transpose, for dX = dZ * W)
 * Weights change every optimizer step, so the mirror re-packs them once per forward. */
int64_t rr_packed_weight_ld(int k1, int k2);
int rr_pack_weight_f32(const float* src, int64_t ld_src, int transpose, int rows, int c0, int k1, int k2,
                       float* dst, rr_stream_t stream);

/* rr_pack_weight_f32 for up to RR_MAX_PACK weights in one launch (`descs` is a HOST array). */
#define RR_MAX_PACK 24
typedef struct rr_pack_desc {
  const float* src;  int64_t ld_src;  int transpose, rows, c0, k1, k2;
  float* dst;
  int split;         /* 0: the f32 layout above.  1: dst (rr_split_weight_bytes(rows, k1, k2) bytes, 16-byte aligned,
                        rows <= 608) receives every element of L as three bf16 terms t0 + t1 + t2 == L[r, c] EXACTLY
                        (t0 = bf16(x), t1 = bf16(x - t0), t2 = x - t0 - t1), laid out as the LDS image of each
                        32-deep k-step of rr_linear_f32's w_packed = 2 path:
                        [k-step][16-column tile][term][lane 0..63][8 bf16], lane = (k-group of 8) * 16 + column.
                        2: the same buffer receives TWO f16 terms of S * L[r, c] per element (same layout with two terms
                        per tile), S = the power of two with 2^14 <= S * max |L| < 2^15, stored as one float right after
                        the last k-step's image (rr_linear_f32, w_packed = 3, reads it there) */
} rr_pack_desc;
int rr_pack_weights_f32(const rr_pack_desc* descs, int n, rr_stream_t stream);
size_t rr_split_weight_bytes(int rows, int k1, int k2);

/* dW[n, k] (+)= sum_m dZ[m,n] * X[m,k],   dbias[n] (+)= sum_m dZ[m,n]
 * dZ[m,n] = dy[m,n] * (mask ? (mask[m,n] > 0) * mask_scale : 1);  X = [X1 | X2] described
 * exactly like A above.  Weight gradients of every nn.Linear on the path; the sum over the
 * M rows is split across workgroups and finished by a fixed-order second pass.
 * workspace: rr_linear_wgrad_workspace_bytes(M, N, k1+k2). */
typedef struct rr_wgrad_args {
  int64_t M;
  int N;
  const float* dy;       int64_t ld_dy;
  const float* mask;     int64_t ld_mask;   float mask_scale;
  const float* x1;       int64_t ldx1;      int k1;   const int32_t* x1_idx;
  const float* x1_sub;   int64_t ldx1_sub;            const int32_t* x1_sub_idx;
  const float* x2;       int64_t ldx2;      int k2;
  float* dw;             int64_t ld_dw;             /* [N, k1+k2] */
  float* dbias;                                     /* [N] or NULL */
  int accumulate;                                   /* 0: overwrite, 1: add into dw/dbias */
  void* workspace;       size_t workspace_bytes;
  int split;                                        /* 1: three-bf16-term operands on the bf16 matrix core (f32-equivalent
                                                       accuracy, see rr_pack_desc.split) where the geometry allows vector
                                                       loads (16-byte aligned rows, N % 4 == 0); the f32 path otherwise.
                                                       2: two-f16-term operands (rr_linear_args.w_packed = 3), which needs */
  const float* dy_amax;                             /* ... the magnitude slots of dy, x1, x1_sub, x2 (as rr_linear_args.a1_amax) */
  const float* x1_amax;
  const float* x1_sub_amax;
  const float* x2_amax;
} rr_wgrad_args;

size_t rr_linear_wgrad_workspace_bytes(int64_t M, int N, int K);
int rr_linear_wgrad_f32(const rr_wgrad_args* args, rr_stream_t stream);

/* A magnitude slot: RR_AMAX_LANES floats RR_AMAX_STRIDE floats apart (RR_AMAX_FLOATS floats in all, device memory) whose
 * MAXIMUM is the bound - a producer's workgroups max their values into different lanes, because atomics on one busy address
 * serialise (~5 ns each).  Zero the RR_AMAX_FLOATS floats before the first producer; every kernel only ever raises them.
 * rr_amax_f32: slot = max(slot, max over the [rows, cols] block of |x[r * ld + c]|)  (NaN elements are skipped).
 * The operand bounds of the two-f16-term GEMMs (rr_linear_args.a1_amax, rr_wgrad_args.dy_amax). */
#define RR_AMAX_LANES 16
#define RR_AMAX_STRIDE 32
#define RR_AMAX_FLOATS (RR_AMAX_LANES * RR_AMAX_STRIDE)
int rr_amax_f32(const float* x, int64_t rows, int cols, int64_t ld, float* amax, rr_stream_t stream);

/* ------------------------------------------------------------------ elementwise ---- */

/* 1 iff element `index` is kept by dropout stream `seed` at rate p (host + device agree). */
int rr_dropout_keep_host(uint64_t seed, uint64_t index, float p);

/* out[i] = keep(seed, i) ? x[i] / (1 - p) : 0   — nn.Dropout in training mode with the counter-based
 * mask stream; the same call on a gradient is its backward.  (FFN input dropout, models/base_model.py:32-36.) */
int rr_dropout_f32(const float* x, int64_t n, float p, uint64_t seed, float* out, rr_stream_t stream);

/* dz = dy * (y > 0) * scale (dz may be NULL when only the accumulation is wanted);  if acc != NULL: acc += dz.   (ReLU + inverted-dropout backward;
 * y is the layer's stored post-dropout output, so y > 0 <=> kept and active.) */
int rr_relu_bwd_f32(const float* dy, const float* y, float scale, float* dz, float* acc,
                    int64_t n, rr_stream_t stream);

/* out = sum_{k<n_adds} adds[k] + dy * (y > 0) * scale.  `adds` is a HOST array of n_adds (<= 15) device pointers.
 * The gradient of a residual read by several layers: `input` of models/mpn.py:80 enters every message-passing
 * iteration (:94), so d input = sum_it dZ_it + relu'(input) * d message_0 - formed in one pass over memory from
 * the per-iteration dZ buffers the input-gradient GEMMs write (rr_linear_args.dz_out). */
int rr_relu_bwd_sum_f32(const float* dy, const float* y, float scale, const float* const* adds, int n_adds,
                        float* out, int64_t n, rr_stream_t stream);

/* out = alpha * a + beta * b   (b may be NULL).  diff = p_h - r_h (models/base_model.py:168). */
int rr_axpby_f32(float alpha, const float* a, float beta, const float* b, float* out,
                 int64_t n, rr_stream_t stream);

/* Adam update of up to RR_MAX_ADAM tensors in ONE launch (ABI revision 6) - the optimizer the reference builds
 * (train/utils.py:100-113: torch.optim.Adam, lr 1e-4, weight_decay 0; betas / eps torch's defaults), per element:
 *   g' = g + weight_decay * p;  m = m + (g' - m) * (1 - beta1);  v = beta2 * v + (1 - beta2) * g' * g'
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * i.e. torch's own formula (fused_adam_utils.cuh: scalars in double, each element rounded to f32 once), with `step` the
 * 1-based count of this update.  `descs` is a HOST array;
 * a tensor with g == NULL is skipped.  torch's multi-tensor kernel walks 64k-element chunks, ~20 workgroups for this
 * model's 0.8 M parameters (45 us on MI355X); here a workgroup takes 1024 elements. */
#define RR_MAX_ADAM 64
typedef struct rr_adam_tensor { float* p; const float* g; float* m; float* v; int64_t n; } rr_adam_tensor;
int rr_adam_step_f32(const rr_adam_tensor* descs, int n_tensors, int64_t step, double lr, double beta1, double beta2, double eps,
                     double weight_decay, rr_stream_t stream);

/* FFN output heads (models/base_model.py:61-106), applied to raw[M, N] -> out[M, N]. */
typedef enum rr_head {
  RR_HEAD_IDENTITY = 0,            /* 'no_softplus', 'with_softplus' with task_num==1, ... */
  RR_HEAD_SOFTPLUS = 1,            /* 'listnet_with_softplus' */
  RR_HEAD_SOFTPLUS_PLUS1 = 2,      /* 'listnet_with_uncertainty', 'evidential' */
  RR_HEAD_EVIDENTIAL_RANKING = 3,  /* [score, softplus(u)+1e-6] interleaved (:91-98) */
  RR_HEAD_GAUSSIAN_SOFTPLUS = 4,   /* [mu, softplus(v)] (:71-82) */
  RR_HEAD_LOGNORM_SOFTPLUS = 5,    /* [softplus(mu)+1e-6, softplus(v)+1e-6] (:83-90) */
  RR_HEAD_EVIDENTIAL4_SOFTPLUS = 6 /* [mu, sp+1e-6, sp+1e-6+1, sp+1e-6] (:61-70) */
} rr_head;
int rr_head_fwd_f32(const float* raw, int64_t M, int N, int head, float* out, rr_stream_t stream);
int rr_head_bwd_f32(const float* dout, const float* raw, int64_t M, int N, int head, float* draw,
                    rr_stream_t stream);

/* ------------------------------------------------------------------ readout -------- */

/* out[m, 0:H] = mean over atoms [start, start+size) of x;  out[m, H:H+F] = feat[m, 0:F];
 * then inverted dropout (stream seed, index m*(H+F)+c) if drop_p > 0.
 * a_scope is [M,2] = (start, size) as BatchMolGraph.a_scope (featurization.py:276);
 * size == 0 gives a zero row (cached_zero_vector, models/mpn.py:226-227).
 * Replaces the per-molecule narrow/sum/div/stack loop + cat (models/mpn.py:224-238). */
int rr_segment_mean_fwd_f32(const float* x, int64_t ldx, const int32_t* a_scope, int64_t M, int H,
                            const float* feat, int F, float drop_p, uint64_t drop_seed,
                            float* out, int64_t ld_out, rr_stream_t stream);
/* dx[a, 0:H] = dout[mol(a), 0:H] * keep/(1-p) / size(mol(a));  atom2mol[a] < 0 -> zero row. */
int rr_segment_mean_bwd_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope,
                            const int32_t* atom2mol, int64_t n_atoms, int H, int F,
                            float drop_p, uint64_t drop_seed,
                            float* dx, int64_t ldx, rr_stream_t stream);

/* The same followed by the ReLU / dropout backward of the layer that produced the readout's input:
 *   dx[a, c] = (mask[a, c] > 0) ? dx[a, c] * mask_scale : 0      (mask / mask_bits as in rr_gather_epi; one of them required)
 * so the two dX GEMMs of W_o's column blocks (models/mpn.py:217-219) read a plain operand.  H % 4 == 0 and aligned rows. */
int rr_segment_mean_bwd_masked_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope,
                                   const int32_t* atom2mol, int64_t n_atoms, int H, int F,
                                   float drop_p, uint64_t drop_seed,
                                   const float* mask, int64_t ld_mask, const uint8_t* mask_bits, float mask_scale,
                                   float* dx, int64_t ldx, rr_stream_t stream);
/* The same with a magnitude output (see rr_gather_epi.amax_out): max |dx| maxed into the slot amax_out (may be NULL). */
int rr_segment_mean_bwd_masked_amax_f32(const float* dout, int64_t ld_dout, const int32_t* a_scope,
                                        const int32_t* atom2mol, int64_t n_atoms, int H, int F, float drop_p,
                                        uint64_t drop_seed, const float* mask, int64_t ld_mask, const uint8_t* mask_bits,
                                        float mask_scale, float* dx, int64_t ldx, float* amax_out, rr_stream_t stream);

/* ------------------------------------------------------------------ ranking losses - */
/* Lists are described by seg_off[Q+1] (prefix sums of the reference's `scope` list);
 * max_len = longest list (host value, sizes the LDS staging; supported up to 8192).
 * score/targets/var elements are read at p[i * stride].  `loss` is one float on the device.
 * `gloss` is the upstream gradient (one float on the device). */

/* ListMLE: MLEloss + LogCumsumExp (train/loss.py:9-99). Ties in targets break by index. */
int rr_listmle_fwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len,
                       float* loss, float* partial /* [Q] */, rr_stream_t stream);
int rr_listmle_bwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len, const float* gloss,
                       float* dscore, int64_t dscore_stride, rr_stream_t stream);

/* Fused loss + gradient ("step") forms, ABI revision 8: ONE launch writes `loss` AND d loss / d score for an upstream
 * gradient of one - what `loss.backward()` feeds a loss that is the root of the graph (train/train_listwise.py:287-288) -
 * so a training step needs no second loss kernel, no separate reduction launch and no device-side gradient seed.  `loss`,
 * `partial` and `dscore` hold the bits rr_*_fwd_f32 / rr_*_bwd_f32 (with *gloss == 1.0f) write: same operations in the same
 * order, the per-query partials summed in the same fixed order by the workgroup that finishes last.  `counter` is ONE
 * zero-initialised device word per concurrent launch that the caller keeps; the kernel leaves it at zero. */
int rr_listmle_step_f32(const float* score, int64_t score_stride, const float* targets,
                        const int32_t* seg_off, int Q, int max_len, float* loss, float* partial /* [Q] */,
                        unsigned int* counter, float* dscore, int64_t dscore_stride, rr_stream_t stream);
int rr_listnet_step_f32(const float* score, int64_t score_stride, const float* targets,
                        const int32_t* seg_off, int Q, int max_len, int64_t total, float* loss, float* partial,
                        unsigned int* counter, float* dscore, int64_t dscore_stride, rr_stream_t stream);
int rr_evidential_ranking_step_f32(const float* mu, const float* var, int64_t stride, const float* targets,
                                   const int32_t* seg_off, int Q, int max_len, float* loss, float* partial,
                                   unsigned int* counter, float* dmu, float* dvar, int64_t dstride, rr_stream_t stream);

/* ListNet top-1, one global mean over all candidates (train/loss.py:327-352). */
int rr_listnet_fwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len, int64_t total /* = seg_off[Q] */,
                       float* loss, float* partial, rr_stream_t stream);
int rr_listnet_bwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len, int64_t total, const float* gloss,
                       float* dscore, int64_t dscore_stride, rr_stream_t stream);

/* UC-Listwise evidential_ranking live branch (train/loss.py:526-556): mu/var are the two
 * columns of the [M,2] model output. */
int rr_evidential_ranking_fwd_f32(const float* mu, const float* var, int64_t stride,
                                  const float* targets, const int32_t* seg_off, int Q, int max_len,
                                  float* loss, float* partial, rr_stream_t stream);
int rr_evidential_ranking_bwd_f32(const float* mu, const float* var, int64_t stride,
                                  const float* targets, const int32_t* seg_off, int Q, int max_len,
                                  const float* gloss, float* dmu, float* dvar, int64_t dstride,
                                  rr_stream_t stream);

/* RankNet pairwise logistic, 'sum_session' (train/train_pairwise.py:99-122):
 * loss_sum = sum over queries with >= 1 positive pair of sum_ij pos*log(1+e^{-s d}) + neg*log(1+e^{s d});
 * pairs = total ordered pairs of those queries (:106).  partial is [2*Q] floats. */
int rr_ranknet_fwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len, float sigma,
                       float* loss_sum, int64_t* pairs, float* partial, rr_stream_t stream);
/* mode 0: dscore = gloss * d(loss_sum)/d(score)   (what autograd gives 'sum_session')
 * mode 1: dscore = gloss * lambda_i, the 'accelerate_grad' row sums (train_pairwise.py:125-133) */
int rr_ranknet_bwd_f32(const float* score, int64_t score_stride, const float* targets,
                       const int32_t* seg_off, int Q, int max_len, float sigma, int mode,
                       const float* gloss, float* dscore, int64_t dscore_stride, rr_stream_t stream);

/* Pointwise: nn.MSELoss (train/train_listwise.py:166-167) and GaussDisLoss (train/loss.py:154-162).
 * partial: rr_pointwise_partial_count(n) floats. */
int64_t rr_pointwise_partial_count(int64_t n);
int rr_mse_fwd_f32(const float* pred, int64_t stride, const float* targets, int64_t n,
                   float* loss, float* partial, rr_stream_t stream);
int rr_mse_bwd_f32(const float* pred, int64_t stride, const float* targets, int64_t n,
                   const float* gloss, float* dpred, int64_t dstride, rr_stream_t stream);
int rr_gauss_nll_fwd_f32(const float* mean, const float* var, int64_t stride, const float* targets,
                         int64_t n, float* loss, float* partial, rr_stream_t stream);
int rr_gauss_nll_bwd_f32(const float* mean, const float* var, int64_t stride, const float* targets,
                         int64_t n, const float* gloss, float* dmean, float* dvar, int64_t dstride,
                         rr_stream_t stream);

/* Per-query ranking evaluation on the device - the metric halves of `ranking_metrics` (train/eval.py:475-555),
 * `evaluate_top_scores` (:76-177) and `calculate_ndcg` (:329-457) without the one-forward-per-query loops and the
 * Python lists; same list description as the losses.  ABI revision 6: `ratio`, `ndcg_cut`, 12 statistics per query.
 *   order[off_q + r] = position (inside its list) of the candidate ranked r-th by score, descending, ties
 *                      keeping the original order (python's stable sorted(..., reverse=True), :516-519)
 *   stats[q*RR_RANKING_NSTATS + 0..11], float64:
 *     0..6  ranking_metrics: top-1 hit, PREDICTED top-1 in the target top-25%, recall@25%, NDCG1, NDCG2, NDCG25%,
 *           NDCG_all (:521-546; exp gains, compute_NDCG :460-472, python round() for the 25% cut, NDCG2 without
 *           discount exactly as the reference computes it)
 *     7     NDCG@10 with exp2 gains of `targets` taken as relevance grades (metrics.py:54-71)
 *     8     evaluate_top_scores' third value: the TARGET's top-1 (first maximum, :133) in the predicted
 *           top-round(C*ratio) (:144-146, :156-159) - not the same quantity as stat 1
 *     9     calculate_ndcg's NDCG: gains C + 1 - predicted rank over the first ceil(C*ndcg_cut) positions of the
 *           target order (:406-425, cal_NDCG :309-325); ties by position (the reference's torch.sort leaves them open)
 *     10    calculate_ndcg's KL(softmax(targets) || softmax(score)) with un-shifted f32 exponentials (:401-404)
 *     11    evaluate_top_scores' second value: share of the predicted top-round(C*ratio) inside the target
 *           top-round(C*ratio) (:147-151); equals stat 2 at ratio = 0.25
 *   evaluate_top_scores' first value (first-maximum top-1, :133-136) is stat 0.
 * 0 <= ratio, ndcg_cut <= 1.  The trainer-level numbers are the means over queries. */
#define RR_RANKING_NSTATS 12
int rr_ranking_metrics_f32(const float* score, int64_t score_stride, const float* targets, const int32_t* seg_off,
                           int Q, int max_len, double ratio, double ndcg_cut, int32_t* order, double* stats,
                           rr_stream_t stream);

/* LogCumsumExp along dim 0 of a 1-D tensor (train/loss.py:9-61); n <= 8192.
 * backward keeps the reference's un-shifted exp(x) (:59). */
int rr_logcumsumexp_fwd_f32(const float* x, int n, float* y, rr_stream_t stream);
int rr_logcumsumexp_bwd_f32(const float* x, const float* y, const float* gy, int n, float* gx,
                            rr_stream_t stream);

/* ------------------------------------------------------------------ graph packing (HOST) */
/* BatchMolGraph.__init__ (features/featurization.py:246-288) as one native call over
 * concatenated per-molecule arrays; ALL pointers here are HOST memory.
 *   in : mol_atoms[M], mol_bonds[M]               atoms / directed bonds per molecule
 *        f_atoms_cat[sum atoms, atom_fdim], f_bonds_cat[sum bonds, bond_fdim]
 *        b2a_local, b2revb_local [sum bonds]      molecule-local indices
 *        a2b_off[sum atoms + 1], a2b_local[...]   incoming-bond lists (CSR, molecule-local)
 *        K_override: 0 -> K = max(1, max in-degree) (:281); else pad width to use (>= that)
 *   out: sizes via rr_pack_sizes; arrays with the padding row 0 (:255-264):
 *        f_atoms[nA, ld_fa], f_bonds[nB, ld_fb] (rows zero-padded to the given ld),
 *        a2b[nA,K], b2a[nB], b2revb[nB], a2a[nA,K] (= b2a[a2b], :326-327), a_scope[M,2],
 *        and the backward tables: a2b_rev_t[nA,K] (b2revb[a2b], pads -1, row0 = {0,-1..}),
 *        b2t[nB] (target atom b2a[b2revb[b]]; b2t[0] = -1), a2a_t[nA,K] (a2a with pads -1),
 *        npad[nA] (float, K - deg; npad[0] = K), atom2mol[nA] (-1 for the pad row). */
int rr_pack_sizes(const int32_t* mol_atoms, const int32_t* mol_bonds, int64_t M,
                  const int64_t* a2b_off, int K_override,
                  int64_t* nA, int64_t* nB, int32_t* K);
int rr_pack_graphs(const int32_t* mol_atoms, const int32_t* mol_bonds, int64_t M,
                   const float* f_atoms_cat, int atom_fdim, const float* f_bonds_cat, int bond_fdim,
                   const int32_t* b2a_local, const int32_t* b2revb_local,
                   const int64_t* a2b_off, const int32_t* a2b_local, int K,
                   float* f_atoms, int64_t ld_fa, float* f_bonds, int64_t ld_fb,
                   int32_t* a2b, int32_t* b2a, int32_t* b2revb, int32_t* a2a, int32_t* a_scope,
                   int32_t* a2b_rev_t, int32_t* b2t, int32_t* a2a_t, float* npad, int32_t* atom2mol);
/* Backward tables for a batch that already exists as padded arrays (e.g. a reference
 * BatchMolGraph's tensors converted to int32); HOST pointers. */
int rr_derive_tables(const int32_t* a2b, const int32_t* b2a, const int32_t* b2revb,
                     int64_t nA, int64_t nB, int K, const int32_t* a_scope, int64_t M,
                     int32_t* a2a, int32_t* a2b_rev_t, int32_t* b2t, int32_t* a2a_t,
                     float* npad, int32_t* atom2mol);

/* ------------------------------------------------------------------ FFN head as one launch (ABI revision 8) --- */
/* The FFN head (models/base_model.py:32-60) and its input-gradient chain: up to RR_MAX_FFN dependent small GEMMs - one row per
 * molecule - in ONE launch; a workgroup owns 16 rows and walks every stage, the activations stay in LDS between stages
 * (csrc/ffn.hip).  Stage s computes  y = x_s W_s^T (+ bias) [ReLU] [dropout]  on the f32 MFMA with W_s in the packed layout of
 * rr_pack_weights_f32 (split = 0: [n_out][r16(n_in)] zero-padded), writes y to `out` (may be NULL except for the last stage)
 * and hands  x_{s+1} = post_mask ? (post_mask > 0 ? y * mask_scale : 0) : y  to the next stage, which reads its first n_in
 * columns.  rowdot = 1 (last stage only, n_out <= 8, n_in % 4 == 0, plain row-major weight rows of pitch ldw): the 16-lane dot
 * product of the scorer's last Linear.  Results are bit-identical to the same layers issued one by one through
 * rr_linear_f32 (M <= 8192 geometry or not).  Forward chain: stages = the Linear layers, ReLU + dropout on all but the last
 * (rowdot); backward chain: stages = the layers in reverse with their transposed packs, post_mask = the forward layer's
 * output below.  Returns RR_ERR_UNSUPPORTED for shapes it does not take (unaligned rows, n_out % 4 != 0, widths > 1024):
 * issue the layers one by one then. */
#ifndef RR_MAX_FFN
#define RR_MAX_FFN 8
#endif
typedef struct rr_ffn_stage {
  const float* w;  int64_t ldw;       /* packed weight [n_out][ldw], ldw = r16(n_in) (rowdot: any row-major [n_out][ldw >= n_in]) */
  const float* bias;                  /* [n_out] or NULL */
  int n_out, n_in;
  int relu, dropout, rowdot;
  uint64_t drop_seed;                 /* dropout stream of this stage (rate: rr_ffn_chain_args.drop_p; element m * n_out + n) */
  float* out;  int64_t ld_out;        /* y [M, ld_out] or NULL */
  const float* post_mask;  int64_t ld_mask;   /* [M, ld_mask] or NULL */
} rr_ffn_stage;
typedef struct rr_ffn_chain_args {
  int64_t M;
  int n_stages;
  const float* x;  int64_t ldx;       /* input of stage 0: [M, ldx], its first stage[0].n_in columns */
  float drop_p, mask_scale;
  rr_ffn_stage stage[RR_MAX_FFN];
} rr_ffn_chain_args;
size_t rr_abi_ffn_chain_size(void);
int rr_ffn_chain_f32(const rr_ffn_chain_args* args, rr_stream_t stream);

/* ------------------------------------------------------------------ whole-model step plans --- */
/* ReactionModel.forward (models/base_model.py:150-171: encoder on reactants and products, p_h - r_h, diff encoder,
 * readout, FFN + head) and its explicit backward as ONE call each: the host-side orchestration of the per-op entry
 * points above (operand wiring, dropout stream seeds, the weight-gradient / reactant-encoder streams and their events)
 * for hosts that should not - or cannot - issue ~140 launches per training step themselves.  All activations live in ONE
 * caller-provided workspace whose layout is a pure function of (model, step): rr_reaction_backward re-derives every
 * saved address, the library keeps no state between the two calls.  Requirements: H % 4 == 0, 16-byte aligned rows
 * (what the packer produces); other shapes use the per-op entry points.  Device pointers unless noted.
 * A data-parallel step is forward, loss kernel, backward, then ONE all-reduce of the gradient buffers handed to
 * rr_reaction_backward (rr_allreduce_f32 below, or the host's own collective - the Python mirror uses torch.distributed). */
#ifndef RR_MAX_FFN
#define RR_MAX_FFN 8
#endif

typedef struct rr_graph {             /* a packed batch resident in HBM (arrays of rr_pack_graphs / rr_derive_*) */
  int64_t nA, nB, M;
  int K, Kb;
  const float* f_atoms;  int64_t ld_fa;
  const float* f_bonds;  int64_t ld_fb;    /* may be NULL for rr_step.r in RR_STEP_PREFIX mode (only u.f_bonds is read) */
  const int32_t *a2b, *b2a, *b2revb, *a2a, *a_scope, *b2t, *a2a_t, *atom2mol, *b2b_t;
  const float *npad, *npad_b;
  const float* fb_sum;   int64_t ld_fbs;   /* sum_k f_bonds[a2b[a,k]] [nA, ld_fbs] (input-only; needed when diff_depth > 1) */
} rr_graph;

typedef struct rr_linear_w {          /* one nn.Linear: weight [out, in] row-major with row stride ldw, bias [out] or NULL */
  const float* w;
  const float* b;
  int out, in;
  int64_t ldw;
} rr_linear_w;

typedef struct rr_model {
  int H, depth, diff_depth, n_ffn, head;          /* head: rr_head of the FFN output (0 = none) */
  int atom_fdim, bond_fdim;                       /* 61, 83 (= 61 + 22) */
  rr_linear_w enc_wi, enc_wh, enc_wo;             /* MPN     (models/mpn.py:40-59);  enc_wh unused when depth == 1 */
  rr_linear_w dif_wi, dif_wh, dif_wo;             /* MPNDiff (models/mpn.py:161-168) */
  rr_linear_w ffn[RR_MAX_FFN];                    /* FFN Linear layers in order (models/base_model.py:32-57) */
} rr_model;

enum { RR_STEP_PLAIN = 0,    /* encoder(r) on the full reactant batch */
       RR_STEP_DEDUP = 1,    /* dropout inactive: `r` holds the DISTINCT reactants, amap / amap_t map product atoms to them */
       RR_STEP_PREFIX = 2 }; /* train mode: `u` holds the distinct reactants, only the deterministic prefix is shared */
enum { RR_PLAN_NO_SIDE_STREAM = 1, RR_PLAN_NO_AUX_STREAM = 2,
       RR_PLAN_F32_GEMM = 4,         /* encoder GEMMs on the f32 matrix core instead of the three-bf16-term path (w_packed = 2);
                                        forward and backward of a step must agree on it (it changes the workspace layout) */
       RR_PLAN_AUX_BACKWARD = 8,     /* reactant-encoder backward on the aux stream beside the product pass */
       RR_PLAN_TRAIN = 16,           /* a backward WILL follow: rr_reaction_forward also packs the transposed weights of the
                                        input-gradient GEMMs, in the same launch as the forward's packs, so the backward
                                        starts with its first GEMM instead of a pack.  Layout-changing like RR_PLAN_F32_GEMM:
                                        pass the same flags to both calls (ABI revision 6). */
       RR_PLAN_F16X2_GEMM = 32,      /* encoder GEMMs and weight gradients on two f16 terms per operand (w_packed = 3: half the
                                        matrix instructions of the three-bf16-term path, 22 significant bits per operand); the
                                        plan finds every operand's largest magnitude itself (rr_amax_f32, one slot per tensor
                                        in the workspace).  Ignored with RR_PLAN_F32_GEMM; layout-changing (ABI revision 7). */
       RR_PLAN_NO_FFN_CHAIN = 64,    /* the FFN head's layers as separate launches instead of rr_ffn_chain_f32 (same results
                                        bit for bit, same workspace layout: an A/B knob; ABI revision 8) */
       RR_PLAN_TIME = 128,           /* measurement: HIP events on the launch stream around every split-GEMM and gather-sum launch
                                        of this call, collected by rr_plan_timing_take (costs ~5 us of stream time per launch:
                                        not for timed runs; ABI revision 8) */
       RR_PLAN_WGRAD_EARLY = 256 };  /* rr_reaction_backward only: the W_h weight gradient of a message-passing layer is issued in
                                        front of that layer's input-gradient GEMM instead of behind it (both read the same dZ).
                                        Same kernels, same order on the weight-gradient stream, same workspace layout, same bits;
                                        which order is faster depends on the workload (-1.3 % ... +2.4 % on the step over the four
                                        BASELINE configurations, profiles/r05_experiments.txt item 19): a host may time both and
                                        keep the better one, as reactranker_amd.functions.WgradOrder does (ABI revision 8) */

typedef struct rr_step {
  rr_graph p, r, u;
  int mode;
  const int32_t* amap;    const int32_t* amap_t;  int amap_t_cols;    /* RR_STEP_DEDUP */
  const int32_t* bmap;    const int32_t* bmap_t;  int bmap_t_cols;    /* RR_STEP_PREFIX */
  const float* feat;      int F;                  /* add_features [M, F] or NULL */
  float drop_p;           uint64_t seed;          /* dropout rate in effect (0 in eval mode) and the step's stream seed */
  void* workspace;        size_t workspace_bytes; /* >= rr_reaction_workspace_bytes(model, step); kept until backward */
  float* out;                                     /* [M, ffn[n_ffn-1].out] */
} rr_step;

enum { RR_G_ENC_WI = 0, RR_G_ENC_WH, RR_G_ENC_WO, RR_G_DIF_WI, RR_G_DIF_WH, RR_G_DIF_WO, RR_G_FFN0 };
typedef struct rr_grads {             /* gradient buffers, same shapes / row strides as the parameters (dense: ld = in) */
  float* w[RR_G_FFN0 + RR_MAX_FFN];
  float* b[RR_G_FFN0 + RR_MAX_FFN];   /* NULL where the layer has no bias */
} rr_grads;

/* sizeof of the four plan structs as compiled, so a binding can verify its layout. */
void rr_abi_plan_struct_sizes(size_t* graph, size_t* model, size_t* step, size_t* grads);
size_t rr_reaction_workspace_bytes(const rr_model* model, const rr_step* step);
int rr_reaction_forward(const rr_model* model, const rr_step* step, int flags, rr_stream_t stream);
int rr_reaction_backward(const rr_model* model, const rr_step* step, const float* dout, const rr_grads* grads, int flags,
                         rr_stream_t stream);

/* Per-launch durations of the heavy kernels INSIDE a step plan (RR_PLAN_TIME): what bench.py's roofline object is made of, measured
 * on the path `value` is measured on.  One record per rr_linear_f32 launch on the split path (M >= 8192) and per gather-sum launch. */
typedef struct rr_plan_timing {
  int kind;                /* 0 = rr_linear_f32 (split GEMM), 1 = rr_gather_sum_f32 / _padrow / _amax, 2 = rr_gather_sum_epi_f32 */
  int mode;                /* GEMM: 0 plain, 1 subtract operand, 2 f32 mask, 3 sign-bit mask */
  int64_t M;               /* GEMM rows / gather destination rows */
  int64_t n_src;           /* gather: source rows */
  int N, k1, k2;           /* GEMM: columns, segment widths; gather: N = H, k1 = table width K, k2 = addends */
  int residual, c_pre, dz_out, bits_out, bits_in, mask;   /* which optional operands / outputs the launch carried (0 / 1) */
  float us;                /* HIP-event time around the launch on its stream */
} rr_plan_timing;
/* Waits for the recorded launches of this process, copies up to max_n records (oldest first) and forgets them all; returns the count. */
int rr_plan_timing_take(rr_plan_timing* out, int max_n);
/* Which launches carry events under RR_PLAN_TIME: bit k of `kinds` = launches of kind k, bit m of `modes` = GEMMs of mode m (default:
 * all).  Every event pair costs its stream ~5 us and loosens the overlap between the streams: timing ONE kernel key at a time keeps a
 * step within ~1 % of an untimed one, and its durations at what rocprofv3 --kernel-trace reports for the untimed step. */
int rr_plan_timing_select(int kinds, int modes);

/* Where a step keeps an activation inside its workspace (ABI revision 6) - valid from rr_reaction_forward until the
 * workspace is reused; nothing is launched.  For hosts that want the encoder's outputs (atom hiddens, reaction vectors:
 * `return_atom_hiddens` / `vecs` of models/mpn.py:61-108, 224-238) without a second forward, and for tests that need the
 * ReLU gates this arithmetic took.  `flags` as passed to rr_reaction_forward.  *ptr = NULL when the step has no such tensor
 * (e.g. RR_SAVED_R_MSG index 0 in RR_STEP_PREFIX mode, where the first message exists per DISTINCT reactant only). */
enum { RR_SAVED_R_MSG = 0,    /* reactant encoder: message after iteration `index` (0 = relu(W_i f_bonds)) [r.nB, H], post-dropout */
       RR_SAVED_R_H = 1,      /* reactant atom hiddens [r.nA, H] */
       RR_SAVED_P_MSG = 2, RR_SAVED_P_H = 3,            /* the same for the product encoder */
       RR_SAVED_D_MSG = 4,    /* diff encoder: message after iteration `index` [p.nA, H] */
       RR_SAVED_D_HID = 5,    /* diff encoder atom hiddens [p.nA, H] */
       RR_SAVED_VECS = 6,     /* readout + add_features (after the FFN's input dropout) [M, ld] */
       RR_SAVED_FFN_H = 7,    /* FFN hidden layer `index` (1 .. n_ffn-1) [M, ld] */
       RR_SAVED_R_MSG0_U = 8, /* RR_STEP_PREFIX: relu(W_i f_bonds) of the distinct reactants [u.nB, H] */
       RR_SAVED_R_Z1_U = 9 }; /* RR_STEP_PREFIX: their first W_h layer before dropout [u.nB, H] */
int rr_reaction_saved_f32(const rr_model* model, const rr_step* step, int flags, int which, int index,
                          const float** ptr, int64_t* rows, int64_t* ld);

/* ------------------------------------------------------------------ data-parallel gradient exchange (RCCL) --- */
/* Queries are independent: one process per GPU, whole queries per rank, identical replicas, and ONE sum all-reduce of the
 * gradient buffers per optimizer step (the reference itself is single-process).  `comm` is an RCCL
 * communicator (ncclComm_t): either one the host created with the RCCL it links - pass it as is - or one made by
 * rr_comm_init_rank below.  RCCL is resolved at run time (symbols already in the process first, then librccl.so.1), so this
 * library has no link-time dependency on it; RR_ERR_UNSUPPORTED when no RCCL can be found.
 *   rr_allreduce_f32: buf[0:n] <- scale * sum over ranks of buf[0:n], in place, enqueued on `stream` (the all-reduce, then
 *   one scaling pass unless scale == 1).  With equal shards pass scale = 1 / n_ranks; with ragged shards scale the local
 *   gradient by local_count / global_count first (reactranker_amd.dp.loss_weight names the count per loss) and pass 1.
 *   rr_comm_unique_id: rank 0 fills `id` (RR_COMM_ID_BYTES host bytes) and hands it to the other ranks by any host channel;
 *   rr_comm_init_rank: collective over all ranks, each on its own current device. */
typedef void* rr_comm_t;
#define RR_COMM_ID_BYTES 128
int rr_comm_backend(void);   /* how RCCL was resolved: 0 not found, 1 symbols already in the process, 2 dlopen("librccl.so.1") */
int rr_comm_unique_id(void* id);
int rr_comm_init_rank(rr_comm_t* comm, int n_ranks, const void* id, int rank);
int rr_comm_destroy(rr_comm_t comm);
int rr_allreduce_f32(float* buf, int64_t n, float scale, rr_comm_t comm, rr_stream_t stream);
/* The same result as two collectives (ABI revision 8): reduce-scatter + all-gather, in place, over the largest prefix of the
 * bucket that divides by the communicator's rank count, the remaining < n_ranks elements through an all-reduce; then the
 * scale.  For the 3.16 / 12.1 MB buckets on xGMI's point-to-point links; which form is faster is a
 * measurement for an 8-GPU node - rr_allreduce_f32 stays the default.  With 2 ranks the bits equal rr_allreduce_f32's. */
int rr_allreduce_rsag_f32(float* buf, int64_t n, float scale, rr_comm_t comm, rr_stream_t stream);

/* Bond-to-bond backward table (HOST pointers), derived from the tables above.  The adjoint of
 *   message[b] = a_message[b2a[b]] - message[b2revb[b]],  a_message[a] = sum_k message[a2b[a,k]]   (models/mpn.py:89-92)
 * is  d message[b] = sum of d m_in over the bonds leaving atom target(b), minus d m_in[rev(b)]; rev(b) is one of those
 * bonds, so it is a plain gather-sum over  b2b_t[nB, Kb]  (the bonds leaving target(b) except rev(b); pads -1;
 * Kb >= max(1, K-1); row 0 all -1) - one pass instead of gather-sum + gather-diff.  The padding row's adjoint is the
 * weighted column sum  sum_b npad_b[b] * d m_in[b]  with  npad_b[b] = npad[b2a[b]]  (npad_b[0] = K - 1). */
int rr_derive_bond_tables(const int32_t* a2b_rev_t, const int32_t* b2t, const int32_t* b2revb, const int32_t* b2a,
                          const float* npad, int64_t nA, int64_t nB, int K, int Kb,
                          int32_t* b2b_t, float* npad_b);

#ifdef __cplusplus
}
#endif
#endif /* REACTRANKER_HIP_H */
