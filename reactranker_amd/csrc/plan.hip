// Whole-model step plans: ReactionModel.forward (reference models/base_model.py:150-171) and its explicit backward as
// ONE C-ABI call each.  The kernels are the library's own entry points (rr_linear_f32, rr_gather_sum_f32, ...); what
// lives here is the host-side orchestration the Python mirror used to do launch by launch (~140 ctypes calls and
// ~4 ms of interpreter time per training step): operand wiring, dropout stream seeds, the weight-gradient stream and
// the reactant-encoder stream with their events, and a bump allocator over ONE caller-provided workspace whose layout
// is a pure function of (model, step) - so the backward call re-derives every saved activation's address and the
// library keeps no state between calls.  A non-Python host drives training with these two calls plus a loss kernel.
//
// Scope: hidden sizes with H % 4 == 0 (every operand 16-byte addressable: the straight-line kernels); other shapes
// stay on the per-op entry points.  Numerics, kernel order and dropout streams are exactly those of
// reactranker_amd/functions.py (mpn_forward / mpn_forward_shared / mpndiff_forward / ffn_forward and their
// adjoints), which the parity tests compare bit for bit.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "rr_common.h"

namespace {

constexpr int ATOM_FDIM = 61;

inline uint64_t site_seed(uint64_t seed, uint64_t site) {
  return seed * 0x9E3779B97F4A7C15ull + site * 0xD1B54A32D192ED03ull;
}
inline int64_t r4(int64_t n) { return (n + 3) / 4 * 4; }

struct Arena {                       // bump allocator over the caller's workspace (or a dry run that only measures)
  char* base;
  size_t off, cap;
  bool overflow;
  float* f(int64_t rows, int64_t cols) {
    const size_t bytes = (static_cast<size_t>(rows < 1 ? 1 : rows) * static_cast<size_t>(cols) * sizeof(float) + 255) & ~size_t(255);
    const size_t at = off;
    off += bytes;
    if (base == nullptr) return nullptr;
    if (off > cap) { overflow = true; return reinterpret_cast<float*>(base); }
    return reinterpret_cast<float*>(base + at);
  }
};

struct Streams {
  hipStream_t main, side, aux;
};

// per-device side / aux streams (created once; non-blocking so they never synchronise with the null stream)
struct DevStreams { bool init; hipStream_t side, aux; };
DevStreams g_streams[64];
std::mutex g_streams_mu;             // two host threads may issue their first plan at the same time

int get_streams(hipStream_t main, Streams* s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return RR_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(g_streams_mu);
  DevStreams& d = g_streams[dev];
  if (!d.init) {
    if (hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking) != hipSuccess) return RR_ERR_LAUNCH;
    if (hipStreamCreateWithFlags(&d.aux, hipStreamNonBlocking) != hipSuccess) return RR_ERR_LAUNCH;
    d.init = true;
  }
  s->main = main;
  s->side = d.side;
  s->aux = d.aux;
  return RR_OK;
}

// `waiter` waits for everything enqueued on `signaller` so far
int stream_wait(hipStream_t waiter, hipStream_t signaller) {
  if (waiter == signaller) return RR_OK;
  hipEvent_t e;
  // The event orders two streams of ONE device: the producing kernel's own end-of-kernel release and the consumer's acquire
  // (agent scope, as between any two kernels of one stream) already make the data visible, so the marker does not need the
  // system-scope fence a host-visible event carries - its cache writeback / invalidate sits on the RECORDING stream, in front
  // of that stream's next kernel: 5.0 us per event with the fence, 2.4 us without (tools/evgap.cpp), ~20 events per
  // training step, -0.9 % on the step (profiles/r05_experiments.txt item 1).  RR_EV_FENCE=1 restores the default event.
  unsigned fl = hipEventDisableTiming;
  if (!getenv("RR_EV_FENCE")) fl |= hipEventDisableSystemFence;
  if (hipEventCreateWithFlags(&e, fl) != hipSuccess) return RR_ERR_LAUNCH;
  int st = RR_OK;
  if (hipEventRecord(e, signaller) != hipSuccess || hipStreamWaitEvent(waiter, e, 0) != hipSuccess) st = RR_ERR_LAUNCH;
  (void)hipEventDestroy(e);           // destruction is deferred by the runtime until the event has completed
  return st;
}

// packed weight [rows, r16(k1)+r16(k2)] of W[:, c0 : c0+k1+k2] (transpose = 0) or of its transpose
struct Packed { const float* w; int64_t ld; int mode; };   // mode = rr_linear_args.w_packed (1 f32 layout, 2 bf16 terms)

// Two-f16-term GEMMs (RR_PLAN_F16X2_GEMM) need an upper bound of every operand tensor's magnitude.  The plan keeps one
// magnitude slot (RR_AMAX_FLOATS floats, include/reactranker_hip.h) per tensor in a block of the workspace, zeroed at the
// start of the forward / of the backward's own part.  A tensor's slot is filled by the kernel that PRODUCES it where that
// kernel can (amax_claim: the split GEMM's epilogue, the gather kernels, the readout's adjoint - the producer maxes what it
// stores into the slot),
// otherwise by one rr_amax_f32 pass on the stream of the tensor's first consumer (amax_of: the step's input features and the
// outputs of the few kernels without a magnitude output), and serves every later consumer - a forward activation's slot
// serves its weight gradient in the backward: the workspace is kept.  Tensors are known by address: every arena allocation
// and the step's input arrays are recorded with their shape.
constexpr int MAX_TENSORS = 768, MAX_AMAX = 256;
struct TensorRec { const float* p; int64_t rows, cols, ld; float* amax; };

struct Ctx {
  bool launch;                        // false: layout pass only (no kernel is enqueued)
  int status;
  Arena ar;
  bool f16;                           // split GEMMs on two f16 terms (needs split)
  float* amax_base;                   // MAX_AMAX magnitude slots (null in a measuring pass)
  int namax, ntr;
  TensorRec tr[MAX_TENSORS];
  float* alloc(int64_t rows, int64_t cols) {            // arena tensor, remembered for amax_of()
    float* p = ar.f(rows, cols);
    if (f16 && p != nullptr && !ar.overflow) reg(p, rows, cols, cols);
    return p;
  }
  void reg(const float* p, int64_t rows, int64_t cols, int64_t ld) {
    if (!f16 || p == nullptr || ar.base == nullptr) return;
    for (int i = 0; i < ntr; ++i)
      if (tr[i].p == p) return;
    if (ntr == MAX_TENSORS) { fail(RR_ERR_UNSUPPORTED); return; }
    tr[ntr].p = p; tr[ntr].rows = rows; tr[ntr].cols = cols; tr[ntr].ld = ld; tr[ntr].amax = nullptr;
    ++ntr;
  }
  Streams s;
  bool use_side, use_aux, aux_bwd;
  bool train;                         // RR_PLAN_TRAIN: the forward packs the backward's transposed weights too
  bool ffn_chain;                     // the FFN head and its input-gradient chain as one launch each (rr_ffn_chain_f32)
  bool timing;                        // RR_PLAN_TIME: events around the heavy launches
  bool gather_multi;                  // sums of per-copy tensors ride on the gather over the copies (rr_gather_sum_multi_f32)
  bool wgrad_early;                   // RR_PLAN_WGRAD_EARLY: W_h weight gradients issued in front of the dX GEMM of their layer
  bool tail_pass;                     // the encoder pass being enqueued is the last one of the backward call (see wgrad)
  bool side_joined;                   // the chain has waited for the side stream and nothing was put there since
  hipStream_t cur;                    // stream of the backward chain being enqueued (main, or aux for the reactant pass)
  bool split;                         // encoder GEMMs on the bf16 matrix core (three exact bf16 terms per f32 operand)
  rr_pack_desc pq[RR_MAX_PACK];       // weight packs waiting for flush_packs()
  int npq;
  void fail(int st) { if (status == RR_OK && st != RR_OK) status = st; }
};

#define RR_TRY(ctx, expr)                                                                         \
  do {                                                                                            \
    if ((ctx).launch && (ctx).status == RR_OK) {                                                  \
      (ctx).fail(expr);                                                                           \
      if ((ctx).status != RR_OK && getenv("RR_PLAN_DEBUG"))                                       \
        fprintf(stderr, "[rr plan] status %d at plan.hip:%d: %s\n", (ctx).status, __LINE__, #expr); \
    }                                                                                             \
  } while (0)

// the magnitude slot of tensor t; the first request enqueues its pass on `st` (where t's producer ran or was waited for)
const float* amax_of(Ctx& c, const float* t, hipStream_t st) {
  if (!c.f16 || t == nullptr || c.ar.base == nullptr) return nullptr;
  for (int i = 0; i < c.ntr; ++i) {
    TensorRec& r = c.tr[i];
    if (r.p != t) continue;
    if (r.amax == nullptr) {
      if (c.namax == MAX_AMAX) { c.fail(RR_ERR_UNSUPPORTED); return nullptr; }
      r.amax = c.amax_base + static_cast<size_t>(c.namax++) * RR_AMAX_FLOATS;
      RR_TRY(c, rr_amax_f32(r.p, r.rows, static_cast<int>(r.cols), r.ld, r.amax, st));
    }
    return r.amax;
  }
  if (getenv("RR_PLAN_DEBUG")) fprintf(stderr, "[rr plan] amax_of: unknown tensor %p\n", static_cast<const void*>(t));
  c.fail(RR_ERR_ARG);
  return nullptr;
}

// the slot of a tensor whose PRODUCER maxes its magnitude in as it stores (no pass); null when the mode does not need one
float* amax_claim(Ctx& c, const float* t) {
  if (!c.f16 || t == nullptr || c.ar.base == nullptr) return nullptr;
  for (int i = 0; i < c.ntr; ++i) {
    TensorRec& r = c.tr[i];
    if (r.p != t) continue;
    if (r.amax == nullptr) {
      if (c.namax == MAX_AMAX) { c.fail(RR_ERR_UNSUPPORTED); return nullptr; }
      r.amax = c.amax_base + static_cast<size_t>(c.namax++) * RR_AMAX_FLOATS;
    }
    return r.amax;
  }
  return nullptr;
}

// A tensor that is about to be modified IN PLACE by a kernel without a magnitude output (axpby, the accumulating ReLU
// backward): a slot claimed before that write would bound only what the producer stored - the f16 scale leaves 2x-4x of
// headroom, beyond it the split overflows to inf / NaN.  Such a tensor must reach its consumers WITHOUT a slot, so that
// amax_of() runs its pass after the last write; a plan that violates this fails here instead of training on a stale bound.
// (Since the shared-prefix tail forms d input in one gather epilogue no plan writes in place any more: the guard stays for
// whoever adds such a kernel.)
[[maybe_unused]] void inplace_write(Ctx& c, const float* t) {
  if (!c.f16 || t == nullptr || c.ar.base == nullptr) return;
  for (int i = 0; i < c.ntr; ++i) {
    if (c.tr[i].p != t || c.tr[i].amax == nullptr) continue;
    if (getenv("RR_PLAN_DEBUG")) fprintf(stderr, "[rr plan] in-place write to %p after its magnitude slot was claimed\n", static_cast<const void*>(t));
    c.fail(RR_ERR_ARG);
  }
}

// RR_PLAN_TIME: HIP events around the heavy launches of a plan call (rr_plan_timing_take reads them)
struct TimingRec { hipEvent_t e0, e1; rr_plan_timing info; };
std::vector<TimingRec> g_timing;
std::mutex g_timing_mu;
int g_timing_kinds = 7, g_timing_modes = 15;             // rr_plan_timing_select: which launches carry events

struct Timed {                       // records e0 now, e1 at destruction - both on `st`
  bool on;
  hipStream_t st;
  TimingRec r;
  Timed(Ctx& c, hipStream_t st_, const rr_plan_timing& info, bool want = true);
  ~Timed() {
    if (!on) return;
    if (hipEventRecord(r.e1, st) != hipSuccess) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); return; }
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_timing.push_back(r);
  }
};

void flush_packs(Ctx& c, hipStream_t st) {
  if (c.npq > 0) RR_TRY(c, rr_pack_weights_f32(c.pq, c.npq, st));
  c.npq = 0;
}

Timed::Timed(Ctx& c, hipStream_t st_, const rr_plan_timing& info, bool want) : on(false), st(st_) {
  if (!want || !c.timing || !c.launch || c.status != RR_OK) return;
  if (!((g_timing_kinds >> info.kind) & 1) || (info.kind == 0 && !((g_timing_modes >> info.mode) & 1))) return;
  r.info = info;
  // (no system-scope fence with the markers: they only have to order and stamp - see stream_wait)
  if (hipEventCreateWithFlags(&r.e0, hipEventDisableSystemFence) != hipSuccess) return;
  if (hipEventCreateWithFlags(&r.e1, hipEventDisableSystemFence) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
  if (hipEventRecord(r.e0, st) != hipSuccess) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); return; }
  on = true;
}

// big = the GEMM runs over atoms / bonds (the FFN head runs over molecules: a few thousand rows, launch-bound either way)
Packed pack(Ctx& c, const rr_linear_w& L, int transpose, int rows, int c0, int k1, int k2, hipStream_t st, bool big = true) {
  Packed p;
  const bool split = c.split && big && rows <= 608 && rows % 4 == 0;
  float* dst;
  if (split) {
    p.ld = 0;
    p.mode = c.f16 ? 3 : 2;
    dst = c.alloc(1, static_cast<int64_t>(rr_split_weight_bytes(rows, k1, k2) / 4));
  } else {
    p.ld = rr_packed_weight_ld(k1, k2);
    p.mode = 1;
    dst = c.alloc(rows, p.ld);
  }
  p.w = dst;
  if (c.npq == RR_MAX_PACK) flush_packs(c, st);
  rr_pack_desc& q = c.pq[c.npq++];
  q.src = L.w; q.ld_src = L.ldw; q.transpose = transpose; q.rows = rows; q.c0 = c0; q.k1 = k1; q.k2 = k2; q.dst = dst;
  q.split = split ? (c.f16 ? 2 : 1) : 0;
  return p;
}

// sign-bit image of an [rows, N] activation produced by a split GEMM (null: the consumer reads the f32 tensor instead)
uint8_t* mask_bits(Ctx& c, const Packed& producer, int64_t rows, int N) {
  if (producer.mode < 2) return nullptr;
  return reinterpret_cast<uint8_t*>(c.alloc(rows, rr_mask_bits_row_bytes(N) / 4));
}

rr_linear_args LA(int64_t M, int N) {
  rr_linear_args a;
  memset(&a, 0, sizeof(a));
  a.M = M;
  a.N = N;
  a.mask_scale = 1.0f;
  a.w_packed = 1;
  return a;
}

// rr_linear_f32 with the operand bounds of the two-f16-term path filled in
void lin(Ctx& c, rr_linear_args& a, hipStream_t st) {
  if (a.w_packed == 3) {
    if (a.k1 > 0) a.a1_amax = amax_of(c, a.a1, st);
    if (a.a1_sub) a.a1_sub_amax = amax_of(c, a.a1_sub, st);
    if (a.k2 > 0) a.a2_amax = amax_of(c, a.a2, st);
    a.c_amax_out = amax_claim(c, a.c);                  // what the next GEMM needs of this one's outputs
    if (a.dz_out) a.dz_amax_out = amax_claim(c, a.dz_out);
  }
  rr_plan_timing ti;
  memset(&ti, 0, sizeof(ti));
  ti.kind = 0; ti.M = a.M; ti.N = a.N; ti.k1 = a.k1; ti.k2 = a.k2;
  ti.mode = a.a_mask_bits ? 3 : (a.a_mask ? 2 : (a.a1_sub ? 1 : 0));
  ti.residual = a.residual != nullptr; ti.c_pre = a.c_pre != nullptr; ti.dz_out = a.dz_out != nullptr;
  ti.bits_out = a.mask_bits_out != nullptr; ti.bits_in = a.a_mask_bits != nullptr; ti.mask = a.a_mask != nullptr;
  Timed t(c, st, ti, a.w_packed >= 2 && a.M >= 8192);      // (only the split GEMMs over atoms / bonds are timed)
  RR_TRY(c, rr_linear_f32(&a, st));
}

// Packs are queued and issued together (rr_pack_weights_f32: one launch for up to 16 weights) by flush_packs().

void set_w(rr_linear_args& a, const Packed& p) {
  a.w = p.w; a.ldw = p.ld; a.w_packed = p.mode;
  if (p.mode < 2) a.a_mask_bits = nullptr;            // only the split GEMM reads sign-bit masks (a_mask stays set)
}

// claim = false: `out` is modified in place afterwards - no slot is claimed for it (see inplace_write)
void gather_sum(Ctx& c, const float* src, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out, int K, int H,
                float* out, int64_t ld_out, hipStream_t st, const float* part = nullptr, int64_t n_part = 0, int64_t ld_part = 0,
                bool claim = true) {
  rr_plan_timing ti;
  memset(&ti, 0, sizeof(ti));
  ti.kind = 1; ti.M = n_out; ti.n_src = n_src; ti.N = H; ti.k1 = K;
  Timed t(c, st, ti, n_out >= 8192);
  if (c.f16 && claim) RR_TRY(c, rr_gather_sum_amax_f32(src, n_src, ld_src, idx, n_out, K, H, part, n_part, ld_part, out, ld_out, amax_claim(c, out), st));
  else if (part) RR_TRY(c, rr_gather_sum_padrow_f32(src, n_src, ld_src, idx, n_out, K, H, part, n_part, ld_part, out, ld_out, st));
  else RR_TRY(c, rr_gather_sum_f32(src, n_src, ld_src, idx, n_out, K, H, out, ld_out, st));
}

// out[r] = sum over idx of (srcs[0] + srcs[1] + ...) (rr_gather_sum_multi_f32); `out` is modified in place afterwards, no slot
void gather_multi(Ctx& c, const float* const* srcs, int n_srcs, int64_t n_src, int64_t ld_src, const int32_t* idx, int64_t n_out,
                  int K, int H, float* out, int64_t ld_out, hipStream_t st) {
  if (!c.launch) return;
  RR_TRY(c, rr_gather_sum_multi_f32(srcs, n_srcs, n_src, ld_src, idx, n_out, K, H, out, ld_out, st));
}

// gather-sum whose epilogue applies the ReLU / dropout mask of the activation `y` (sign bits when a split GEMM wrote
// them, the f32 tensor otherwise; y == nullptr: no mask) and adds `n_adds` row-aligned tensors (rr_gather_sum_epi_f32)
void gather_epi(Ctx& c, const float* src, int64_t n_src, const int32_t* idx, int64_t n_out, int K, int H, float* out,
                hipStream_t st, const float* part, int64_t n_part, bool masked, const float* y, const uint8_t* y_bits,
                float scale, const float* const* adds, int n_adds) {
  rr_gather_epi e;
  memset(&e, 0, sizeof(e));
  if (masked) { e.mask = y; e.ld_mask = H; e.mask_bits = y_bits; }
  e.mask_scale = scale;
  e.n_adds = n_adds;
  e.ld_add = H;
  for (int j = 0; j < n_adds; ++j) e.adds[j] = adds[j];
  e.amax_out = amax_claim(c, out);
  rr_plan_timing ti;
  memset(&ti, 0, sizeof(ti));
  ti.kind = 2; ti.M = n_out; ti.n_src = n_src; ti.N = H; ti.k1 = K; ti.k2 = n_adds;
  ti.mask = masked ? 1 : 0; ti.bits_in = (masked && y_bits != nullptr) ? 1 : 0;
  Timed t(c, st, ti, n_out >= 8192);
  RR_TRY(c, rr_gather_sum_epi_f32(src, n_src, H, idx, n_out, K, H, part, n_part, r4(H), &e, out, H, st));
}

// weight gradient on the side stream: waits for the main stream's work so far (its operands), returns immediately
void wgrad(Ctx& c, rr_wgrad_args& a, bool tail = false) {
  const size_t wb = rr_linear_wgrad_workspace_bytes(a.M, a.N, a.k1 + a.k2);
  a.workspace = c.alloc(1, static_cast<int64_t>((wb + 3) / 4));
  a.workspace_bytes = wb;
  a.split = (c.split && a.M >= 8192) ? (c.f16 ? 2 : 1) : 0;   // the FFN head (one row per molecule) stays on the f32 matrix core
  if (a.split == 2) {                                   // bounds on the chain's stream, BEFORE the side stream waits for it
    a.dy_amax = amax_of(c, a.dy, c.cur);
    if (a.k1 > 0) a.x1_amax = amax_of(c, a.x1, c.cur);
    if (a.x1_sub) a.x1_sub_amax = amax_of(c, a.x1_sub, c.cur);
    if (a.k2 > 0) a.x2_amax = amax_of(c, a.x2, c.cur);
  }
  if (!c.launch || c.status != RR_OK) return;
  // tail: the last weight gradient of a backward call - nothing is left on the chain for it to overlap with, the optimizer waits
  // for it.  It runs ON the chain, behind everything the side stream holds (same order of accumulation into dw), instead of one
  // stream hop to the side stream and one back (~11 us each against ~5 us between two kernels of one stream).
  const bool join = tail && c.use_side && c.tail_pass && c.cur == c.s.main && !getenv("RR_NO_TAIL_JOIN");
  hipStream_t st = (c.use_side && !join) ? c.s.side : c.cur;
  if (join) {
    c.fail(stream_wait(c.cur, c.s.side));
    c.side_joined = true;
  } else if (c.use_side) {
    c.fail(stream_wait(c.s.side, c.cur));
    c.side_joined = false;
  }
  c.fail(rr_linear_wgrad_f32(&a, st));
  if (c.status != RR_OK && getenv("RR_PLAN_DEBUG"))
    fprintf(stderr, "[rr plan] wgrad status %d: M %lld N %d k1 %d k2 %d ws %zu\n", c.status, (long long)a.M, a.N, a.k1, a.k2, wb);
}

rr_wgrad_args WA(int64_t M, int N, const float* dy, int64_t ld_dy, float* dw, int64_t ld_dw, float* dbias, int accumulate) {
  rr_wgrad_args a;
  memset(&a, 0, sizeof(a));
  a.M = M; a.N = N; a.dy = dy; a.ld_dy = ld_dy; a.dw = dw; a.ld_dw = ld_dw; a.dbias = dbias; a.accumulate = accumulate;
  a.mask_scale = 1.0f;
  return a;
}

// ------------------------------------------------------------------------------------------------ saved state
constexpr int MAXD = 16;             // message-passing depth supported by a plan

struct EncSaved {                    // mpn_forward / mpn_forward_shared
  uint8_t* bits[MAXD];               // sign bits of msgs[i] where a split GEMM produced it (rr_linear_args.mask_bits_out), else null
  uint8_t* bits_h;
  float* msgs[MAXD];                 // [nB, H] each (shared mode: msgs[0] unused)
  float* amsgs[MAXD];                // [nA, H]
  float* a_last;
  float* h;                          // [nA, H] atom hiddens (output)
  float *msg0_u, *a0_u;              // shared prefix (distinct molecules)
  float* z1_u;                       // ... its pre-dropout output, and the stream the copies' masks were drawn from
  uint64_t seed0;
};
struct DiffSaved {
  uint8_t* bits[MAXD];
  uint8_t* bits_hid;
  float* msgs[MAXD];
  float* amsgs[MAXD];
  float* a_last;
  float* hid;
  float* vecs; int64_t ld_vecs;      // [M, r4(H+F)]
};
struct FfnSaved {
  float* hs[RR_MAX_FFN + 1];         // hs[0] = vecs
  int64_t ld_hs[RR_MAX_FFN + 1];
  float* raw;                        // [M, n_out]
};
struct PackedW {
  Packed enc_wi, enc_wh, enc_wo, dif_wi, dif_wh, dif_wo, ffn[RR_MAX_FFN];
};
struct PackedT {                     // transposed weights of the input-gradient GEMMs (dX = dZ * W): packed in one launch
  Packed ffn[RR_MAX_FFN];
  Packed dif_wo_x, dif_wo_a, dif_wh, dif_wi, enc_wh, enc_wo;
};
struct Plan {
  PackedW pk;
  PackedT T;                         // filled by the forward under RR_PLAN_TRAIN, by the backward otherwise
  EncSaved r, p;
  DiffSaved d;
  FfnSaved f;
  size_t fwd_end;
};

// ------------------------------------------------------------------------------------------------ forward pieces
// MPN.forward, return_atom_hiddens=True (models/mpn.py:61-108)
void mpn_forward(Ctx& c, const rr_model& m, const rr_graph& g, const PackedW& pk, float p, uint64_t seed, EncSaved& S,
                 hipStream_t st) {
  const int H = m.H, depth = m.depth, FB = m.bond_fdim;
  float* inp = c.alloc(g.nB, H);
  S.msgs[0] = c.alloc(g.nB, H);
  {
    rr_linear_args a = LA(g.nB, H);
    a.a1 = g.f_bonds; a.lda1 = g.ld_fb; a.k1 = FB;
    set_w(a, pk.enc_wi); a.bias = m.enc_wi.b; a.act = RR_ACT_RELU;
    a.c = S.msgs[0]; a.ldc = H; a.c_pre = inp; a.ld_pre = H;
    S.bits[0] = mask_bits(c, pk.enc_wi, g.nB, H); a.mask_bits_out = S.bits[0];           // relu'(input) for the backward
    lin(c, a, st);                                                              // :80-81
  }
  for (int it = 0; it < depth - 1; ++it) {                                              // :84
    S.amsgs[it] = c.alloc(g.nA, H);
    gather_sum(c, S.msgs[it], g.nB, H, g.a2b, g.nA, g.K, H, S.amsgs[it], H, st);       // :89-90
    S.msgs[it + 1] = c.alloc(g.nB, H);
    rr_linear_args a = LA(g.nB, H);
    a.a1 = S.amsgs[it]; a.lda1 = H; a.k1 = H; a.a1_idx = g.b2a;
    a.a1_sub = S.msgs[it]; a.lda1_sub = H; a.a1_sub_idx = g.b2revb;
    set_w(a, pk.enc_wh); a.bias = m.enc_wh.b; a.residual = inp; a.ldr = H; a.act = RR_ACT_RELU;
    a.drop_p = p; a.drop_seed = site_seed(seed, it);
    a.c = S.msgs[it + 1]; a.ldc = H;
    S.bits[it + 1] = mask_bits(c, pk.enc_wh, g.nB, H); a.mask_bits_out = S.bits[it + 1];
    lin(c, a, st);                                                              // :91-97
  }
  S.a_last = c.alloc(g.nA, H);
  gather_sum(c, S.msgs[depth - 1], g.nB, H, g.a2b, g.nA, g.K, H, S.a_last, H, st);     // :101-102
  S.h = c.alloc(g.nA, H);
  rr_linear_args a = LA(g.nA, H);
  a.a1 = g.f_atoms; a.lda1 = g.ld_fa; a.k1 = m.atom_fdim; a.a2 = S.a_last; a.lda2 = H; a.k2 = H;
  set_w(a, pk.enc_wo); a.bias = m.enc_wo.b; a.act = RR_ACT_RELU; a.drop_p = p; a.drop_seed = site_seed(seed, 1000);
  a.c = S.h; a.ldc = H;
  S.bits_h = mask_bits(c, pk.enc_wo, g.nA, H); a.mask_bits_out = S.bits_h;
  lin(c, a, st);                                                              // :103-105
}

// the same for a batch whose molecules repeat: the deterministic prefix runs once per distinct molecule
void mpn_forward_shared(Ctx& c, const rr_model& m, const rr_graph& gu, const rr_graph& g, const int32_t* bmap,
                        const PackedW& pk, float p, uint64_t seed, EncSaved& S, hipStream_t st) {
  const int H = m.H, depth = m.depth, FB = m.bond_fdim;
  float* inp_u = c.alloc(gu.nB, H);
  S.msg0_u = c.alloc(gu.nB, H);
  {
    rr_linear_args a = LA(gu.nB, H);
    a.a1 = gu.f_bonds; a.lda1 = gu.ld_fb; a.k1 = FB;
    set_w(a, pk.enc_wi); a.bias = m.enc_wi.b; a.act = RR_ACT_RELU;
    a.c = S.msg0_u; a.ldc = H; a.c_pre = inp_u; a.ld_pre = H;
    lin(c, a, st);
  }
  S.a0_u = c.alloc(gu.nA, H);
  gather_sum(c, S.msg0_u, gu.nB, H, gu.a2b, gu.nA, gu.K, H, S.a0_u, H, st);
  float* z1_u = c.alloc(gu.nB, H);
  {
    rr_linear_args a = LA(gu.nB, H);
    a.a1 = S.a0_u; a.lda1 = H; a.k1 = H; a.a1_idx = gu.b2a;
    a.a1_sub = S.msg0_u; a.lda1_sub = H; a.a1_sub_idx = gu.b2revb;
    set_w(a, pk.enc_wh); a.bias = m.enc_wh.b; a.residual = inp_u; a.ldr = H; a.act = RR_ACT_RELU;
    a.c = z1_u; a.ldc = H;
    lin(c, a, st);                                                              // pre-dropout, shared
  }
  S.msgs[0] = nullptr;
  S.msgs[1] = c.alloc(g.nB, H);
  RR_TRY(c, rr_gather_dropout_amax_f32(z1_u, gu.nB, H, bmap, g.nB, H, p, site_seed(seed, 0), S.msgs[1], H, amax_claim(c, S.msgs[1]), st));   // per-copy masks
  S.z1_u = z1_u;
  S.seed0 = site_seed(seed, 0);
  for (int it = 1; it < depth - 1; ++it) {
    S.amsgs[it] = c.alloc(g.nA, H);
    gather_sum(c, S.msgs[it], g.nB, H, g.a2b, g.nA, g.K, H, S.amsgs[it], H, st);
    S.msgs[it + 1] = c.alloc(g.nB, H);
    rr_linear_args a = LA(g.nB, H);
    a.a1 = S.amsgs[it]; a.lda1 = H; a.k1 = H; a.a1_idx = g.b2a;
    a.a1_sub = S.msgs[it]; a.lda1_sub = H; a.a1_sub_idx = g.b2revb;
    set_w(a, pk.enc_wh); a.bias = m.enc_wh.b; a.residual = inp_u; a.ldr = H; a.residual_idx = bmap; a.act = RR_ACT_RELU;
    a.drop_p = p; a.drop_seed = site_seed(seed, it);
    a.c = S.msgs[it + 1]; a.ldc = H;
    S.bits[it + 1] = mask_bits(c, pk.enc_wh, g.nB, H); a.mask_bits_out = S.bits[it + 1];
    lin(c, a, st);
  }
  S.a_last = c.alloc(g.nA, H);
  gather_sum(c, S.msgs[depth - 1], g.nB, H, g.a2b, g.nA, g.K, H, S.a_last, H, st);
  S.h = c.alloc(g.nA, H);
  rr_linear_args a = LA(g.nA, H);
  a.a1 = g.f_atoms; a.lda1 = g.ld_fa; a.k1 = m.atom_fdim; a.a2 = S.a_last; a.lda2 = H; a.k2 = H;
  set_w(a, pk.enc_wo); a.bias = m.enc_wo.b; a.act = RR_ACT_RELU; a.drop_p = p; a.drop_seed = site_seed(seed, 1000);
  a.c = S.h; a.ldc = H;
  S.bits_h = mask_bits(c, pk.enc_wo, g.nA, H); a.mask_bits_out = S.bits_h;
  lin(c, a, st);
}

// MPNDiff.forward (models/mpn.py:170-240): atom_features = x - x_sub[x_sub_idx]
void mpndiff_forward(Ctx& c, const rr_model& m, const rr_graph& g, const PackedW& pk, float p, uint64_t seed, const float* x,
                     const float* x_sub, const int32_t* x_sub_idx, const float* feat, int F, uint64_t out_seed, DiffSaved& S,
                     hipStream_t st) {
  const int H = m.H, depth = m.diff_depth, FB = m.bond_fdim;
  float* inp = c.alloc(g.nA, H);
  S.msgs[0] = c.alloc(g.nA, H);
  {
    rr_linear_args a = LA(g.nA, H);
    a.a1 = x; a.lda1 = H; a.k1 = H; a.a1_sub = x_sub; a.lda1_sub = H; a.a1_sub_idx = x_sub_idx;
    set_w(a, pk.dif_wi); a.bias = m.dif_wi.b; a.act = RR_ACT_RELU;
    a.drop_p = depth == 0 ? p : 0.f; a.drop_seed = site_seed(seed, 2000);              // :221 (depth 0: dropout(message))
    a.c = S.msgs[0]; a.ldc = H; a.c_pre = inp; a.ld_pre = H;
    S.bits[0] = mask_bits(c, pk.dif_wi, g.nA, H); a.mask_bits_out = S.bits[0];
    lin(c, a, st);                                                              // :194-195
  }
  if (depth > 0) {
    for (int it = 0; it < depth - 1; ++it) {                                            // :199
      S.amsgs[it] = c.alloc(g.nA, H);
      gather_sum(c, S.msgs[it], g.nA, H, g.a2a, g.nA, g.K, H, S.amsgs[it], H, st);     // :201
      S.msgs[it + 1] = c.alloc(g.nA, H);
      rr_linear_args a = LA(g.nA, H);
      a.a1 = S.amsgs[it]; a.lda1 = H; a.k1 = H; a.a2 = g.fb_sum; a.lda2 = g.ld_fbs; a.k2 = FB;
      set_w(a, pk.dif_wh); a.bias = m.dif_wh.b; a.residual = inp; a.ldr = H; a.act = RR_ACT_RELU;
      a.drop_p = p; a.drop_seed = site_seed(seed, 2001 + it);
      a.c = S.msgs[it + 1]; a.ldc = H;
      S.bits[it + 1] = mask_bits(c, pk.dif_wh, g.nA, H); a.mask_bits_out = S.bits[it + 1];
      lin(c, a, st);                                                              // :202-213
    }
    S.a_last = c.alloc(g.nA, H);
    gather_sum(c, S.msgs[depth - 1], g.nA, H, g.a2a, g.nA, g.K, H, S.a_last, H, st);   // :215-216
    S.hid = c.alloc(g.nA, H);
    rr_linear_args a = LA(g.nA, H);
    a.a1 = x; a.lda1 = H; a.k1 = H; a.a1_sub = x_sub; a.lda1_sub = H; a.a1_sub_idx = x_sub_idx;
    a.a2 = S.a_last; a.lda2 = H; a.k2 = H;
    set_w(a, pk.dif_wo); a.bias = m.dif_wo.b; a.act = RR_ACT_RELU; a.drop_p = p; a.drop_seed = site_seed(seed, 3000);
    a.c = S.hid; a.ldc = H;
    S.bits_hid = mask_bits(c, pk.dif_wo, g.nA, H); a.mask_bits_out = S.bits_hid;
    lin(c, a, st);                                                              // :217-219
  } else {
    S.a_last = nullptr;
    S.hid = S.msgs[0];
  }
  S.ld_vecs = r4(H + F);
  S.vecs = c.alloc(g.M, S.ld_vecs);
  RR_TRY(c, rr_segment_mean_fwd_f32(S.hid, H, g.a_scope, g.M, H, feat, F, p, out_seed, S.vecs, S.ld_vecs, st));   // :224-238
}

// FFN.forward after its input dropout (models/base_model.py:32-60).  The whole head is ONE launch where rr_ffn_chain_f32 takes the
// shape (csrc/ffn.hip: bit-identical to the layers issued one by one); otherwise - RR_PLAN_NO_FFN_CHAIN, an odd width, the
// MFMA form of the last layer asked for with RR_NO_ROWDOT - the per-layer launches below.  Same workspace layout either way.
void ffn_forward(Ctx& c, const rr_model& m, const PackedW& pk, int64_t M, float p, uint64_t seed, const float* x, int64_t ldx,
                 float* out, FfnSaved& S, hipStream_t st) {
  S.hs[0] = const_cast<float*>(x);
  S.ld_hs[0] = ldx;
  for (int li = 0; li < m.n_ffn - 1; ++li) {
    S.ld_hs[li + 1] = r4(m.ffn[li].out);
    S.hs[li + 1] = c.alloc(M, S.ld_hs[li + 1]);
  }
  const rr_linear_w& L = m.ffn[m.n_ffn - 1];
  S.raw = m.head == 0 ? out : c.alloc(M, L.out);
  bool chained = false;
  if (c.ffn_chain && m.n_ffn >= 2 && L.out <= 8 && L.in % 4 == 0 && pk.ffn[m.n_ffn - 1].mode == 1 && !getenv("RR_NO_ROWDOT") &&
      c.launch && c.status == RR_OK) {
    rr_ffn_chain_args a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.n_stages = m.n_ffn; a.x = x; a.ldx = ldx; a.drop_p = p; a.mask_scale = 1.0f;
    bool ok = true;
    for (int li = 0; li < m.n_ffn; ++li) {
      rr_ffn_stage& g = a.stage[li];
      const rr_linear_w& Ll = m.ffn[li];
      ok = ok && pk.ffn[li].mode == 1;
      g.w = pk.ffn[li].w; g.ldw = pk.ffn[li].ld; g.bias = Ll.b; g.n_out = Ll.out; g.n_in = Ll.in;
      if (li < m.n_ffn - 1) {
        g.relu = 1; g.dropout = 1; g.drop_seed = site_seed(seed, 4000 + li);
        g.out = S.hs[li + 1]; g.ld_out = S.ld_hs[li + 1];
      } else {
        g.rowdot = 1; g.out = S.raw; g.ld_out = L.out;
      }
    }
    if (ok) {
      const int stc = rr_ffn_chain_f32(&a, st);
      if (stc == RR_OK) chained = true;
      else if (stc != RR_ERR_UNSUPPORTED) c.fail(stc);
    }
  }
  if (!chained) {
    for (int li = 0; li < m.n_ffn - 1; ++li) {
      const rr_linear_w& Ll = m.ffn[li];
      rr_linear_args a = LA(M, Ll.out);
      a.a1 = S.hs[li]; a.lda1 = S.ld_hs[li]; a.k1 = Ll.in;
      set_w(a, pk.ffn[li]); a.bias = Ll.b; a.act = RR_ACT_RELU; a.drop_p = p; a.drop_seed = site_seed(seed, 4000 + li);
      a.c = S.hs[li + 1]; a.ldc = S.ld_hs[li + 1];
      lin(c, a, st);
    }
    rr_linear_args a = LA(M, L.out);
    a.a1 = S.hs[m.n_ffn - 1]; a.lda1 = S.ld_hs[m.n_ffn - 1]; a.k1 = L.in;
    set_w(a, pk.ffn[m.n_ffn - 1]); a.bias = L.b;
    a.c = S.raw; a.ldc = L.out;
    lin(c, a, st);
  }
  if (m.head != 0) RR_TRY(c, rr_head_fwd_f32(S.raw, M, L.out, m.head, out, st));
}

int ffn_dx_rows(const rr_model& m, int li);

// every transposed weight of the backward, queued for ONE pack launch
void pack_transposes(Ctx& c, const rr_model& m, PackedT& T, hipStream_t main) {
  const int H = m.H;
  memset(&T, 0, sizeof(T));
  for (int li = 0; li < m.n_ffn; ++li) T.ffn[li] = pack(c, m.ffn[li], 1, ffn_dx_rows(m, li), 0, m.ffn[li].out, 0, main, false);
  if (m.diff_depth > 0) {
    T.dif_wo_x = pack(c, m.dif_wo, 1, H, 0, H, 0, main);
    T.dif_wo_a = pack(c, m.dif_wo, 1, H, H, H, 0, main);
  }
  if (m.diff_depth > 1) T.dif_wh = pack(c, m.dif_wh, 1, H, 0, H, 0, main);
  T.dif_wi = pack(c, m.dif_wi, 1, H, 0, H, 0, main);
  if (m.depth > 1) T.enc_wh = pack(c, m.enc_wh, 1, H, 0, H, 0, main);
  T.enc_wo = pack(c, m.enc_wo, 1, H, m.atom_fdim, H, 0, main);
}

void forward_all(Ctx& c, const rr_model& m, const rr_step& s, Plan& P) {
  const int H = m.H;
  const float p = s.drop_p;
  hipStream_t main = c.s.main;
  c.ntr = 0;
  c.namax = 0;
  c.amax_base = nullptr;
  if (c.f16) {
    c.amax_base = c.ar.f(MAX_AMAX, RR_AMAX_FLOATS);
    if (c.launch && c.status == RR_OK && hipMemsetAsync(c.amax_base, 0, static_cast<size_t>(MAX_AMAX) * RR_AMAX_FLOATS * sizeof(float), main) != hipSuccess) c.fail(RR_ERR_LAUNCH);
    const rr_graph* gs[3] = {&s.p, &s.r, &s.u};
    for (int i = 0; i < (s.mode == RR_STEP_PREFIX ? 3 : 2); ++i) {
      const rr_graph& g = *gs[i];
      c.reg(g.f_bonds, g.nB, m.bond_fdim, g.ld_fb);
      c.reg(g.f_atoms, g.nA, m.atom_fdim, g.ld_fa);
      c.reg(g.fb_sum, g.nA, m.bond_fdim, g.ld_fbs);
    }
  }
  // packed weights (shared by both encoder passes): once, on the main stream
  P.pk.enc_wi = pack(c, m.enc_wi, 0, H, 0, m.bond_fdim, 0, main);
  if (m.depth > 1) P.pk.enc_wh = pack(c, m.enc_wh, 0, H, 0, H, 0, main);
  P.pk.enc_wo = pack(c, m.enc_wo, 0, H, 0, m.atom_fdim, H, main);
  P.pk.dif_wi = pack(c, m.dif_wi, 0, H, 0, H, 0, main);
  if (m.diff_depth > 1) P.pk.dif_wh = pack(c, m.dif_wh, 0, H, 0, H, m.bond_fdim, main);
  if (m.diff_depth > 0) P.pk.dif_wo = pack(c, m.dif_wo, 0, H, 0, H, H, main);
  for (int li = 0; li < m.n_ffn; ++li) P.pk.ffn[li] = pack(c, m.ffn[li], 0, m.ffn[li].out, 0, m.ffn[li].in, 0, main, false);
  if (c.train) pack_transposes(c, m, P.T, main);         // ... and, when a backward follows, its transposes ride along
  flush_packs(c, main);                                  // every weight of the step in one launch
  hipStream_t rs = c.use_aux ? c.s.aux : main;
  if (c.launch && c.use_aux) c.fail(stream_wait(c.s.aux, main));
  const uint64_t s_r = site_seed(s.seed, 1), s_p = site_seed(s.seed, 2);
  if (s.mode == RR_STEP_PREFIX) mpn_forward_shared(c, m, s.u, s.r, s.bmap, P.pk, p, s_r, P.r, rs);
  else mpn_forward(c, m, s.r, P.pk, p, s_r, P.r, rs);
  mpn_forward(c, m, s.p, P.pk, p, s_p, P.p, main);
  if (c.launch && c.use_aux) c.fail(stream_wait(main, c.s.aux));
  mpndiff_forward(c, m, s.p, P.pk, p, site_seed(s.seed, 3), P.p.h, P.r.h, s.mode == RR_STEP_DEDUP ? s.amap : nullptr,
                  s.feat, s.F, site_seed(s.seed, 4), P.d, main);
  ffn_forward(c, m, P.pk, s.p.M, p, site_seed(s.seed, 5), P.d.vecs, P.d.ld_vecs, s.out, P.f, main);
  P.fwd_end = c.ar.off;
}

// ------------------------------------------------------------------------------------------------ backward pieces
struct EncGrads { float *wi, *bi, *wh, *bh, *wo, *bo; };

// d message = adjoint of the bond message (one gather over b2b_t; row 0 from the GEMM's weighted column sums)
float* bond_adjoint(Ctx& c, const rr_graph& g, int H, const float* d_min, const float* part, hipStream_t st) {
  float* d_msg = c.alloc(g.nB, H);
  gather_sum(c, d_min, g.nB, H, g.b2b_t, g.nB, g.Kb, H, d_msg, H, st, part, rr_linear_colsum_rows(g.nB), r4(H));
  return d_msg;
}

// adjoint of mpn_forward; accumulate = the gradient buffers already hold the other encoder pass
void mpn_backward(Ctx& c, const rr_model& m, const rr_graph& g, const PackedW& pk, const Packed& wh_t, const Packed& wo_t,
                  float p, const EncSaved& S, const float* dH, float sign, const EncGrads& G, int accumulate) {
  const int H = m.H, depth = m.depth;
  const float ks = 1.0f / (1.0f - p);
  hipStream_t st = c.cur;
  float* dz_o = c.alloc(g.nA, H);
  float* d_a = c.alloc(g.nA, H);
  float* part = c.alloc(rr_linear_colsum_rows(g.nA), r4(H));
  {
    rr_linear_args a = LA(g.nA, H);
    a.a1 = dH; a.lda1 = H; a.k1 = H; a.a_mask = S.h; a.a_mask_bits = S.bits_h; a.ld_mask = H; a.mask_scale = sign * ks;
    a.dz_out = dz_o; a.ld_dz = H; set_w(a, wo_t);
    a.colsum_w = g.npad; a.colsum_partial = part; a.ld_partial = r4(H);
    a.c = d_a; a.ldc = H;
    lin(c, a, st);
  }
  {
    rr_wgrad_args w = WA(g.nA, H, dz_o, H, G.wo, m.enc_wo.in, G.bo, accumulate);
    w.x1 = g.f_atoms; w.ldx1 = g.ld_fa; w.k1 = m.atom_fdim; w.x2 = S.a_last; w.ldx2 = H; w.k2 = H;
    wgrad(c, w);
  }
  // From here on every gradient that reaches a layer is produced ALREADY masked by that layer's ReLU / dropout pattern:
  // the gather that forms d message applies (y > 0) / (1 - p) of the layer below in its epilogue (rr_gather_sum_epi_f32),
  // so dZ is the gather's output - the dX GEMMs read it as a plain operand and the weight gradients stream it as is -
  // and the last gather also adds every iteration's dZ: its output is d input (models/mpn.py:94), no pass of its own.
  const float* dzs[MAXD];
  int ndz = 0;
  float* cur = c.alloc(g.nB, H);        // dZ of iteration depth-2 (or d input when depth == 1)
  {
    const int top = depth - 1;         // the activation whose pattern masks this gradient: msgs[depth-1]
    gather_epi(c, d_a, g.nA, g.b2t, g.nB, 1, H, cur, st, part, rr_linear_colsum_rows(g.nA), true, S.msgs[top], S.bits[top],
               top == 0 ? 1.0f : ks, nullptr, 0);
  }
  for (int it = depth - 2; it >= 0; --it) {
    float* dz = cur;
    float* d_min = c.alloc(g.nB, H);
    float* partb = c.alloc(rr_linear_colsum_rows(g.nB), r4(H));
    rr_linear_args a = LA(g.nB, H);
    a.a1 = dz; a.lda1 = H; a.k1 = H; set_w(a, wh_t);
    a.colsum_w = g.npad_b; a.colsum_partial = partb; a.ld_partial = r4(H);
    a.c = d_min; a.ldc = H;
    rr_wgrad_args w = WA(g.nB, H, dz, H, G.wh, H, G.bh, (accumulate || it != depth - 2) ? 1 : 0);
    w.x1 = S.amsgs[it]; w.ldx1 = H; w.k1 = H; w.x1_idx = g.b2a; w.x1_sub = S.msgs[it]; w.ldx1_sub = H; w.x1_sub_idx = g.b2revb;
    if (c.wgrad_early) wgrad(c, w);                      // (starts with the dX GEMM that reads the same dZ instead of behind it)
    lin(c, a, st);
    if (!c.wgrad_early) wgrad(c, w);
    dzs[ndz++] = dz;
    // adjoint of the bond message (one gather over b2b_t; row 0 from the GEMM's weighted column sums), masked by msgs[it];
    // the last one (it == 0: msgs[0] = relu(input), no dropout) adds the dZ of every iteration -> d input
    cur = c.alloc(g.nB, H);
    gather_epi(c, d_min, g.nB, g.b2b_t, g.nB, g.Kb, H, cur, st, partb, rr_linear_colsum_rows(g.nB), true, S.msgs[it], S.bits[it],
               it == 0 ? 1.0f : ks, dzs, it == 0 ? ndz : 0);
  }
  float* d_inp = cur;
  rr_wgrad_args w = WA(g.nB, H, d_inp, H, G.wi, m.enc_wi.in, G.bi, accumulate);
  w.x1 = g.f_bonds; w.ldx1 = g.ld_fb; w.k1 = m.bond_fdim;
  wgrad(c, w, /*tail=*/true);
}

void mpn_backward_shared(Ctx& c, const rr_model& m, const rr_graph& gu, const rr_graph& g, const int32_t* bmap_t, int bmap_t_cols,
                         const Packed& wh_t, const Packed& wo_t, float p, const EncSaved& S, const float* dH, float sign,
                         const EncGrads& G, int accumulate) {
  const int H = m.H, depth = m.depth;
  const float ks = 1.0f / (1.0f - p);
  hipStream_t st = c.cur;
  float* dz_o = c.alloc(g.nA, H);
  float* d_a = c.alloc(g.nA, H);
  float* part = c.alloc(rr_linear_colsum_rows(g.nA), r4(H));
  {
    rr_linear_args a = LA(g.nA, H);
    a.a1 = dH; a.lda1 = H; a.k1 = H; a.a_mask = S.h; a.a_mask_bits = S.bits_h; a.ld_mask = H; a.mask_scale = sign * ks;
    a.dz_out = dz_o; a.ld_dz = H; set_w(a, wo_t);
    a.colsum_w = g.npad; a.colsum_partial = part; a.ld_partial = r4(H);
    a.c = d_a; a.ldc = H;
    lin(c, a, st);
  }
  {
    rr_wgrad_args w = WA(g.nA, H, dz_o, H, G.wo, m.enc_wo.in, G.bo, accumulate);
    w.x1 = g.f_atoms; w.ldx1 = g.ld_fa; w.k1 = m.atom_fdim; w.x2 = S.a_last; w.ldx2 = H; w.k2 = H;
    wgrad(c, w);
  }
  // per-copy W_h layers (it >= 1): gradients arrive masked from the gather that forms them (see mpn_backward); the one
  // that reaches the shared prefix stays unmasked - rr_gather_sum_masked_f32 masks it while summing over the copies
  float* d_msg = c.alloc(g.nB, H);
  {
    const bool per_copy = depth - 2 >= 1;
    gather_epi(c, d_a, g.nA, g.b2t, g.nB, 1, H, d_msg, st, part, rr_linear_colsum_rows(g.nA), per_copy, S.msgs[depth - 1],
               S.bits[depth - 1], ks, nullptr, 0);
  }
  const float* fulls[RR_MAX_GATHER_SRCS];   // the per-copy layers' dZ, oldest first (counted, not pointer-tested: a layout pass
  int n_full = 0;                           // hands out null pointers)
  const bool multi = H % 4 == 0 && c.gather_multi;
  int wh_started = accumulate;
  for (int it = depth - 2; it >= 1; --it) {                                             // per-copy W_h layers
    float* dz = d_msg;
    float* d_min = c.alloc(g.nB, H);
    float* partb = c.alloc(rr_linear_colsum_rows(g.nB), r4(H));
    rr_linear_args a = LA(g.nB, H);
    a.a1 = dz; a.lda1 = H; a.k1 = H; set_w(a, wh_t);
    a.colsum_w = g.npad_b; a.colsum_partial = partb; a.ld_partial = r4(H);
    a.c = d_min; a.ldc = H;
    rr_wgrad_args w = WA(g.nB, H, dz, H, G.wh, H, G.bh, wh_started);
    w.x1 = S.amsgs[it]; w.ldx1 = H; w.k1 = H; w.x1_idx = g.b2a; w.x1_sub = S.msgs[it]; w.ldx1_sub = H; w.x1_sub_idx = g.b2revb;
    if (c.wgrad_early) wgrad(c, w);
    lin(c, a, st);
    if (!c.wgrad_early) wgrad(c, w);
    wh_started = 1;
    // d input of a copy = the sum of these dZ; only its sum over the copies is needed (below).  Up to RR_MAX_GATHER_SRCS
    // addends ride on that gather (rr_gather_sum_multi_f32); beyond that - or with rows that are not whole 16-byte chunks -
    // the oldest ones are pre-summed by rr_axpby_f32 as before.  Either way the additions and their order are the same.
    if (n_full < (multi ? RR_MAX_GATHER_SRCS : 1)) {
      fulls[n_full++] = dz;
    } else {
      float* sum = c.alloc(g.nB, H);                                                     // fresh buffer (side-stream readers)
      RR_TRY(c, rr_axpby_f32(1.0f, fulls[0], 1.0f, multi ? fulls[1] : dz, sum, g.nB * static_cast<int64_t>(H), st));
      fulls[0] = sum;
      if (multi) {
        for (int q = 1; q + 1 < n_full; ++q) fulls[q] = fulls[q + 1];
        fulls[n_full - 1] = dz;
      }
    }
    d_msg = c.alloc(g.nB, H);
    gather_epi(c, d_min, g.nB, g.b2b_t, g.nB, g.Kb, H, d_msg, st, partb, rr_linear_colsum_rows(g.nB), it - 1 >= 1, S.msgs[it],
               S.bits[it], ks, nullptr, 0);
  }
  // ---- shared prefix: msgs[1] = drop_copy(z1_u[bmap]),  z1_u = relu(inp_u + m_in0_u W_h^T + b_h)
  // dz1 of every copy is read once, by the sum over the copies: mask and gather in one pass (no [nB, H] round trip), the
  // copies' masks re-derived from the dropout stream and z1_u instead of read back from msgs[1]
  float* dz1_u = c.alloc(gu.nB, H);
  if (H % 4 == 0)
    RR_TRY(c, rr_gather_sum_dropmask_f32(d_msg, g.nB, H, S.z1_u, H, bmap_t, gu.nB, bmap_t_cols, H, p, S.seed0, ks, dz1_u, H, st));
  else
    RR_TRY(c, rr_gather_sum_masked_f32(d_msg, S.msgs[1], g.nB, H, bmap_t, gu.nB, bmap_t_cols, H, ks, dz1_u, H, st));
  // the prefix's W_h weight gradient needs dz1_u only: issued here, before the main stream's gather / axpby below, so that the
  // weight-gradient stream's tail (this one, then W_i's) ends ~40 us earlier in front of the optimizer (same order among the
  // weight gradients, so the accumulation order - and every bit - is unchanged)
  {
    rr_wgrad_args w = WA(gu.nB, H, dz1_u, H, G.wh, H, G.bh, wh_started);
    w.x1 = S.a0_u; w.ldx1 = H; w.k1 = H; w.x1_idx = gu.b2a; w.x1_sub = S.msg0_u; w.ldx1_sub = H; w.x1_sub_idx = gu.b2revb;
    wgrad(c, w);
  }
  // d input of the distinct bonds = sum over the copies of the per-copy layers' dZ  +  dz1_u  +  relu'(msg0_u) (.) d msg0_u.
  // The first summand is a gather of its own (a different table); the other two ride on the epilogue of the gather that forms
  // d msg0_u - addends in this order, the masked gather last: the additions of the former axpby / ReLU-backward passes over
  // d_inp_u in their order, without those passes, and the tensor is written once (its bound comes out of that launch)
  const float* adds[2];
  int n_adds = 0;
  if (n_full >= 1) {
    float* over_copies = c.alloc(gu.nB, H);
    if (n_full == 1) gather_sum(c, fulls[0], g.nB, H, bmap_t, gu.nB, bmap_t_cols, H, over_copies, H, st, nullptr, 0, 0, /*claim=*/false);
    else gather_multi(c, fulls, n_full, g.nB, H, bmap_t, gu.nB, bmap_t_cols, H, over_copies, H, st);
    adds[n_adds++] = over_copies;
  }
  adds[n_adds++] = dz1_u;
  float* d_min_u = c.alloc(gu.nB, H);
  float* part_u = c.alloc(rr_linear_colsum_rows(gu.nB), r4(H));
  {
    rr_linear_args a = LA(gu.nB, H);
    a.a1 = dz1_u; a.lda1 = H; a.k1 = H; set_w(a, wh_t);
    a.colsum_w = gu.npad_b; a.colsum_partial = part_u; a.ld_partial = r4(H);
    a.c = d_min_u; a.ldc = H;
    lin(c, a, st);
  }
  float* d_inp_u = c.alloc(gu.nB, H);
  gather_epi(c, d_min_u, gu.nB, gu.b2b_t, gu.nB, gu.Kb, H, d_inp_u, st, part_u, rr_linear_colsum_rows(gu.nB), true, S.msg0_u, nullptr,
             1.0f, adds, n_adds);                          // msg0 = relu(inp): no dropout, no sign-bit image
  rr_wgrad_args w = WA(gu.nB, H, d_inp_u, H, G.wi, m.enc_wi.in, G.bi, accumulate);
  w.x1 = gu.f_bonds; w.ldx1 = gu.ld_fb; w.k1 = m.bond_fdim;
  wgrad(c, w, /*tail=*/true);
}

// adjoint of mpndiff_forward -> d_x [nA, H]
float* mpndiff_backward(Ctx& c, const rr_model& m, const rr_graph& g, float p, const DiffSaved& S, const float* x, const float* x_sub,
                        const int32_t* x_sub_idx, const float* dvecs, int64_t ld_dvecs, int F, uint64_t out_seed,
                        const rr_grads& G, const PackedT& T) {
  const int H = m.H, depth = m.diff_depth, FB = m.bond_fdim;
  const float ks = 1.0f / (1.0f - p);
  hipStream_t st = c.s.main;
  float* d_hid = c.alloc(g.nA, H);
  float* d_x = nullptr;
  float* d_inp = nullptr;
  if (depth > 0) {
    // hid = drop(relu(.)): the readout's adjoint applies that pattern as it writes (dZ of W_o), so both column blocks
    // of W_o read a plain operand; the message-passing iterations below follow mpn_backward
    float* dz_o = d_hid;
    RR_TRY(c, rr_segment_mean_bwd_masked_amax_f32(dvecs, ld_dvecs, g.a_scope, g.atom2mol, g.nA, H, F, p, out_seed, S.hid, H, S.bits_hid,
                                                  ks, dz_o, H, amax_claim(c, dz_o), st));
    const Packed wo_x = T.dif_wo_x, wo_a = T.dif_wo_a;
    d_x = c.alloc(g.nA, H);
    {
      rr_linear_args a = LA(g.nA, H);
      a.a1 = dz_o; a.lda1 = H; a.k1 = H; set_w(a, wo_x); a.c = d_x; a.ldc = H;
      lin(c, a, st);
    }
    {
      rr_wgrad_args w = WA(g.nA, H, dz_o, H, G.w[RR_G_DIF_WO], 2 * H, G.b[RR_G_DIF_WO], 0);
      w.x1 = x; w.ldx1 = H; w.k1 = H; w.x1_sub = x_sub; w.ldx1_sub = H; w.x1_sub_idx = x_sub_idx;
      w.x2 = S.a_last; w.ldx2 = H; w.k2 = H;
      wgrad(c, w);
    }
    float* d_a = c.alloc(g.nA, H);
    float* part = c.alloc(rr_linear_colsum_rows(g.nA), r4(H));
    {
      rr_linear_args a = LA(g.nA, H);
      a.a1 = dz_o; a.lda1 = H; a.k1 = H;
      set_w(a, wo_a); a.colsum_w = g.npad; a.colsum_partial = part; a.ld_partial = r4(H);
      a.c = d_a; a.ldc = H;
      lin(c, a, st);
    }
    const float* dzs[MAXD];
    int ndz = 0;
    float* cur = c.alloc(g.nA, H);
    {
      const int top = depth - 1;
      gather_epi(c, d_a, g.nA, g.a2a_t, g.nA, g.K, H, cur, st, part, rr_linear_colsum_rows(g.nA), true, S.msgs[top], S.bits[top],
                 top == 0 ? 1.0f : ks, nullptr, 0);
    }
    const Packed wh_t = T.dif_wh;
    for (int it = depth - 2; it >= 0; --it) {
      float* dz = cur;
      float* d_a2 = c.alloc(g.nA, H);
      float* part2 = c.alloc(rr_linear_colsum_rows(g.nA), r4(H));
      rr_linear_args a = LA(g.nA, H);
      a.a1 = dz; a.lda1 = H; a.k1 = H; set_w(a, wh_t);
      a.colsum_w = g.npad; a.colsum_partial = part2; a.ld_partial = r4(H);
      a.c = d_a2; a.ldc = H;
      rr_wgrad_args w = WA(g.nA, H, dz, H, G.w[RR_G_DIF_WH], H + FB, G.b[RR_G_DIF_WH], it != depth - 2 ? 1 : 0);
      w.x1 = S.amsgs[it]; w.ldx1 = H; w.k1 = H; w.x2 = g.fb_sum; w.ldx2 = g.ld_fbs; w.k2 = FB;
      if (c.wgrad_early) wgrad(c, w);
      lin(c, a, st);
      if (!c.wgrad_early) wgrad(c, w);
      dzs[ndz++] = dz;
      cur = c.alloc(g.nA, H);
      gather_epi(c, d_a2, g.nA, g.a2a_t, g.nA, g.K, H, cur, st, part2, rr_linear_colsum_rows(g.nA), true, S.msgs[it], S.bits[it],
                 it == 0 ? 1.0f : ks, dzs, it == 0 ? ndz : 0);
    }
    d_inp = cur;
  } else {
    RR_TRY(c, rr_segment_mean_bwd_f32(dvecs, ld_dvecs, g.a_scope, g.atom2mol, g.nA, H, F, p, out_seed, d_hid, H, st));
    d_inp = c.alloc(g.nA, H);
    RR_TRY(c, rr_relu_bwd_f32(d_hid, S.msgs[0], ks, d_inp, nullptr, g.nA * static_cast<int64_t>(H), st));   // hid = drop(relu(inp))
  }
  {
    rr_wgrad_args w = WA(g.nA, H, d_inp, H, G.w[RR_G_DIF_WI], H, G.b[RR_G_DIF_WI], 0);
    w.x1 = x; w.ldx1 = H; w.k1 = H; w.x1_sub = x_sub; w.ldx1_sub = H; w.x1_sub_idx = x_sub_idx;
    wgrad(c, w);
  }
  const Packed wi_t = T.dif_wi;
  rr_linear_args a = LA(g.nA, H);
  a.a1 = d_inp; a.lda1 = H; a.k1 = H; set_w(a, wi_t);
  if (depth == 0) {
    d_x = c.alloc(g.nA, H);
  } else {
    a.residual = d_x; a.ldr = H;
  }
  a.c = d_x; a.ldc = H;
  lin(c, a, st);
  return d_x;
}

// adjoint of ffn_forward -> d vecs [M, H] (the readout columns only)
// rows of the transposed pack of FFN layer li (= input columns whose gradient is formed): the first layer only
// needs the readout columns (the appended feature columns are inputs without gradient)
int ffn_dx_rows(const rr_model& m, int li) {
  if (li == m.n_ffn - 1 || li > 0) return m.ffn[li].in;
  return m.n_ffn > 1 ? m.H : m.ffn[li].in;
}

float* ffn_backward(Ctx& c, const rr_model& m, int64_t M, float p, const FfnSaved& S, const float* dout, const rr_grads& G,
                    const PackedT& T, int64_t* ld_out) {
  const float ks = 1.0f / (1.0f - p);
  hipStream_t st = c.s.main;
  const int nl = m.n_ffn;
  const rr_linear_w& L = m.ffn[nl - 1];
  const float* d = dout;
  if (m.head != 0) {
    float* draw = c.alloc(M, L.out);
    RR_TRY(c, rr_head_bwd_f32(dout, S.raw, M, L.out, m.head, draw, st));
    d = draw;
  }
  {
    rr_wgrad_args w = WA(M, L.out, d, L.out, G.w[RR_G_FFN0 + nl - 1], L.in, G.b[RR_G_FFN0 + nl - 1], 0);
    w.x1 = S.hs[nl - 1]; w.ldx1 = S.ld_hs[nl - 1]; w.k1 = L.in;
    wgrad(c, w);
  }
  // the input gradients of every layer: dx[nl-1] = d W_last, dx[li] = (dx[li+1] * relu'/dropout pattern of hs[li+1]) W_li
  float* dx[RR_MAX_FFN];
  int64_t ldx[RR_MAX_FFN];
  for (int li = nl - 1; li >= 0; --li) {
    ldx[li] = r4(ffn_dx_rows(m, li));
    dx[li] = c.alloc(M, ldx[li]);
  }
  // ONE launch for the whole chain where rr_ffn_chain_f32 takes the shape (see ffn_forward); the weight gradients of the hidden
  // layers, which read the chain's intermediate dx, are issued after it (each writes its own buffers: the order is free)
  bool chained = false;
  if (c.ffn_chain && nl >= 2 && c.launch && c.status == RR_OK) {
    rr_ffn_chain_args a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.n_stages = nl; a.x = d; a.ldx = L.out; a.drop_p = 0.f; a.mask_scale = ks;
    bool ok = true;
    for (int j = 0; j < nl; ++j) {
      const int li = nl - 1 - j;
      rr_ffn_stage& g = a.stage[j];
      ok = ok && T.ffn[li].mode == 1;
      g.w = T.ffn[li].w; g.ldw = T.ffn[li].ld; g.n_out = ffn_dx_rows(m, li); g.n_in = m.ffn[li].out;
      g.out = dx[li]; g.ld_out = ldx[li];
      if (li > 0) { g.post_mask = S.hs[li]; g.ld_mask = S.ld_hs[li]; }
    }
    if (ok) {
      const int stc = rr_ffn_chain_f32(&a, st);
      if (stc == RR_OK) chained = true;
      else if (stc != RR_ERR_UNSUPPORTED) c.fail(stc);
    }
  }
  if (!chained) {
    const Packed wt = T.ffn[nl - 1];
    rr_linear_args a = LA(M, ffn_dx_rows(m, nl - 1));
    a.a1 = d; a.lda1 = L.out; a.k1 = L.out; set_w(a, wt); a.c = dx[nl - 1]; a.ldc = ldx[nl - 1];
    lin(c, a, st);
  }
  for (int li = nl - 2; li >= 0; --li) {
    const rr_linear_w& Lh = m.ffn[li];
    const float* y = S.hs[li + 1];                       // drop(relu(.)) output of this layer
    {
      rr_wgrad_args w = WA(M, Lh.out, dx[li + 1], ldx[li + 1], G.w[RR_G_FFN0 + li], Lh.in, G.b[RR_G_FFN0 + li], 0);
      w.mask = y; w.ld_mask = S.ld_hs[li + 1]; w.mask_scale = ks;
      w.x1 = S.hs[li]; w.ldx1 = S.ld_hs[li]; w.k1 = Lh.in;
      wgrad(c, w);
    }
    if (chained) continue;
    const Packed wt = T.ffn[li];
    rr_linear_args a = LA(M, ffn_dx_rows(m, li));
    a.a1 = dx[li + 1]; a.lda1 = ldx[li + 1]; a.k1 = Lh.out; a.a_mask = y; a.ld_mask = S.ld_hs[li + 1]; a.mask_scale = ks;
    set_w(a, wt); a.c = dx[li]; a.ldc = ldx[li];
    lin(c, a, st);
  }
  *ld_out = ldx[0];
  return dx[0];
}

void backward_all(Ctx& c, const rr_model& m, const rr_step& s, Plan& P, const float* dout, const rr_grads& G) {
  const int H = m.H;
  const float p = s.drop_p;
  hipStream_t main = c.s.main;
  // the transposed weights of the input-gradient GEMMs: packed by the forward (RR_PLAN_TRAIN), or here in ONE launch
  if (!c.train) {
    pack_transposes(c, m, P.T, main);
    flush_packs(c, main);
  }
  if (c.f16 && c.launch && c.status == RR_OK && c.amax_base != nullptr && c.namax < MAX_AMAX &&     // the backward's own slots
      hipMemsetAsync(c.amax_base + static_cast<size_t>(c.namax) * RR_AMAX_FLOATS, 0, static_cast<size_t>(MAX_AMAX - c.namax) * RR_AMAX_FLOATS * sizeof(float), main) != hipSuccess)
    c.fail(RR_ERR_LAUNCH);
  const PackedT& T = P.T;
  c.reg(dout, s.p.M, m.ffn[m.n_ffn - 1].out, m.ffn[m.n_ffn - 1].out);   // (the FFN's weight gradients split too from 8192 molecules per step on)
  int64_t ld_dvecs = 0;
  float* dvecs = ffn_backward(c, m, s.p.M, p, P.f, dout, G, T, &ld_dvecs);
  const int32_t* xsi = s.mode == RR_STEP_DEDUP ? s.amap : nullptr;
  float* d_diff = mpndiff_backward(c, m, s.p, p, P.d, P.p.h, P.r.h, xsi, dvecs, ld_dvecs, s.F, site_seed(s.seed, 4), G, T);
  // de-duplicated reactants: d r_h[u] = -(sum over the copies of atom u of d_diff) (fixed-order segment sum)
  const float* d_r = d_diff;
  if (s.mode == RR_STEP_DEDUP) {
    float* t = c.alloc(s.r.nA, H);
    gather_sum(c, d_diff, s.p.nA, H, s.amap_t, s.r.nA, s.amap_t_cols, H, t, H, main);
    d_r = t;
  }
  const Packed wh_t = T.enc_wh, wo_t = T.enc_wo;
  EncGrads E;
  E.wi = G.w[RR_G_ENC_WI]; E.bi = G.b[RR_G_ENC_WI]; E.wh = G.w[RR_G_ENC_WH]; E.bh = G.b[RR_G_ENC_WH];
  E.wo = G.w[RR_G_ENC_WO]; E.bo = G.b[RR_G_ENC_WO];
  // the two encoder passes share weights: the product pass writes the gradient buffers, the reactant pass accumulates
  // The reactant pass may run on the aux stream beside the product pass (independent chains; their gather kernels are
  // HBM-bound and co-reside with the other chain's one-workgroup-per-CU GEMMs).  Weight gradients of both go to the one
  // side stream in issue order - product pass first - so the accumulation order does not depend on the overlap.
  // Only with the side stream: without it the weight gradients launch on c.cur, so the reactant pass on aux would
  // accumulate into buffers the product pass is still writing on main (no ordering between the two).
  const bool fork = c.aux_bwd && c.use_aux && c.use_side;
  if (fork && c.split) {               // both passes read these bounds: found on main before the aux stream forks off
    amax_of(c, d_diff, main);
    amax_of(c, d_r, main);
  }
  if (c.launch && fork) c.fail(stream_wait(c.s.aux, main));
  mpn_backward(c, m, s.p, P.pk, wh_t, wo_t, p, P.p, d_diff, 1.0f, E, 0);
  if (fork) c.cur = c.s.aux;
  c.tail_pass = !fork;
  if (s.mode == RR_STEP_PREFIX) mpn_backward_shared(c, m, s.u, s.r, s.bmap_t, s.bmap_t_cols, wh_t, wo_t, p, P.r, d_r, -1.0f, E, 1);
  else mpn_backward(c, m, s.r, P.pk, wh_t, wo_t, p, P.r, d_r, -1.0f, E, 1);
  c.cur = main;
  c.tail_pass = false;
  if (c.launch && fork) c.fail(stream_wait(main, c.s.aux));
  if (c.launch && c.use_side && !c.side_joined) c.fail(stream_wait(main, c.s.side));     // weight gradients are complete from here on
}

int check_graph(const rr_graph& g, int need_fb_sum, int need_f_bonds) {
  RR_CHECK_ARG(g.nA >= 1 && g.nB >= 1 && g.M >= 0 && g.K >= 1 && g.Kb >= 1);
  RR_CHECK_ARG(!need_f_bonds || g.f_bonds);
  RR_CHECK_ARG(g.f_atoms && g.a2b && g.b2a && g.b2revb && g.a2a && g.a_scope && g.b2t && g.a2a_t && g.atom2mol &&
               g.b2b_t && g.npad && g.npad_b);
  RR_CHECK_ARG(!need_fb_sum || g.fb_sum);
  return RR_OK;
}

int check(const rr_model* m, const rr_step* s) {
  RR_CHECK_ARG(m && s);
  RR_CHECK_ARG(m->H >= 4 && m->H % 4 == 0 && m->depth >= 1 && m->depth <= MAXD && m->diff_depth >= 0 && m->diff_depth <= MAXD);
  RR_CHECK_ARG(m->n_ffn >= 1 && m->n_ffn <= RR_MAX_FFN && m->atom_fdim == ATOM_FDIM && m->bond_fdim >= 1);
  RR_CHECK_ARG(m->enc_wi.w && m->enc_wo.w && m->dif_wi.w && (m->depth == 1 || m->enc_wh.w) && (m->diff_depth <= 1 || m->dif_wh.w) &&
               (m->diff_depth == 0 || m->dif_wo.w));
  RR_CHECK_ARG(s->mode == RR_STEP_PLAIN || s->mode == RR_STEP_DEDUP || s->mode == RR_STEP_PREFIX);
  RR_CHECK_ARG(s->drop_p >= 0.f && s->drop_p < 1.f && s->F >= 0 && (s->F == 0 || s->feat) && s->out);
  int st = check_graph(s->p, m->diff_depth > 1, 1);
  if (st != RR_OK) return st;
  st = check_graph(s->r, 0, s->mode != RR_STEP_PREFIX);    // shared prefix: W_i reads the distinct reactants' f_bonds only
  if (st != RR_OK) return st;
  if (s->mode == RR_STEP_DEDUP) RR_CHECK_ARG(s->amap && s->amap_t && s->amap_t_cols >= 1 && s->drop_p == 0.f);
  else RR_CHECK_ARG(s->r.nA == s->p.nA);
  if (s->mode == RR_STEP_PREFIX) {
    RR_CHECK_ARG(m->depth >= 2 && s->bmap && s->bmap_t && s->bmap_t_cols >= 1);
    st = check_graph(s->u, 0, 1);
    if (st != RR_OK) return st;
  }
  RR_CHECK_ARG(m->ffn[0].in == m->H + s->F);
  return RR_OK;
}

}  // namespace

extern "C" {

void rr_abi_plan_struct_sizes(size_t* graph, size_t* model, size_t* step, size_t* grads) {
  if (graph) *graph = sizeof(rr_graph);
  if (model) *model = sizeof(rr_model);
  if (step) *step = sizeof(rr_step);
  if (grads) *grads = sizeof(rr_grads);
}

size_t rr_reaction_workspace_bytes(const rr_model* model, const rr_step* step) {
  if (check(model, step) != RR_OK) return 0;
  size_t need = 0;
  for (int v = 0; v < 6; ++v) {                        // every GEMM path (f32, three bf16 terms, two f16 terms), with or without RR_PLAN_TRAIN, must fit
    Ctx c;
    c.launch = false; c.status = RR_OK; c.ar.base = nullptr; c.ar.off = 0; c.ar.cap = 0; c.ar.overflow = false; c.npq = 0;
    c.use_side = c.use_aux = false;
    c.aux_bwd = false;
    c.cur = nullptr;
    c.s.main = c.s.side = c.s.aux = nullptr;
    c.split = (v >> 1) != 0;
    c.f16 = (v >> 1) == 2;
    c.ntr = c.namax = 0; c.amax_base = nullptr;
    c.train = (v & 1) != 0;
    c.ffn_chain = true;
    c.timing = false;
    c.gather_multi = false;             // (the layout with the pre-summed buffers: the larger one)
    c.tail_pass = c.side_joined = false;
    c.wgrad_early = false;
    Plan P;
    memset(&P, 0, sizeof(P));
    forward_all(c, *model, *step, P);
    rr_grads G;
    memset(&G, 0, sizeof(G));
    backward_all(c, *model, *step, P, nullptr, G);
    if (c.ar.off > need) need = c.ar.off;
  }
  return need + 256;
}

int rr_reaction_forward(const rr_model* model, const rr_step* step, int flags, rr_stream_t stream) {
  int st = check(model, step);
  if (st != RR_OK) return st;
  RR_CHECK_ARG(step->workspace && rr_aligned16(step->workspace));
  Ctx c;
  c.launch = true; c.status = RR_OK; c.npq = 0;
  c.ar.base = static_cast<char*>(step->workspace); c.ar.off = 0; c.ar.cap = step->workspace_bytes; c.ar.overflow = false;
  c.use_side = (flags & RR_PLAN_NO_SIDE_STREAM) == 0;
  c.use_aux = (flags & RR_PLAN_NO_AUX_STREAM) == 0;
  c.split = (flags & RR_PLAN_F32_GEMM) == 0;
  c.f16 = c.split && (flags & RR_PLAN_F16X2_GEMM) != 0;
  c.ntr = c.namax = 0; c.amax_base = nullptr;
  c.aux_bwd = (flags & RR_PLAN_AUX_BACKWARD) != 0;
  c.train = (flags & RR_PLAN_TRAIN) != 0;
  c.ffn_chain = (flags & RR_PLAN_NO_FFN_CHAIN) == 0 && !getenv("RR_NO_FFN_CHAIN");
  c.timing = (flags & RR_PLAN_TIME) != 0;
  c.gather_multi = !getenv("RR_NO_GATHER_MULTI");
  c.tail_pass = c.side_joined = false;
  c.wgrad_early = (flags & RR_PLAN_WGRAD_EARLY) != 0 || getenv("RR_WGRAD_EARLY") != nullptr;
  st = get_streams(static_cast<hipStream_t>(stream), &c.s);
  if (st != RR_OK) return st;
  c.cur = c.s.main;
  // the layout must fit BEFORE anything is launched (a dry pass costs microseconds)
  {
    Ctx d = c;
    d.launch = false; d.ar.base = nullptr;
    Plan Q;
    memset(&Q, 0, sizeof(Q));
    forward_all(d, *model, *step, Q);
    if (d.ar.off > step->workspace_bytes) return RR_ERR_WORKSPACE;
  }
  Plan P;
  memset(&P, 0, sizeof(P));
  forward_all(c, *model, *step, P);
  return c.ar.overflow ? RR_ERR_WORKSPACE : c.status;
}

int rr_reaction_backward(const rr_model* model, const rr_step* step, const float* dout, const rr_grads* grads, int flags,
                         rr_stream_t stream) {
  int st = check(model, step);
  if (st != RR_OK) return st;
  RR_CHECK_ARG(step->workspace && dout && grads);
  for (int i = 0; i < RR_G_FFN0 + model->n_ffn; ++i) {
    const bool needed = !((i == RR_G_ENC_WH && model->depth <= 1) || (i == RR_G_DIF_WH && model->diff_depth <= 1) ||
                          (i == RR_G_DIF_WO && model->diff_depth == 0));
    RR_CHECK_ARG(!needed || grads->w[i]);
  }
  Ctx c;
  c.launch = false; c.status = RR_OK; c.npq = 0;
  c.ar.base = static_cast<char*>(step->workspace); c.ar.off = 0; c.ar.cap = step->workspace_bytes; c.ar.overflow = false;
  c.use_side = (flags & RR_PLAN_NO_SIDE_STREAM) == 0;
  c.use_aux = (flags & RR_PLAN_NO_AUX_STREAM) == 0;
  c.split = (flags & RR_PLAN_F32_GEMM) == 0;
  c.f16 = c.split && (flags & RR_PLAN_F16X2_GEMM) != 0;
  c.ntr = c.namax = 0; c.amax_base = nullptr;
  c.aux_bwd = (flags & RR_PLAN_AUX_BACKWARD) != 0;
  c.train = (flags & RR_PLAN_TRAIN) != 0;
  c.ffn_chain = (flags & RR_PLAN_NO_FFN_CHAIN) == 0 && !getenv("RR_NO_FFN_CHAIN");
  c.timing = (flags & RR_PLAN_TIME) != 0;
  c.gather_multi = !getenv("RR_NO_GATHER_MULTI");
  c.tail_pass = c.side_joined = false;
  c.wgrad_early = (flags & RR_PLAN_WGRAD_EARLY) != 0 || getenv("RR_WGRAD_EARLY") != nullptr;
  st = get_streams(static_cast<hipStream_t>(stream), &c.s);
  if (st != RR_OK) return st;
  c.cur = c.s.main;
  Plan P;
  memset(&P, 0, sizeof(P));
  forward_all(c, *model, *step, P);                      // layout pass: re-derives the address of every saved activation
  {
    Ctx d = c;                                           // ... and the backward's temporaries must fit as well
    d.ar.base = nullptr;
    Plan Q = P;
    backward_all(d, *model, *step, Q, dout, *grads);
    if (d.ar.off > step->workspace_bytes) return RR_ERR_WORKSPACE;
  }
  c.launch = true;
  backward_all(c, *model, *step, P, dout, *grads);
  return c.ar.overflow ? RR_ERR_WORKSPACE : c.status;
}

int rr_plan_timing_select(int kinds, int modes) {
  RR_CHECK_ARG(kinds >= 0 && kinds <= 7 && modes >= 0 && modes <= 15);
  g_timing_kinds = kinds;
  g_timing_modes = modes;
  return RR_OK;
}

int rr_plan_timing_take(rr_plan_timing* out, int max_n) {
  RR_CHECK_ARG(max_n >= 0 && (out || max_n == 0));
  std::vector<TimingRec> recs;
  {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    recs.swap(g_timing);
  }
  int n = 0, st = RR_OK;
  for (TimingRec& r : recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) st = RR_ERR_LAUNCH;
    if (st == RR_OK && n < max_n) {
      out[n] = r.info;
      out[n].us = ms * 1000.f;
      ++n;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  return st == RR_OK ? n : st;
}

int rr_reaction_saved_f32(const rr_model* model, const rr_step* step, int flags, int which, int index,
                          const float** ptr, int64_t* rows, int64_t* ld) {
  int st = check(model, step);
  if (st != RR_OK) return st;
  RR_CHECK_ARG(step->workspace && ptr && rows && ld && index >= 0 && index < MAXD);
  Ctx c;
  c.launch = false; c.status = RR_OK; c.npq = 0;
  c.ar.base = static_cast<char*>(step->workspace); c.ar.off = 0; c.ar.cap = step->workspace_bytes; c.ar.overflow = false;
  c.use_side = (flags & RR_PLAN_NO_SIDE_STREAM) == 0;
  c.use_aux = (flags & RR_PLAN_NO_AUX_STREAM) == 0;
  c.split = (flags & RR_PLAN_F32_GEMM) == 0;
  c.f16 = c.split && (flags & RR_PLAN_F16X2_GEMM) != 0;
  c.ntr = c.namax = 0; c.amax_base = nullptr;
  c.aux_bwd = (flags & RR_PLAN_AUX_BACKWARD) != 0;
  c.train = (flags & RR_PLAN_TRAIN) != 0;
  c.ffn_chain = (flags & RR_PLAN_NO_FFN_CHAIN) == 0 && !getenv("RR_NO_FFN_CHAIN");
  c.timing = (flags & RR_PLAN_TIME) != 0;
  c.gather_multi = !getenv("RR_NO_GATHER_MULTI");
  c.tail_pass = c.side_joined = false;
  c.wgrad_early = (flags & RR_PLAN_WGRAD_EARLY) != 0 || getenv("RR_WGRAD_EARLY") != nullptr;
  c.s.main = c.s.side = c.s.aux = nullptr;
  c.cur = nullptr;
  Plan P;
  memset(&P, 0, sizeof(P));
  forward_all(c, *model, *step, P);                      // layout pass only
  if (c.ar.overflow) return RR_ERR_WORKSPACE;
  const int H = model->H;
  const float* q = nullptr;
  int64_t r = 0, l = H;
  switch (which) {
    case RR_SAVED_R_MSG: if (index < model->depth) { q = P.r.msgs[index]; r = step->r.nB; } break;
    case RR_SAVED_R_H: q = P.r.h; r = step->r.nA; break;
    case RR_SAVED_P_MSG: if (index < model->depth) { q = P.p.msgs[index]; r = step->p.nB; } break;
    case RR_SAVED_P_H: q = P.p.h; r = step->p.nA; break;
    case RR_SAVED_D_MSG: if (index < (model->diff_depth > 0 ? model->diff_depth : 1)) { q = P.d.msgs[index]; r = step->p.nA; } break;
    case RR_SAVED_D_HID: q = P.d.hid; r = step->p.nA; break;
    case RR_SAVED_VECS: q = P.d.vecs; r = step->p.M; l = P.d.ld_vecs; break;
    case RR_SAVED_FFN_H: if (index >= 1 && index < model->n_ffn) { q = P.f.hs[index]; r = step->p.M; l = P.f.ld_hs[index]; } break;
    case RR_SAVED_R_MSG0_U: if (step->mode == RR_STEP_PREFIX) { q = P.r.msg0_u; r = step->u.nB; } break;
    case RR_SAVED_R_Z1_U: if (step->mode == RR_STEP_PREFIX) { q = P.r.z1_u; r = step->u.nB; } break;
    default: return RR_ERR_ARG;
  }
  *ptr = q;
  *rows = q ? r : 0;
  *ld = l;
  return RR_OK;
}

}  // extern "C"
