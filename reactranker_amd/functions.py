"""Op wrappers + the hand-written forward/backward of the reaction scorer.

Layer 1: thin wrappers that launch one C-ABI entry point on torch device tensors
         (pointers, sizes and the current HIP stream go straight to the library).
Layer 2: the encoder / diff-encoder / FFN forward and backward passes, written against
         layer 1 only — the backward is explicit (no autograd graph of tiny ops): every
         gather-sum's adjoint is a gather-sum over the packer's transposed tables, weight
         gradients are split-M MFMA GEMMs with a fixed-order reduction.
Layer 3: torch.autograd.Function shells so the nn.Modules behave like the reference's.

Reference sites are cited per function (paths relative to the reference root).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional

import torch

from . import _lib
from ._lib import LinearArgs, WgradArgs, check, lib, ptr, stream

ACT_NONE, ACT_RELU = 0, 1
ATOM_FDIM, BOND_FDIM = 61, 22
FBOND = ATOM_FDIM + BOND_FDIM      # 83: width of an f_bonds row

HEADS = {  # FFN task_type string -> rr_head (reference models/base_model.py:61-106)
    "evidential_with_softplus": 6, "gauss_regression_with_softplus": 4, "gaussian_with_softplus": 4,
    "listnetdis_lognorm_with_softplus": 5, "evidential_ranking": 3, "listnet_with_softplus": 1,
    "listnet_with_uncertainty": 2, "evidential": 2,
}


class Profiler:
    """Optional live timing of every launch of the heavy kernels with HIP events on the launch
    stream (bench.py's roofline leg).  Off by default; records (kernel key, work, event pair)."""
    enabled = False
    records = []
    only = None             # optional set of kernel keys: time just these (keeps the event overhead out of the rest)

    @classmethod
    def start(cls, only=None):
        cls.enabled, cls.records, cls.only = True, [], (set(only) if only else None)

    @classmethod
    def stop(cls):
        cls.enabled = False
        return cls.records


class _Timed:
    def __init__(self, key, flops, nbytes):
        self.on = Profiler.enabled and (Profiler.only is None or key in Profiler.only)
        if self.on:
            self.key, self.flops, self.nbytes = key, flops, nbytes
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            Profiler.records.append((self.key, self.flops, self.nbytes, self.e0, self.e1))
        return False


def _rowmajor(t: torch.Tensor, name: str) -> torch.Tensor:
    """float32 CUDA tensor whose rows are contiguous (row stride >= row length); copies otherwise
    (e.g. the stride-0 expanded gradients autograd hands to backward)."""
    _lib.require_cuda(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if t.dim() == 2:
        if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
            t = t.contiguous()
    elif not t.is_contiguous():
        t = t.contiguous()
    return t


def _ld(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    return t.stride(0) if t.dim() == 2 else t.numel()


def _new(like: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


# =========================================================================== layer 1
def gather_sum(src: torch.Tensor, idx: torch.Tensor, H: int, out: torch.Tensor = None, row0_partial=None, *, mask=None,
               mask_scale: float = 1.0, adds=()) -> torch.Tensor:
    """out[r] = sum_k src[idx[r,k]] — index_select_ND + sum(dim=1) (utils.py:176-193; models/mpn.py:89-90).
    row0_partial [n, >=H]: output row 0 (the padding row) is the fixed-order sum of these rows instead — the
    weighted column sums a dX GEMM wrote next to its output (linear(..., colsum_w=...)).
    mask / adds: the fused epilogue of rr_gather_sum_epi_f32 — out = (mask > 0 ? out * mask_scale : 0) + sum(adds), i.e.
    the ReLU / dropout backward of the layer below and the sum over the residual's readers ride on the gather (the
    mask's sign-bit image is read instead of the f32 tensor when the GEMM that produced it wrote one)."""
    n_out, K = (idx.shape[0], idx.shape[1]) if idx.dim() == 2 else (idx.shape[0], 1)
    if out is None:
        out = _new(src, n_out, H)
    adds = list(adds)
    epi = mask is not None or len(adds) > 0
    # algorithmic bytes: every source row once, every output row once, the index table once (+ the epilogue's operands)
    nbytes = 4 * (src.shape[0] * H + n_out * H + n_out * K)
    if epi:
        bits = None if mask is None else getattr(mask, "_rr_bits", None)
        nbytes += 4 * n_out * H * len(adds) + (0 if mask is None else (n_out * bits.shape[1] if bits is not None else 4 * n_out * H))
        E = _lib.GatherEpi()
        if mask is not None:
            assert mask.shape[0] == n_out
            E.mask, E.ld_mask, E.mask_bits = ptr(mask), _ld(mask), ptr(bits)
        E.mask_scale = float(mask_scale)
        E.n_adds, E.ld_add = len(adds), (_ld(adds[0]) if adds else 0)
        for j, t in enumerate(adds):
            assert t.shape[0] == n_out and _ld(t) == E.ld_add
            E.adds[j] = t.data_ptr()
    with _Timed("gather_sum_epi_kernel" if epi else "gather_sum_kernel", 0, nbytes):
        if epi:
            check(lib().rr_gather_sum_epi_f32(ptr(src), src.shape[0], _ld(src), ptr(idx), n_out, K, H, ptr(row0_partial),
                                              0 if row0_partial is None else row0_partial.shape[0], _ld(row0_partial),
                                              C.byref(E), ptr(out), _ld(out), stream()), "rr_gather_sum_epi_f32")
        elif row0_partial is None:
            check(lib().rr_gather_sum_f32(ptr(src), src.shape[0], _ld(src), ptr(idx), n_out, K, H, ptr(out), _ld(out),
                                          stream()), "rr_gather_sum_f32")
        else:
            check(lib().rr_gather_sum_padrow_f32(ptr(src), src.shape[0], _ld(src), ptr(idx), n_out, K, H,
                                                 ptr(row0_partial), row0_partial.shape[0], _ld(row0_partial), ptr(out),
                                                 _ld(out), stream()), "rr_gather_sum_padrow_f32")
    return out


def gather_sum_masked(src, mask, scale: float, idx, H: int, out=None):
    """out[r] = sum_k (mask[idx[r,k]] > 0 ? src[idx[r,k]] * scale : 0): relu_bwd + gather_sum in one pass, bit-identical."""
    n_out, K = idx.shape
    if out is None:
        out = _new(src, n_out, H)
    assert mask.shape == src.shape and mask.stride(0) == src.stride(0)
    with _Timed("gather_sum_masked_kernel", 0, 4 * (2 * src.shape[0] * H + n_out * H + n_out * K)):
        check(lib().rr_gather_sum_masked_f32(ptr(src), ptr(mask), src.shape[0], _ld(src), ptr(idx), n_out, K, H, float(scale),
                                             ptr(out), _ld(out), stream()), "rr_gather_sum_masked_f32")
    return out


def gather_sum_dropmask(src, y, scale: float, idx, H: int, p: float, seed: int, out=None):
    """gather_sum_masked(src, gather_dropout(y, copies, H, p, seed), scale, idx, H) without the materialised copies: the keep
    bits come from the dropout stream, the sign from `y` (the pre-dropout activations of the DESTINATION rows); bit-identical."""
    n_out, K = idx.shape
    if out is None:
        out = _new(src, n_out, H)
    assert y.shape[0] == n_out
    with _Timed("gather_sum_dropmask_kernel", 0, 4 * (src.shape[0] * H + 2 * n_out * H + n_out * K)):
        check(lib().rr_gather_sum_dropmask_f32(ptr(src), src.shape[0], _ld(src), ptr(y), _ld(y), ptr(idx), n_out, K, H, float(p),
                                               int(seed) & 0xFFFFFFFFFFFFFFFF, float(scale), ptr(out), _ld(out), stream()),
              "rr_gather_sum_dropmask_f32")
    return out


MAX_GATHER_SRCS = _lib.RR_MAX_GATHER_SRCS


def gather_sum_multi(srcs, idx, H: int, out=None):
    """out[r] = sum_k (((srcs[0] + srcs[1]) + srcs[2]) + ...)[idx[r,k]] (rr_gather_sum_multi_f32): the chain of axpby passes
    that would form the inner sum followed by gather_sum, bit-identical, with every addend read once and no [n_src, H]
    intermediate.  More than MAX_GATHER_SRCS addends (or rows that are not whole 16-byte chunks): the oldest are pre-summed."""
    srcs = list(srcs)
    limit = MAX_GATHER_SRCS if (H % 4 == 0 and not os.environ.get("RR_NO_GATHER_MULTI")) else 1
    while len(srcs) > limit:
        srcs = [axpby(1.0, srcs[0], 1.0, srcs[1])] + srcs[2:]
    if len(srcs) == 1:
        return gather_sum(srcs[0], idx, H, out=out)
    n_out, K = idx.shape
    if out is None:
        out = _new(srcs[0], n_out, H)
    assert all(t.shape == srcs[0].shape and t.stride(0) == srcs[0].stride(0) for t in srcs)
    arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    with _Timed("gather_sum_multi_kernel", 0, 4 * (len(srcs) * srcs[0].shape[0] * H + n_out * H + n_out * K)):
        check(lib().rr_gather_sum_multi_f32(arr, len(srcs), srcs[0].shape[0], _ld(srcs[0]), ptr(idx), n_out, K, H, ptr(out),
                                            _ld(out), stream()), "rr_gather_sum_multi_f32")
    return out


def gather_sum_csr(src: torch.Tensor, offsets: torch.Tensor, idx: torch.Tensor, n_out: int, H: int) -> torch.Tensor:
    """out[r] = sum of src[idx[j]] for j in [offsets[r], offsets[r+1]) — adjoint of a gather through a generic index."""
    out = _new(src, n_out, H)
    check(lib().rr_gather_sum_csr_f32(ptr(src), src.shape[0], _ld(src), ptr(offsets), ptr(idx), n_out, H, ptr(out),
                                      _ld(out), stream()), "rr_gather_sum_csr_f32")
    return out


def _transpose_index(flat_idx: torch.Tensor, n_src: int):
    """CSR transpose of a flat destination->source index, built on the device without any host sync:
    (offsets [n_src+1] int32, dest [nnz] int32) with the destinations of every source row in increasing order."""
    flat = flat_idx.reshape(-1).to(torch.int64)
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=n_src)[:n_src]
    offsets = torch.zeros(n_src + 1, dtype=torch.int32, device=flat.device)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return offsets, order.to(torch.int32)


def gather_diff(a, ia, m, im, H: int, out=None):
    """out[r] = a[ia[r]] - m[im[r]] (models/mpn.py:91-92)."""
    n_out = ia.shape[0]
    if out is None:
        out = _new(a, n_out, H)
    check(lib().rr_gather_diff_f32(ptr(a), a.shape[0], _ld(a), ptr(ia), ptr(m), m.shape[0], _ld(m), ptr(im), n_out, H,
                                   ptr(out), _ld(out), stream()), "rr_gather_diff_f32")
    return out


def gather_dropout(src, idx, H: int, p: float, seed: int):
    """out[r] = dropout_r(src[idx[r]]): rows shared by several destinations, each with its own mask."""
    out = _new(src, idx.shape[0], H)
    check(lib().rr_gather_dropout_f32(ptr(src), src.shape[0], _ld(src), ptr(idx), idx.shape[0], H, float(p),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(out), _ld(out), stream()),
          "rr_gather_dropout_f32")
    return out


def weighted_colsum(x, w, H: int, out, accumulate: bool):
    """out[0:H] (+)= sum_r w[r] x[r]; the padding row's adjoint."""
    n = x.shape[0]
    nbytes = lib().rr_colsum_workspace_bytes(n, H)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=x.device)
    check(lib().rr_weighted_colsum_f32(ptr(x), n, _ld(x), ptr(w), H, ptr(out), int(accumulate), ptr(ws), nbytes,
                                       stream()), "rr_weighted_colsum_f32")
    return out


def amax(t, cols=None):
    """max |t| over the first `cols` columns of a 2-D tensor as a magnitude slot (rr_amax_f32: RR_AMAX_FLOATS device floats
    whose maximum is the bound; `float(slot.max())` reads it)"""
    out = torch.zeros(_lib.RR_AMAX_FLOATS, dtype=torch.float32, device=t.device)
    check(lib().rr_amax_f32(ptr(t), t.shape[0], int(t.shape[1] if cols is None else cols), _ld(t), ptr(out), stream()), "rr_amax_f32")
    return out


def linear(M: int, N: int, w, *, a1=None, k1=0, a1_idx=None, a1_sub=None, a1_sub_idx=None, a2=None, k2=0, a_mask=None,
           mask_scale=1.0, ldw=None, w_packed=False, bias=None, residual=None, act=ACT_NONE, drop_p=0.0, seed=0,
           out=None, c_pre=None, dz_out=None, dz_accumulate=False, residual_idx=None, colsum_w=None, mask_bits_out=None,
           a_mask_bits=None, want_bits=False, amax_of=None):
    """One fused dense layer on the f32 MFMA (see rr_linear_args in include/reactranker_hip.h).
    colsum_w [M]: also returns the per-row-block partial sums of colsum_w[m] * out[m, :]  ->  (out, partial)."""
    ref = a1 if a1 is not None else a2
    if out is None:
        out = _new(ref, M, N)
    partial = None
    if colsum_w is not None:
        partial = _new(ref, int(lib().rr_linear_colsum_rows(M)), (N + 3) // 4 * 4)
    A = LinearArgs()
    A.M, A.N = M, N
    A.a1, A.lda1, A.k1, A.a1_idx = ptr(a1), _ld(a1), k1, ptr(a1_idx)
    A.a1_sub, A.lda1_sub, A.a1_sub_idx = ptr(a1_sub), _ld(a1_sub), ptr(a1_sub_idx)
    A.a2, A.lda2, A.k2 = ptr(a2), _ld(a2), k2
    A.a_mask, A.ld_mask, A.mask_scale = ptr(a_mask), _ld(a_mask), float(mask_scale)
    A.dz_out, A.ld_dz, A.dz_accumulate = ptr(dz_out), _ld(dz_out), int(dz_accumulate)
    if w.dtype == torch.uint8:                          # LinW.pk / pk_t handed out the three-bf16-term images
        w_packed, ldw = (3 if getattr(w, "_rr_f16", False) else 2), 0   # (3: two f16 terms)
    A.w, A.ldw, A.w_packed = ptr(w), (w.stride(0) if ldw is None else ldw), int(w_packed)
    if int(w_packed) == 3:                              # operand bounds of the two-f16-term form (the step plans keep theirs in the workspace)
        bounds = amax_of if amax_of is not None else [amax(t, k) if t is not None else None for t, k in ((a1, k1), (a1_sub, k1), (a2, k2))]
        A.a1_amax, A.a1_sub_amax, A.a2_amax = ptr(bounds[0]), ptr(bounds[1]), ptr(bounds[2])
    A.bias = ptr(bias)
    A.residual, A.ldr, A.residual_idx = ptr(residual), _ld(residual), ptr(residual_idx)
    A.act, A.drop_p, A.drop_seed = act, float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF
    A.c, A.ldc = ptr(out), _ld(out)
    A.c_pre, A.ld_pre = ptr(c_pre), _ld(c_pre)
    A.colsum_w, A.colsum_partial, A.ld_partial = ptr(colsum_w), ptr(partial), _ld(partial)
    split_w = w.dtype == torch.uint8
    if want_bits and split_w and mask_bits_out is None:  # the output will be a ReLU / dropout mask of a dX GEMM: keep its signs
        mask_bits_out = torch.empty(M, int(lib().rr_mask_bits_row_bytes(N)), dtype=torch.uint8, device=out.device)
        out._rr_bits = mask_bits_out
    if a_mask is not None and a_mask_bits is None and split_w:
        a_mask_bits = getattr(a_mask, "_rr_bits", None)  # ... and read them instead of the f32 activation (same values)
    A.mask_bits_out, A.a_mask_bits = ptr(mask_bits_out), ptr(a_mask_bits)    # sign-bit images (split GEMM only)
    nt = 4 if N <= 64 else (10 if N <= 160 else 19)
    if M <= 8192 and not (w.dtype == torch.uint8 or int(w_packed) >= 2):
        nt = 4                                          # few rows: 64-column blocks (rr_linear_f32)
    mode = 3 if a_mask_bits is not None else (2 if a_mask is not None else (1 if a1_sub is not None else 0))
    kk = k1 + k2
    two_src = a1_sub is not None or (a_mask is not None and a_mask_bits is None)
    nbytes = 4 * (M * kk * (2 if two_src else 1) + N * kk + M * N * (1 + (residual is not None) + (c_pre is not None)))
    if a_mask_bits is not None:
        nbytes += M * a_mask_bits.shape[1]              # the mask as one bit per element
    if mask_bits_out is not None:
        nbytes += M * mask_bits_out.shape[1]
    if dz_out is not None:                              # side output d_input (+)= dZ: one write, one read when accumulating
        nbytes += 4 * M * k1 * (2 if dz_accumulate else 1)
    if int(w_packed) >= 2:                              # the symbol rocprofv3 shows: <tiles packed, tiles per WG, mode, waves>
        ntp = 38 if N > 304 else nt
        if nt == 19 and ntp == 19 and M <= 8192:
            key = f"linear_split_kernel<19,5,{mode},8>"   # few rows: column blocks of 5 tiles (rr_linear_f32)
        else:
            key = f"linear_split_kernel<{ntp},{nt},{mode},{12 if nt == 19 else 8}>"
    else:
        key = f"{'linear_fast_kernel' if w_packed else 'linear_kernel'}<{nt},{mode}>"
    with _Timed(key, 2 * M * N * kk, nbytes):
        check(lib().rr_linear_f32(C.byref(A), stream()), "rr_linear_f32")
    return out if colsum_w is None else (out, partial)


class SideStream:
    """Weight-gradient GEMMs are off the backward critical path (nothing downstream reads dW until the
    optimizer), so they run on a second HIP stream next to the dX / gather chain: their workgroups fill
    the tail rounds and the prologue/epilogue bubbles of the main stream's kernels.
    Ordering: the side stream waits for the main stream before each launch (inputs ready); every input
    tensor is record_stream()-ed so the caching allocator does not recycle it early; join() makes the
    main stream wait for all outstanding weight gradients."""
    enabled = True
    _streams = {}

    @classmethod
    def get(cls, device):
        key = str(device)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=device)
        return cls._streams[key]

    @classmethod
    def join(cls, device):
        if cls.enabled and str(device) in cls._streams:
            torch.cuda.current_stream(device).wait_stream(cls._streams[str(device)])


class AuxStream:
    """Second compute stream for the reactant encoder: encoder(r) and encoder(p) are independent
    (models/base_model.py:155-156), so their kernel chains run concurrently and fill each other's
    tail rounds.  Used by ReactionModelFn only."""
    enabled = True
    backward = False        # measured: concurrent r/p backward chains cost ~2 % (they fight the weight-gradient stream)
    _streams = {}

    @classmethod
    def get(cls, device):
        key = str(device)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=device)
        return cls._streams[key]


SPLIT_MIN_ROWS = 8192      # GEMMs over fewer rows (the FFN head: one row per molecule) stay on the f32 matrix core


def wgrad(M: int, N: int, dy, dw, *, dbias=None, mask=None, mask_scale=1.0, x1=None, k1=0, x1_idx=None, x1_sub=None,
          x1_sub_idx=None, x2=None, k2=0, accumulate=False, ld_dw=None, side=False, amax_of=None):
    """dw (+)= dZ^T [X1|X2], dbias (+)= colsum(dZ) with dZ = dy * (mask > 0) * mask_scale.
    side=True (the model's backward passes) launches on the weight-gradient stream; the caller must
    SideStream.join() before anything reads dw/dbias."""
    if side and SideStream.enabled:
        main = torch.cuda.current_stream(dy.device)
        side = SideStream.get(dy.device)
        if side != main:
            side.wait_stream(main)
            for t in (dy, mask, x1, x1_idx, x1_sub, x1_sub_idx, x2, dw, dbias):
                if t is not None:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                return _wgrad_launch(M, N, dy, dw, dbias, mask, mask_scale, x1, k1, x1_idx, x1_sub, x1_sub_idx, x2, k2,
                                     accumulate, ld_dw, amax_of)
    return _wgrad_launch(M, N, dy, dw, dbias, mask, mask_scale, x1, k1, x1_idx, x1_sub, x1_sub_idx, x2, k2, accumulate,
                         ld_dw, amax_of)


def _wgrad_launch(M, N, dy, dw, dbias, mask, mask_scale, x1, k1, x1_idx, x1_sub, x1_sub_idx, x2, k2, accumulate, ld_dw, amax_of=None):
    K = k1 + k2
    nbytes = lib().rr_linear_wgrad_workspace_bytes(M, N, K)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dy.device)
    A = WgradArgs()
    A.M, A.N = M, N
    A.dy, A.ld_dy = ptr(dy), _ld(dy)
    A.mask, A.ld_mask, A.mask_scale = ptr(mask), _ld(mask), float(mask_scale)
    A.x1, A.ldx1, A.k1, A.x1_idx = ptr(x1), _ld(x1), k1, ptr(x1_idx)
    A.x1_sub, A.ldx1_sub, A.x1_sub_idx = ptr(x1_sub), _ld(x1_sub), ptr(x1_sub_idx)
    A.x2, A.ldx2, A.k2 = ptr(x2), _ld(x2), k2
    A.dw, A.ld_dw = ptr(dw), (dw.stride(0) if ld_dw is None else ld_dw)
    A.dbias = ptr(dbias)
    A.accumulate = int(accumulate)
    A.workspace, A.workspace_bytes = ptr(ws), nbytes
    A.split = int(SplitGemm.enabled and M >= SPLIT_MIN_ROWS and N % 4 == 0 and
                  all(t is None or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0) for t in (dy, mask, x1, x1_sub, x2)))
    if A.split and (amax_of is not None or SplitGemm.f16):   # two f16 terms: bounds of (dy, x1, x1_sub, x2) as device floats
        if amax_of is None:                              # (the step plans get them from the producing kernels instead)
            amax_of = [amax(dy), amax(x1, k1) if x1 is not None else None, amax(x1_sub, k1) if x1_sub is not None else None,
                       amax(x2, k2) if x2 is not None else None]
        A.split = 2
        A.dy_amax, A.x1_amax, A.x1_sub_amax, A.x2_amax = (ptr(t) for t in amax_of)
    # the event pair spans the main kernel and its ~12 us fixed-order reduce kernel
    kext = ((k1 + 3) & ~3) + k2 + 1                     # same k-block choice as rr_linear_wgrad_f32 (96 / 128 / 160 columns)
    nblk = (kext + 159) // 160
    per = (kext + nblk - 1) // nblk
    wtk = 3 if per <= 96 else (4 if per <= 128 else 5)
    key = (f"{'wgrad_split_kernel' if A.split else 'wgrad_fast_kernel'}<{'true' if mask is not None else 'false'},"
           f"{'true' if x1_sub is not None else 'false'},{wtk}>")
    with _Timed(key, 2 * M * N * (K + 1), 4 * (M * N * (2 if mask is not None else 1) + M * K + N * K)):
        check(lib().rr_linear_wgrad_f32(C.byref(A), stream()), "rr_linear_wgrad_f32")
    return dw


def relu_bwd(dy, y, scale: float, dz=None, acc=None, want_dz=True):
    if dz is None and want_dz:
        dz = torch.empty_like(y)
    check(lib().rr_relu_bwd_f32(ptr(dy), ptr(y), float(scale), ptr(dz), ptr(acc), y.numel(), stream()),
          "rr_relu_bwd_f32")
    return dz


MAX_ADDS = 15      # = RR_MAX_ADDS (csrc/elementwise.hip): the step plan hands every dZ of a depth <= 16 model to one launch


def relu_bwd_sum(dy, y, scale: float, adds):
    """out = sum(adds) + dy * (y > 0) * scale in one pass (rr_relu_bwd_sum_f32); more than MAX_ADDS addends are
    folded pairwise first."""
    adds = list(adds)
    while len(adds) > MAX_ADDS:
        adds = [axpby(1.0, adds[0], 1.0, adds[1])] + adds[2:]
    out = torch.empty_like(y)
    arr = (C.c_void_p * max(1, len(adds)))(*[t.data_ptr() for t in adds])
    check(lib().rr_relu_bwd_sum_f32(ptr(dy), ptr(y), float(scale), arr, len(adds), ptr(out), y.numel(), stream()),
          "rr_relu_bwd_sum_f32")
    return out


def axpby(alpha: float, a, beta: float = 0.0, b=None, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib().rr_axpby_f32(float(alpha), ptr(a), float(beta), ptr(b), ptr(out), a.numel(), stream()), "rr_axpby_f32")
    return out


def dropout(x, p: float, seed: int, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(lib().rr_dropout_f32(ptr(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(out), stream()),
          "rr_dropout_f32")
    return out


def segment_mean_fwd(x, g, H: int, feat, F: int, drop_p: float, seed: int):
    out = _new(x, g.M, (H + F + 3) // 4 * 4)[:, :H + F]          # rows padded to 16 bytes for the fast GEMM path
    check(lib().rr_segment_mean_fwd_f32(ptr(x), _ld(x), ptr(g.a_scope), g.M, H, ptr(feat), F, float(drop_p),
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(out), _ld(out), stream()),
          "rr_segment_mean_fwd_f32")
    return out


def segment_mean_bwd(dout, g, H: int, F: int, drop_p: float, seed: int, mask=None, mask_scale: float = 1.0):
    """mask: also apply the ReLU / dropout backward of the readout's input, dx = (mask > 0) ? dx * mask_scale : 0."""
    dx = _new(dout, g.nA, H)
    if mask is not None:
        check(lib().rr_segment_mean_bwd_masked_f32(ptr(dout), _ld(dout), ptr(g.a_scope), ptr(g.atom2mol), g.nA, H, F,
                                                   float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(mask), _ld(mask),
                                                   ptr(getattr(mask, "_rr_bits", None)), float(mask_scale), ptr(dx), _ld(dx),
                                                   stream()), "rr_segment_mean_bwd_masked_f32")
        return dx
    check(lib().rr_segment_mean_bwd_f32(ptr(dout), _ld(dout), ptr(g.a_scope), ptr(g.atom2mol), g.nA, H, F,
                                        float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(dx), _ld(dx), stream()),
          "rr_segment_mean_bwd_f32")
    return dx


def head_fwd(raw, head: int):
    out = torch.empty_like(raw)
    check(lib().rr_head_fwd_f32(ptr(raw), raw.shape[0], raw.shape[1], head, ptr(out), stream()), "rr_head_fwd_f32")
    return out


def head_bwd(dout, raw, head: int):
    draw = torch.empty_like(raw)
    check(lib().rr_head_bwd_f32(ptr(dout), ptr(raw), raw.shape[0], raw.shape[1], head, ptr(draw), stream()),
          "rr_head_bwd_f32")
    return draw


# =========================================================================== layer 2
class GradSink:
    """Where the explicit backward writes parameter gradients.  None: fresh tensors.  dp.GradBucket.attach() registers
    a lookup (parameter storage -> a fresh view into its flat all-reduce buffer), so the gradients are BORN inside the
    bucket: no pack / unpack copies around the collective (autograd adopts the returned view as `.grad`)."""
    lookup = None


def _grad_like(t):
    if GradSink.lookup is not None:
        v = GradSink.lookup(t)
        if v is not None:
            return v
    return torch.empty_like(t, memory_format=torch.contiguous_format)


class SplitGemm:
    """Encoder GEMMs on the bf16 matrix core: every f32 operand is written exactly as three bf16 terms and six
    products are accumulated in f32 (rr_linear_args.w_packed = 2; no operand bit is dropped, error at or below the f32 MFMA
    chain's).  This is the default: a query's scores do not depend on what else is in its batch, and an N-process run
    reproduces the 1-process run up to the gradient bucket's summation order.
    enabled = False keeps every GEMM on v_mfma_f32_16x16x4_f32.
    f16 = True (opt-in since round 5; RR_F16X2=1 in the environment turns it on): two f16 terms per operand instead
    (w_packed = 3, RR_PLAN_F16X2_GEMM): three products, 22 significant bits per operand scaled to its TENSOR's largest
    magnitude - narrower than f32 operands and batch-dependent below 2^-40 of that magnitude,
    measured error at the f32 MFMA chain's level.  The step plans take the operand bounds from the producing kernels; the
    per-op mirror below finds them with one rr_amax_f32 pass per operand (slower: it exists for tests and per-kernel timing)."""
    enabled = True
    f16 = os.environ.get("RR_F16X2", "0") not in ("", "0")


class FfnChain:
    """The FFN head and its input-gradient chain as ONE launch each inside the step plans (rr_ffn_chain_f32, csrc/ffn.hip; bit-identical
    to the per-layer launches this mirror issues).  enabled = False (or RR_NO_FFN_CHAIN=1 in the environment) makes the plans
    issue the layers one by one as well: an A/B knob."""
    enabled = True


class LinW:
    """One nn.Linear's tensors plus lazily transposed copies for the input-gradient GEMM.
    big: its GEMMs run over atoms / bonds (the FFN head runs over molecules and stays on the f32 path)."""

    def __init__(self, weight, bias, big: bool = True):
        self.w = None if weight is None else _rowmajor(weight.detach(), "weight")
        self.b = None if bias is None else _rowmajor(bias.detach(), "bias")
        self.big = big
        self._t = {}

    def _pack(self, transpose: int, rows: int, c0: int, k1: int, k2: int):
        split = SplitGemm.enabled and self.big and rows <= 608 and rows % 4 == 0
        d = (_lib.PackDesc * 1)()
        if split:
            dst = torch.empty(int(lib().rr_split_weight_bytes(rows, k1, k2)), dtype=torch.uint8, device=self.w.device)
        else:
            dst = torch.empty(rows, int(lib().rr_packed_weight_ld(k1, k2)), dtype=torch.float32, device=self.w.device)
        d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = ptr(self.w), self.w.stride(0), transpose, rows, c0, k1, k2
        d[0].dst, d[0].split = ptr(dst), (2 if SplitGemm.f16 else 1) if split else 0
        check(lib().rr_pack_weights_f32(d, 1, stream()), "rr_pack_weights_f32")
        if split and SplitGemm.f16:
            dst._rr_f16 = True                           # two f16 terms: linear() passes w_packed = 3 and the operand bounds
        return dst

    def pk(self, k1: int, k2: int = 0):
        """Packed copy of W = [W1 | W2] for the fast GEMM paths (rr_pack_weights_f32: zero-padded f32, or bf16 terms)."""
        key = ("f", k1, k2, SplitGemm.enabled, SplitGemm.f16)
        if key not in self._t:
            self._t[key] = self._pack(0, self.w.shape[0], 0, k1, k2)
        return self._t[key]

    def pk_t(self, c0: int, c1: int):
        """Packed (W[:, c0:c1])^T — the weight operand of dX = dZ * W[:, c0:c1]."""
        key = ("t", c0, c1, SplitGemm.enabled, SplitGemm.f16)
        if key not in self._t:
            self._t[key] = self._pack(1, c1 - c0, c0, self.w.shape[0], 0)
        return self._t[key]

    def grads(self):
        gw = _grad_like(self.w)
        gb = None if self.b is None else _grad_like(self.b)
        return gw, gb


def _site_seed(seed: int, site: int) -> int:
    return (int(seed) * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def mpn_forward(g, H: int, depth: int, Wi: LinW, Wh: Optional[LinW], Wo: LinW, p: float, seed: int):
    """MPN.forward, return_atom_hiddens=True (models/mpn.py:61-108) -> (atom_hiddens [nA,H], saved)."""
    nA, nB = g.nA, g.nB
    inp = _new(g.f_bonds, nB, H)
    msg = _new(g.f_bonds, nB, H)
    linear(nB, H, Wi.pk(FBOND), w_packed=True, a1=g.f_bonds, k1=FBOND, bias=Wi.b, act=ACT_RELU, out=msg, c_pre=inp,
           want_bits=True)                                                                           # :80-81
    msgs, amsgs = [msg], []
    for it in range(depth - 1):                                                                      # :84
        a_msg = gather_sum(msgs[-1], g.a2b, H)                                                       # :89-90
        new = linear(nB, H, Wh.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=g.b2a, a1_sub=msgs[-1], a1_sub_idx=g.b2revb,
                     bias=Wh.b, residual=inp, act=ACT_RELU, drop_p=p, seed=_site_seed(seed, it), want_bits=True)     # :91-97
        amsgs.append(a_msg)
        msgs.append(new)
    del inp
    a_last = gather_sum(msgs[-1], g.a2b, H)                                                          # :101-102
    h = linear(nA, H, Wo.pk(ATOM_FDIM, H), w_packed=True, a1=g.f_atoms, k1=ATOM_FDIM, a2=a_last, k2=H, bias=Wo.b, act=ACT_RELU, drop_p=p,
               seed=_site_seed(seed, 1000), want_bits=True)                                                          # :103-105
    return h, (msgs, amsgs, a_last, h)


def bond_message_adjoint(d_min, g, H: int, partial=None, *, mask=None, mask_scale: float = 1.0, adds=()):
    """Adjoint of  message = a_message[b2a] - message[b2revb],  a_message = sum_k message[a2b]  (models/mpn.py:89-92):
    d message[b] = sum of d_min over the bonds leaving target(b) except rev(b) - ONE gather-sum over the packer's
    bond-to-bond table (rr_derive_bond_tables) instead of gather-sum + gather-diff; the padding row's adjoint
    (row 0 is read K - deg(a) times by atom a) is the weighted column sum over d_min."""
    if partial is not None:                              # the GEMM that produced d_min already summed npad_b[b] * d_min[b]
        return gather_sum(d_min, g.b2b_t, H, row0_partial=partial, mask=mask, mask_scale=mask_scale, adds=adds)
    assert mask is None and not adds
    d_msg = gather_sum(d_min, g.b2b_t, H)
    weighted_colsum(d_min, g.npad_b, H, d_msg[0], accumulate=True)
    return d_msg


def _pad_row_fix(d_src, d_out, g, H):
    """Row 0 of every gathered source is read npad[a] times by atom a (a2b is right-padded with
    bond/atom 0, features/featurization.py:286): add sum_a npad[a] * d_out[a] to d_src[0]."""
    weighted_colsum(d_out, g.npad, H, d_src[0], accumulate=True)


def mpn_backward(g, H: int, depth: int, Wi: LinW, Wh: Optional[LinW], Wo: LinW, p: float, saved, dH, sign: float, into=None):
    """Adjoint of mpn_forward.  `dH` is d loss / d atom_hiddens times `sign` (the reactant encoder
    sees -d_diff, models/base_model.py:168).  Returns grads (gWi, gbi, gWh, gbh, gWo, gbo).
    into: gradient buffers that already hold the OTHER encoder pass (the two passes share weights): this pass
    accumulates into them instead of returning fresh buffers that would have to be added afterwards."""
    msgs, amsgs, a_last, h = saved
    nA, nB = g.nA, g.nB
    ks = 1.0 / (1.0 - p)
    acc0 = into is not None
    if acc0:
        gWi, gbi, gWh, gbh, gWo, gbo = into
    else:
        gWi, gbi = Wi.grads()
        gWo, gbo = Wo.grads()
        gWh, gbh = (Wh.grads() if Wh is not None else (None, None))
    # atom_hiddens = drop(relu([f_atoms | a_last] W_o^T + b_o))
    fused = (H % 4 == 0)                                             # ReLU backward fused into producers / operand loads
    # Fused form, W_o layer: the input-gradient GEMM applies the ReLU/dropout mask in its operand loader (dH has two
    # consumers with different masks - the product and the reactant encoder) and writes the masked gradient dZ as a side
    # output; the weight-gradient GEMM (other stream) streams dZ as is.
    if fused:
        dz_o = torch.empty_like(dH)
        # ... and the padding row's adjoint sum_a npad[a] * d_a[a] as per-row-block partial sums (colsum_w)
        d_a, part = linear(nA, H, Wo.pk_t(ATOM_FDIM, ATOM_FDIM + H), w_packed=True, a1=dH, k1=H, a_mask=h,
                           mask_scale=sign * ks, dz_out=dz_o, colsum_w=g.npad)
        wgrad(nA, H, dz_o, gWo, dbias=gbo, x1=g.f_atoms, k1=ATOM_FDIM, x2=a_last, k2=H, accumulate=acc0, side=True)
        # a_last[a] = sum_k msg[a2b[a,k]]  ->  d_msg[b] = d_a[target(b)].  From here on every gradient that reaches a layer
        # is produced ALREADY masked by that layer's ReLU / dropout pattern: the gather that forms d message applies
        # (y > 0) / (1 - p) of the layer below in its epilogue, so dZ is the gather's output (the dX GEMM reads a plain
        # operand, the weight gradient streams it as is), and the last gather also adds every iteration's dZ -> d input.
        top = depth - 1
        cur = gather_sum(d_a, g.b2t, H, row0_partial=part, mask=msgs[top], mask_scale=(1.0 if top == 0 else ks))
        dzs = []
        for it in reversed(range(depth - 1)):
            dz = cur
            d_min, part = linear(nB, H, Wh.pk_t(0, H), w_packed=True, a1=dz, k1=H, colsum_w=g.npad_b)
            wgrad(nB, H, dz, gWh, dbias=gbh, x1=amsgs[it], k1=H, x1_idx=g.b2a, x1_sub=msgs[it], x1_sub_idx=g.b2revb,
                  accumulate=(acc0 or it != depth - 2), side=True)
            dzs.append(dz)
            cur = bond_message_adjoint(d_min, g, H, part, mask=msgs[it], mask_scale=(1.0 if it == 0 else ks),
                                       adds=(dzs if it == 0 else ()))
        d_inp = cur                                                  # sum_it dZ_it + relu'(inp) * d msgs[0]  (:94)
        wgrad(nB, H, d_inp, gWi, dbias=gbi, x1=g.f_bonds, k1=FBOND, accumulate=acc0, side=True)
        return gWi, gbi, gWh, gbh, gWo, gbo
    else:
        wgrad(nA, H, dH, gWo, dbias=gbo, mask=h, mask_scale=sign * ks, x1=g.f_atoms, k1=ATOM_FDIM, x2=a_last, k2=H,
              accumulate=acc0, side=True)
        d_a = linear(nA, H, Wo.pk_t(ATOM_FDIM, ATOM_FDIM + H), w_packed=True, a1=dH, k1=H, a_mask=h, mask_scale=sign * ks)
        d_msg = gather_sum(d_a, g.b2t, H)
        _pad_row_fix(d_msg, d_a, g, H)
    # H % 4 != 0: separate ReLU-backward passes.  dz is read by the weight-gradient stream, so every iteration gets a fresh
    # buffer and d_inp accumulates in a buffer that stream never reads before the final W_i launch (which is ordered
    # after all main-stream writes).
    d_inp = None
    for it in reversed(range(depth - 1)):
        # msgs[it+1] = drop(relu(inp + m_in W_h^T + b_h)),  m_in = amsgs[it][b2a] - msgs[it][b2revb]
        if d_inp is None:
            d_inp = torch.zeros_like(d_msg)
        dz = relu_bwd(d_msg, msgs[it + 1], ks, dz=torch.empty_like(d_msg), acc=d_inp)
        wgrad(nB, H, dz, gWh, dbias=gbh, x1=amsgs[it], k1=H, x1_idx=g.b2a, x1_sub=msgs[it],
              x1_sub_idx=g.b2revb, accumulate=(acc0 or it != depth - 2), side=True)
        d_min = linear(nB, H, Wh.pk_t(0, H), w_packed=True, a1=dz, k1=H)
        d_msg = bond_message_adjoint(d_min, g, H)
    # msgs[0] = relu(inp);  d inp = sum_it dZ_it + relu'(inp) * d msgs[0]   (inp is the residual of every iteration, :94)
    if d_inp is None:
        d_inp = relu_bwd(d_msg, msgs[0], 1.0)
    else:
        relu_bwd(d_msg, msgs[0], 1.0, acc=d_inp, want_dz=False)
    wgrad(nB, H, d_inp, gWi, dbias=gbi, x1=g.f_bonds, k1=FBOND, accumulate=acc0, side=True)
    return gWi, gbi, gWh, gbh, gWo, gbo


def mpn_forward_shared(gu, g, bmap, H: int, depth: int, Wi: LinW, Wh: LinW, Wo: LinW, p: float, seed: int):
    """mpn_forward for a batch whose molecules repeat (every candidate of a query carries the same
    reactant, train_listwise.py:188): everything BEFORE the first dropout — W_i, relu, the first
    gather-sum and the first W_h layer (models/mpn.py:80-95) — is identical across the copies, so it
    runs once per distinct molecule (graph `gu`) and is expanded with each copy's own dropout mask
    (`bmap`: full bond row -> distinct bond row).  The mask stream and indices are those of the
    plain path, so results are the same numbers.  Needs depth >= 2."""
    nA, nB, nBu = g.nA, g.nB, gu.nB
    inp_u = _new(gu.f_bonds, nBu, H)
    msg0_u = _new(gu.f_bonds, nBu, H)
    linear(nBu, H, Wi.pk(FBOND), w_packed=True, a1=gu.f_bonds, k1=FBOND, bias=Wi.b, act=ACT_RELU, out=msg0_u, c_pre=inp_u)
    a0_u = gather_sum(msg0_u, gu.a2b, H)
    z1_u = linear(nBu, H, Wh.pk(H), w_packed=True, a1=a0_u, k1=H, a1_idx=gu.b2a, a1_sub=msg0_u, a1_sub_idx=gu.b2revb,
                  bias=Wh.b, residual=inp_u, act=ACT_RELU)                          # pre-dropout, shared
    msgs = [None, gather_dropout(z1_u, bmap, H, p, _site_seed(seed, 0))]             # per-copy masks (:97)
    amsgs = [None]
    for it in range(1, depth - 1):
        a_msg = gather_sum(msgs[-1], g.a2b, H)
        new = linear(nB, H, Wh.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=g.b2a, a1_sub=msgs[-1], a1_sub_idx=g.b2revb,
                     bias=Wh.b, residual=inp_u, residual_idx=bmap, act=ACT_RELU, drop_p=p, seed=_site_seed(seed, it), want_bits=True)
        amsgs.append(a_msg)
        msgs.append(new)
    del inp_u
    a_last = gather_sum(msgs[-1], g.a2b, H)
    h = linear(nA, H, Wo.pk(ATOM_FDIM, H), w_packed=True, a1=g.f_atoms, k1=ATOM_FDIM, a2=a_last, k2=H, bias=Wo.b,
               act=ACT_RELU, drop_p=p, seed=_site_seed(seed, 1000), want_bits=True)
    return h, (msgs, amsgs, a_last, h, (msg0_u, a0_u, z1_u, _site_seed(seed, 0)))


def mpn_backward_shared(gu, g, bmap_t, H: int, depth: int, Wi: LinW, Wh: LinW, Wo: LinW, p: float, saved, dH, sign: float,
                        into=None):
    """Adjoint of mpn_forward_shared.  The per-copy layers run as in mpn_backward; the gradients that reach
    the shared prefix are summed over the copies (fixed-order segment sums over `bmap_t`) and the prefix is
    back-propagated once on the distinct molecules — every op there is linear in the gradient, so the sum
    commutes with it."""
    msgs, amsgs, a_last, h, (msg0_u, a0_u, z1_u, seed0) = saved
    nA, nB, nBu = g.nA, g.nB, gu.nB
    ks = 1.0 / (1.0 - p)
    acc0 = into is not None                                          # see mpn_backward
    if acc0:
        gWi, gbi, gWh, gbh, gWo, gbo = into
    else:
        gWi, gbi = Wi.grads()
        gWo, gbo = Wo.grads()
        gWh, gbh = Wh.grads()
    dz_o = torch.empty_like(dH)                                      # masked gradient as a side output (see mpn_backward)
    d_a, part = linear(nA, H, Wo.pk_t(ATOM_FDIM, ATOM_FDIM + H), w_packed=True, a1=dH, k1=H, a_mask=h, mask_scale=sign * ks,
                       dz_out=dz_o, colsum_w=g.npad)
    wgrad(nA, H, dz_o, gWo, dbias=gbo, x1=g.f_atoms, k1=ATOM_FDIM, x2=a_last, k2=H, accumulate=acc0, side=True)
    # per-copy W_h layers (it >= 1): gradients arrive masked from the gather that forms them (see mpn_backward); the one that
    # reaches the shared prefix stays unmasked - gather_sum_masked masks it while summing over the copies
    per_copy = depth - 2 >= 1
    d_msg = gather_sum(d_a, g.b2t, H, row0_partial=part, mask=(msgs[depth - 1] if per_copy else None), mask_scale=ks)
    fulls = []                                                       # the per-copy layers' dZ: d input of a copy is their sum
    wh_started = acc0
    for it in reversed(range(1, depth - 1)):                         # per-copy W_h layers
        dz = d_msg
        d_min, part = linear(nB, H, Wh.pk_t(0, H), w_packed=True, a1=dz, k1=H, colsum_w=g.npad_b)
        wgrad(nB, H, dz, gWh, dbias=gbh, x1=amsgs[it], k1=H, x1_idx=g.b2a, x1_sub=msgs[it], x1_sub_idx=g.b2revb,
              accumulate=wh_started, side=True)
        wh_started = True
        fulls.append(dz)
        d_msg = bond_message_adjoint(d_min, g, H, part, mask=(msgs[it] if it - 1 >= 1 else None), mask_scale=ks)
    # ---- shared prefix: msgs[1] = drop_copy(z1_u[bmap]),  z1_u = relu(inp_u + m_in0_u W_h^T + b_h)
    # (msgs[1] > 0) <=> kept and z1 > 0: mask and sum over the copies in one pass, the copies' masks re-derived from the
    # dropout stream instead of read (rows need 16-byte chunks: H % 4 == 0, else the materialised-mask form)
    if H % 4 == 0:
        dz1_u = gather_sum_dropmask(d_msg, z1_u, ks, bmap_t, H, p, seed0)
    else:
        dz1_u = gather_sum_masked(d_msg, msgs[1], ks, bmap_t, H)
    # d input of the distinct bonds = (the per-copy layers' dZ summed over layers and copies) + dz1_u + relu'(msg0_u) * d msg0_u:
    # the last two ride on the epilogue of the gather that forms d msg0_u (addends in this order, the masked gather last)
    adds = ([gather_sum_multi(fulls, bmap_t, H)] if fulls else []) + [dz1_u]
    wgrad(nBu, H, dz1_u, gWh, dbias=gbh, x1=a0_u, k1=H, x1_idx=gu.b2a, x1_sub=msg0_u, x1_sub_idx=gu.b2revb,
          accumulate=wh_started, side=True)
    d_min_u, part_u = linear(nBu, H, Wh.pk_t(0, H), w_packed=True, a1=dz1_u, k1=H, colsum_w=gu.npad_b)
    d_inp_u = bond_message_adjoint(d_min_u, gu, H, part_u, mask=msg0_u, mask_scale=1.0, adds=adds)   # msg0 = relu(inp)
    wgrad(nBu, H, d_inp_u, gWi, dbias=gbi, x1=gu.f_bonds, k1=FBOND, accumulate=acc0, side=True)
    return gWi, gbi, gWh, gbh, gWo, gbo


def mpndiff_forward(g, H: int, depth: int, Wi: LinW, Wh: Optional[LinW], Wo: Optional[LinW], p: float, seed: int,
                    x, x_sub=None, feat=None, F: int = 0, out_drop_p: float = 0.0, out_seed: int = 0, x_sub_idx=None):
    """MPNDiff.forward (models/mpn.py:170-240).  atom_features = x - x_sub (x_sub=None: x itself);
    returns ([M, H+F] readout (+ optional FFN-input dropout), saved)."""
    nA = g.nA
    Hin = x.shape[1]
    inp = _new(x, nA, H)
    msg = _new(x, nA, H)
    first_p = p if depth == 0 else 0.0                                                   # :221 (depth 0: dropout(message))
    linear(nA, H, Wi.pk(Hin), w_packed=True, a1=x, k1=Hin, a1_sub=x_sub, a1_sub_idx=x_sub_idx, bias=Wi.b, act=ACT_RELU,
           out=msg, c_pre=inp, want_bits=True,
           drop_p=first_p, seed=_site_seed(seed, 2000))                                  # :194-195
    msgs, amsgs = [msg], []
    a_last = None
    if depth > 0:
        fb = g.fb_sum() if depth > 1 else None
        for it in range(depth - 1):                                                      # :199
            a_msg = gather_sum(msgs[-1], g.a2a, H)                                       # :201
            new = linear(nA, H, Wh.pk(H, FBOND), w_packed=True, a1=a_msg, k1=H, a2=fb, k2=FBOND, bias=Wh.b, residual=inp, act=ACT_RELU,
                         drop_p=p, seed=_site_seed(seed, 2001 + it), want_bits=True)     # :202-213
            amsgs.append(a_msg)
            msgs.append(new)
        a_last = gather_sum(msgs[-1], g.a2a, H)                                          # :215-216
        hid = linear(nA, H, Wo.pk(Hin, H), w_packed=True, a1=x, k1=Hin, a1_sub=x_sub, a1_sub_idx=x_sub_idx, a2=a_last, k2=H,
                     bias=Wo.b, act=ACT_RELU, drop_p=p,
                     seed=_site_seed(seed, 3000), want_bits=True)                        # :217-219
    else:
        hid = msg
    del inp
    vecs = segment_mean_fwd(hid, g, H, feat, F, out_drop_p, out_seed)                    # :224-238
    return vecs, (msgs, amsgs, a_last, hid)


def mpndiff_backward(g, H: int, depth: int, Wi: LinW, Wh, Wo, p: float, saved, x, x_sub, dvecs, F: int,
                     out_drop_p: float, out_seed: int, x_sub_idx=None):
    """Adjoint of mpndiff_forward -> (d_x [nA,Hin], gWi, gbi, gWh, gbh, gWo, gbo)."""
    msgs, amsgs, a_last, hid = saved
    nA = g.nA
    Hin = x.shape[1]
    ks = 1.0 / (1.0 - p)
    gWi, gbi = Wi.grads()
    gWh, gbh = (Wh.grads() if Wh is not None else (None, None))
    gWo, gbo = (Wo.grads() if Wo is not None else (None, None))
    d_x = None
    fused = depth > 0 and H % 4 == 0
    if fused:
        # hid = drop(relu(.)): the readout's adjoint applies that pattern as it writes (dZ of W_o), so the dX GEMMs over both
        # column segments of W_o ([d_x | d_a]) read a plain operand; the iterations below follow mpn_backward's fused form
        dz_o = segment_mean_bwd(dvecs, g, H, F, out_drop_p, out_seed, mask=hid, mask_scale=ks)
        d_x = linear(nA, Hin, Wo.pk_t(0, Hin), w_packed=True, a1=dz_o, k1=H)
        wgrad(nA, H, dz_o, gWo, dbias=gbo, x1=x, k1=Hin, x1_sub=x_sub, x1_sub_idx=x_sub_idx, x2=a_last, k2=H, side=True)
        d_a, part = linear(nA, H, Wo.pk_t(Hin, Hin + H), w_packed=True, a1=dz_o, k1=H, colsum_w=g.npad)
        top = depth - 1
        cur = gather_sum(d_a, g.a2a_t, H, row0_partial=part, mask=msgs[top], mask_scale=(1.0 if top == 0 else ks))  # neighbour relation is symmetric
        dzs = []
        fb = g.fb_sum() if depth > 1 else None
        for it in reversed(range(depth - 1)):
            dz = cur
            d_a, part = linear(nA, H, Wh.pk_t(0, H), w_packed=True, a1=dz, k1=H, colsum_w=g.npad)
            wgrad(nA, H, dz, gWh, dbias=gbh, x1=amsgs[it], k1=H, x2=fb, k2=FBOND, accumulate=(it != depth - 2), side=True)
            dzs.append(dz)
            cur = gather_sum(d_a, g.a2a_t, H, row0_partial=part, mask=msgs[it], mask_scale=(1.0 if it == 0 else ks),
                             adds=(dzs if it == 0 else ()))
        d_inp = cur
    elif depth > 0:
        d_hid = segment_mean_bwd(dvecs, g, H, F, out_drop_p, out_seed)                    # [nA,H]
        wgrad(nA, H, d_hid, gWo, dbias=gbo, mask=hid, mask_scale=ks, x1=x, k1=Hin, x1_sub=x_sub, x1_sub_idx=x_sub_idx,
              x2=a_last, k2=H, side=True)
        d_x = linear(nA, Hin, Wo.pk_t(0, Hin), w_packed=True, a1=d_hid, k1=H, a_mask=hid, mask_scale=ks)
        d_a = linear(nA, H, Wo.pk_t(Hin, Hin + H), w_packed=True, a1=d_hid, k1=H, a_mask=hid, mask_scale=ks)
        d_msg = gather_sum(d_a, g.a2a_t, H)
        _pad_row_fix(d_msg, d_a, g, H)
        d_inp = None
        fb = g.fb_sum() if depth > 1 else None
        for it in reversed(range(depth - 1)):
            if d_inp is None:                                        # see mpn_backward: fresh dz per iteration
                d_inp = torch.zeros_like(d_msg)
            dz = relu_bwd(d_msg, msgs[it + 1], ks, dz=torch.empty_like(d_msg), acc=d_inp)
            wgrad(nA, H, dz, gWh, dbias=gbh, x1=amsgs[it], k1=H, x2=fb, k2=FBOND, accumulate=(it != depth - 2), side=True)
            d_a = linear(nA, H, Wh.pk_t(0, H), w_packed=True, a1=dz, k1=H)
            d_msg = gather_sum(d_a, g.a2a_t, H)                     # fresh buffer (side-stream readers)
            _pad_row_fix(d_msg, d_a, g, H)
        if d_inp is None:
            d_inp = relu_bwd(d_msg, msgs[0], 1.0)
        else:
            relu_bwd(d_msg, msgs[0], 1.0, acc=d_inp, want_dz=False)
    else:
        d_hid = segment_mean_bwd(dvecs, g, H, F, out_drop_p, out_seed)                    # [nA,H]
        d_inp = relu_bwd(d_hid, msgs[0], ks)                        # hid = drop(relu(inp))
    wgrad(nA, H, d_inp, gWi, dbias=gbi, x1=x, k1=Hin, x1_sub=x_sub, x1_sub_idx=x_sub_idx, side=True)
    if d_x is None:
        d_x = linear(nA, Hin, Wi.pk_t(0, Hin), w_packed=True, a1=d_inp, k1=H)
    else:
        linear(nA, Hin, Wi.pk_t(0, Hin), w_packed=True, a1=d_inp, k1=H, residual=d_x, out=d_x)
    return d_x, gWi, gbi, gWh, gbh, gWo, gbo


def ffn_forward(x, layers: List[LinW], p: float, seed: int, head: int):
    """FFN.forward after its first Dropout (models/base_model.py:32-60): x is already the
    (possibly dropped) input of the first Linear.  Returns (out [M,N], saved)."""
    M = x.shape[0]
    hs = [x]
    for li, L in enumerate(layers[:-1]):
        hs.append(linear(M, L.w.shape[0], L.pk(L.w.shape[1]), w_packed=True, a1=hs[-1], k1=L.w.shape[1], bias=L.b, act=ACT_RELU, drop_p=p,
                         seed=_site_seed(seed, 4000 + li)))
    L = layers[-1]
    raw = linear(M, L.w.shape[0], L.pk(L.w.shape[1]), w_packed=True, a1=hs[-1], k1=L.w.shape[1], bias=L.b)
    out = raw if head == 0 else head_fwd(raw, head)
    return out, (hs, raw)


def ffn_backward(layers: List[LinW], p: float, head: int, saved, dout, need_dx: bool = True, dx_cols: Optional[int] = None):
    """dx_cols: leading input columns whose gradient the caller needs (the readout part; the appended feature
    columns are inputs without gradient) - lets the first layer's dX run on the straight-line kernel (N % 4 == 0)."""
    hs, raw = saved
    M = raw.shape[0]
    ks = 1.0 / (1.0 - p)
    grads = []
    d = dout if head == 0 else head_bwd(dout, raw, head)
    L = layers[-1]
    gw, gb = L.grads()
    wgrad(M, L.w.shape[0], d, gw, dbias=gb, x1=hs[-1], k1=L.w.shape[1], side=True)
    grads.append((gw, gb))
    dx = None
    if len(layers) > 1 or need_dx:
        dx = linear(M, L.w.shape[1], L.pk_t(0, L.w.shape[1]), w_packed=True, a1=d, k1=L.w.shape[0])
    for li in reversed(range(len(layers) - 1)):
        L = layers[li]
        gw, gb = L.grads()
        y = hs[li + 1]                                              # drop(relu(.)) output of this layer
        wgrad(M, L.w.shape[0], dx, gw, dbias=gb, mask=y, mask_scale=ks, x1=hs[li], k1=L.w.shape[1], side=True)
        grads.append((gw, gb))
        if li > 0 or need_dx:
            nin = L.w.shape[1] if (li > 0 or dx_cols is None) else dx_cols
            dx = linear(M, nin, L.pk_t(0, nin), w_packed=True, a1=dx, k1=L.w.shape[0], a_mask=y, mask_scale=ks)
    grads.reverse()
    return dx, grads


# =========================================================================== layer 3
class GatherSumFn(torch.autograd.Function):
    """index_select_ND(source, index).sum(dim=1) as one op (utils.py:176-193).  `index` follows the
    reference convention (0 = padding row, an ordinary row here)."""

    @staticmethod
    def forward(ctx, source, index):
        src = _rowmajor(source.detach(), "source")
        idx = index.to(torch.int32).contiguous()
        ctx.save_for_backward(idx)
        ctx.n_src = src.shape[0]
        return gather_sum(src, idx, src.shape[1])

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        go = _rowmajor(grad_out, "grad")
        K = idx.shape[1] if idx.dim() == 2 else 1
        # d source[s] = sum over the (r, k) with index[r, k] == s of grad[r]: CSR transpose built on the device
        offsets, dest = _transpose_index(idx, ctx.n_src)
        rows = dest if K == 1 else torch.div(dest, K, rounding_mode="floor").to(torch.int32)
        return gather_sum_csr(go, offsets, rows, ctx.n_src, go.shape[1]), None


class IndexSelectNDFn(torch.autograd.Function):
    """index_select_ND itself (utils.py:176-193): source[index] -> [n, K, H], differentiable in `source`
    (autograd's index_select backward is a scatter-add; here a fixed-order CSR segment sum)."""

    @staticmethod
    def forward(ctx, source, index):
        src = _rowmajor(source.detach(), "source")
        n, K = index.shape
        flat = index.reshape(-1, 1).to(torch.int32).contiguous()
        ctx.save_for_backward(flat)
        ctx.n_src = src.shape[0]
        return gather_sum(src, flat, src.shape[1]).view(n, K, src.shape[1])

    @staticmethod
    def backward(ctx, grad_out):
        flat, = ctx.saved_tensors
        H = grad_out.shape[-1]
        go = _rowmajor(grad_out.reshape(-1, H), "grad")
        offsets, dest = _transpose_index(flat, ctx.n_src)
        return gather_sum_csr(go, offsets, dest, ctx.n_src, H), None


class SegmentMeanFn(torch.autograd.Function):
    """Per-molecule mean readout of atom hiddens (models/mpn.py:110-124), differentiable."""

    @staticmethod
    def forward(ctx, x, g, H):
        ctx.g, ctx.H = g, H
        return segment_mean_fwd(_rowmajor(x.detach(), "atom hiddens"), g, H, None, 0, 0.0, 0)

    @staticmethod
    def backward(ctx, dout):
        return segment_mean_bwd(_rowmajor(dout, "grad"), ctx.g, ctx.H, 0, 0.0, 0), None, None


class _WorkspacePool:
    """Workspaces of the step plans (one ~2 GB buffer per step, live from forward to backward).  The steps of an epoch
    differ in size by a few per cent; asking the caching allocator for a new size every step costs a fresh hipMalloc
    (~80-180 ms) whenever no cached block fits.  The pool hands a returned buffer to the next step if it is large enough
    and allocates with 25 % headroom otherwise, so it stops growing after the first steps.  Reuse is stream-ordered like
    the allocator's own: a buffer comes back after its backward was enqueued, and that backward joins the library's
    side streams to the caller's stream before it returns."""
    free: List[torch.Tensor] = []
    allocs = 0               # buffers allocated so far (bench.py: none should fall inside a timed region)

    @classmethod
    def take(cls, nbytes: int, device) -> torch.Tensor:
        best = None
        for i, t in enumerate(cls.free):
            if t.device == device and t.numel() >= nbytes and (best is None or t.numel() < cls.free[best].numel()):
                best = i
        if best is not None:
            return cls.free.pop(best)
        cls.free = [t for t in cls.free if t.device != device]          # too small for this model now: let them go
        cls.allocs += 1
        return torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)

    @classmethod
    def give(cls, t: torch.Tensor) -> None:
        if len(cls.free) < 4:
            cls.free.append(t)


class StepPlan:
    """The whole forward / backward of the model as ONE C-ABI call each (rr_reaction_forward / rr_reaction_backward,
    csrc/plan.hip): the launch sequence below (mpn_forward ... ffn_backward) issued by native code instead of ~140
    ctypes calls, cutting the host's enqueue time per training step from ~4 ms to well under 1 ms.  Same kernels, same
    order, same dropout streams: results are bit-identical to the per-op path (tests/test_gpu_plan.py).  Used when the
    shape qualifies (H % 4 == 0) and no per-launch profiling is requested."""
    enabled = True
    _ws_bytes = {}
    keep_last = False        # tests / embedding readers: remember the last step so saved() can look into its workspace
    last = None
    timing = False           # RR_PLAN_TIME: HIP events around every split-GEMM / gather-sum launch inside the plans (bench.py)

    @staticmethod
    def select_timings(kinds: int = 7, modes: int = 15) -> None:
        """which launches carry events while `timing` is on: bit k of kinds (0 split GEMM, 1 gather-sum, 2 gather-sum with
        epilogue), bit m of modes (GEMM operand mode 0..3) - rr_plan_timing_select"""
        check(lib().rr_plan_timing_select(int(kinds), int(modes)), "rr_plan_timing_select")

    @staticmethod
    def take_timings():
        """Per-launch durations recorded by plan calls issued with `timing = True` (rr_plan_timing_take) as the records
        bench.py aggregates: (kernel key, algorithmic flops, algorithmic bytes, seconds) - same keys and the same byte / flop
        accounting as the per-op wrappers above (linear(), gather_sum())."""
        buf = (_lib.PlanTiming * 4096)()
        n = lib().rr_plan_timing_take(buf, 4096)
        if n < 0:
            check(n, "rr_plan_timing_take")
        out = []
        for r in buf[:n]:
            if r.kind == 0:
                M, N, kk = r.M, r.N, r.k1 + r.k2
                two_src = r.mode == 1 or (r.mode == 2)
                nbytes = 4 * (M * kk * (2 if two_src else 1) + N * kk + M * N * (1 + r.residual + r.c_pre))
                bits_row = int(lib().rr_mask_bits_row_bytes(r.k1)) if r.bits_in else 0
                nbytes += M * bits_row + (M * int(lib().rr_mask_bits_row_bytes(N)) if r.bits_out else 0)
                if r.dz_out:
                    nbytes += 4 * M * r.k1
                nt = 4 if N <= 64 else (10 if N <= 160 else 19)
                ntp = 38 if N > 304 else nt
                if nt == 19 and ntp == 19 and M <= 8192:
                    key = f"linear_split_kernel<19,5,{r.mode},8>"
                else:
                    key = f"linear_split_kernel<{ntp},{nt},{r.mode},{12 if nt == 19 else 8}>"
                out.append((key, 2 * M * N * kk, nbytes, r.us * 1e-6))
            else:
                H, K = r.N, r.k1
                nbytes = 4 * (r.n_src * H + r.M * H + r.M * K)
                if r.kind == 2:
                    nbytes += 4 * r.M * H * r.k2
                    if r.mask:
                        nbytes += r.M * int(lib().rr_mask_bits_row_bytes(H)) if r.bits_in else 4 * r.M * H
                out.append(("gather_sum_epi_kernel" if r.kind == 2 else "gather_sum_kernel", 0, nbytes, r.us * 1e-6))
        return out

    @staticmethod
    def saved(which: int, index: int = 0) -> Optional[torch.Tensor]:
        """A saved activation of the last plan step (rr_reaction_saved_f32; `which` = _lib.RR_SAVED_*) as a strided view
        of its workspace - valid until the step's backward hands the workspace back.  Needs keep_last = True."""
        if StepPlan.last is None:
            raise RuntimeError("StepPlan.saved: no step on record (set StepPlan.keep_last = True before the forward)")
        M, S, keep, ws, out, flags = StepPlan.last
        p, rows, ld = C.c_void_p(), C.c_int64(), C.c_int64()
        check(lib().rr_reaction_saved_f32(C.byref(M), C.byref(S), flags, which, index, C.byref(p), C.byref(rows), C.byref(ld)),
              "rr_reaction_saved_f32")
        if not p.value:
            return None
        off = p.value - ws.data_ptr()
        width = M.H if which not in (_lib.RR_SAVED_VECS, _lib.RR_SAVED_FFN_H) else ld.value
        flat = ws[off:off + rows.value * ld.value * 4].view(torch.float32)
        return flat.view(rows.value, ld.value)[:, :width]

    @staticmethod
    def eligible(st, params) -> bool:
        if not StepPlan.enabled or Profiler.enabled or st["H"] % 4 != 0:
            return False
        if st["depth"] > 16 or st["diff_depth"] > 16 or (len(params) - 12) // 2 > _lib.RR_MAX_FFN:
            return False
        return all(q is None or (q.dtype == torch.float32 and q.is_cuda) for q in params)

    @staticmethod
    def graph_struct(g, need_fb_sum: bool, need_f_bonds: bool = True) -> "_lib.Graph":
        G = _lib.Graph()
        G.nA, G.nB, G.M, G.K, G.Kb = g.nA, g.nB, g.M, g.K, g.b2b_t.shape[1]
        G.f_atoms, G.ld_fa = ptr(g.f_atoms), g.f_atoms.stride(0)
        if need_f_bonds:                                 # (a streamed batch rebuilds f_bonds lazily: only ask when it is read)
            fb = g.f_bonds
            G.f_bonds, G.ld_fb = ptr(fb), fb.stride(0)
        for k in ("a2b", "b2a", "b2revb", "a2a", "a_scope", "b2t", "a2a_t", "atom2mol", "b2b_t", "npad", "npad_b"):
            setattr(G, k, ptr(getattr(g, k)))
        if need_fb_sum:
            fs = g.fb_sum()
            G.fb_sum, G.ld_fbs = ptr(fs), fs.stride(0)
        return G

    @staticmethod
    def build(st, params, out):
        """(Model, Step, keep-alive list) for this call."""
        keep = []

        def lw(w, b):
            L = _lib.LinearW()
            if w is not None:
                w = _rowmajor(w.detach(), "weight")
                keep.append(w)
                L.w, L.out, L.in_, L.ldw = ptr(w), w.shape[0], w.shape[1], w.stride(0)
                if b is not None:
                    b = _rowmajor(b.detach(), "bias")
                    keep.append(b)
                    L.b = ptr(b)
            return L
        M = _lib.Model()
        M.H, M.depth, M.diff_depth, M.head = st["H"], st["depth"], st["diff_depth"], st["head"]
        M.atom_fdim, M.bond_fdim = ATOM_FDIM, FBOND
        M.enc_wi, M.enc_wh, M.enc_wo = lw(params[0], params[1]), lw(params[2], params[3]), lw(params[4], params[5])
        M.dif_wi, M.dif_wh, M.dif_wo = lw(params[6], params[7]), lw(params[8], params[9]), lw(params[10], params[11])
        nf = (len(params) - 12) // 2
        M.n_ffn = nf
        for i in range(nf):
            M.ffn[i] = lw(params[12 + 2 * i], params[13 + 2 * i])
        S = _lib.Step()
        pg, rg = st["p_graph"], st["r"]
        dd, px = st.get("dedup"), st.get("prefix")
        S.p = StepPlan.graph_struct(pg, st["diff_depth"] > 1)
        S.r = StepPlan.graph_struct(rg, False, need_f_bonds=(px is None))   # shared prefix: W_i runs on the distinct reactants
        S.mode = _lib.RR_STEP_PLAIN
        if dd is not None:
            S.mode = _lib.RR_STEP_DEDUP
            S.amap, S.amap_t, S.amap_t_cols = ptr(dd[1]), ptr(dd[2]), dd[2].shape[1]
            keep += [dd[1], dd[2]]
        elif px is not None:
            S.mode = _lib.RR_STEP_PREFIX
            S.u = StepPlan.graph_struct(px[0], False)
            S.bmap, S.bmap_t, S.bmap_t_cols = ptr(px[1]), ptr(px[2]), px[2].shape[1]
            keep += [px[1], px[2]]
        feat = st["feat"]
        S.feat, S.F = ptr(feat), st["F"]
        S.drop_p, S.seed = float(st["p"]), int(st["seed"]) & 0xFFFFFFFFFFFFFFFF
        S.out = ptr(out)
        return M, S, keep

    @staticmethod
    def flags(train: bool = False) -> int:
        return ((0 if SideStream.enabled else _lib.RR_PLAN_NO_SIDE_STREAM) | (0 if AuxStream.enabled else _lib.RR_PLAN_NO_AUX_STREAM) |
                (0 if SplitGemm.enabled else _lib.RR_PLAN_F32_GEMM) | (_lib.RR_PLAN_AUX_BACKWARD if AuxStream.backward else 0) |
                (_lib.RR_PLAN_TRAIN if (train and not os.environ.get("RR_NO_TRAIN_PACK")) else 0) |
                (_lib.RR_PLAN_F16X2_GEMM if (SplitGemm.enabled and SplitGemm.f16) else 0) |
                (0 if FfnChain.enabled else _lib.RR_PLAN_NO_FFN_CHAIN) | (_lib.RR_PLAN_TIME if StepPlan.timing else 0))


class WgradOrder:
    """Which order of a message-passing layer's W_h weight gradient and input-gradient GEMM the backward plan uses
    (RR_PLAN_WGRAD_EARLY).  Both orders run the same kernels on the same values - every gradient is the same bits - but they
    interleave the weight-gradient stream with the main chain differently, and which one is faster depends on the workload:
    over the four BASELINE configurations the step moves by -1.3 % ... +2.4 % (profiles/r05_experiments.txt item 19).  So the
    choice is MEASURED: for the first calls of a workload (hidden size, depths, reactant mode, product bonds in units of 16k)
    the backward alternates the two orders with HIP events around the plan call, and keeps "early" only if its median is
    clearly below "late"'s.  mode: "auto" (default), "late", "early" (RR_WGRAD_ORDER).  Nothing here synchronises: finished
    events are read with query() on a later call."""
    mode = os.environ.get("RR_WGRAD_ORDER", "auto")
    warm = 2                 # calls of a workload before the first sample
    samples = 6              # per order
    margin = 0.004           # "early" must win by this fraction of the median
    _state = {}

    @classmethod
    def reset(cls):
        cls._state = {}

    @classmethod
    def _get(cls, sig):
        st = cls._state.get(sig)
        if st is None:
            st = cls._state[sig] = dict(calls=0, choice=None, times={False: [], True: []}, pending=[])
        return st

    @classmethod
    def _harvest(cls, st):
        keep = []
        for early, e0, e1 in st["pending"]:
            if e1.query():
                st["times"][early].append(e0.elapsed_time(e1))
            else:
                keep.append((early, e0, e1))
        st["pending"] = keep
        if st["choice"] is None and all(len(st["times"][k]) >= cls.samples for k in (False, True)):
            late, early = (sorted(st["times"][k])[len(st["times"][k]) // 2] for k in (False, True))
            st["choice"] = bool(early < late * (1.0 - cls.margin))

    @classmethod
    def begin(cls, sig):
        """-> (early, events or None): the order for this call and, while the workload is still being measured, the pair of
        events to record around the plan call (pass them to end())."""
        if cls.mode != "auto":
            return cls.mode == "early", None
        st = cls._get(sig)
        st["calls"] += 1
        if st["choice"] is not None:
            return st["choice"], None
        cls._harvest(st)
        if st["choice"] is not None:
            return st["choice"], None
        if st["calls"] <= cls.warm:
            return False, None
        n = {k: len(st["times"][k]) + sum(1 for q in st["pending"] if q[0] == k) for k in (False, True)}
        if n[False] >= cls.samples and n[True] >= cls.samples:       # everything is in flight: wait for it in the order in use
            return False, None
        early = n[True] < n[False]
        return early, (early, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    @classmethod
    def end(cls, sig, ev):
        if ev is not None:
            cls._get(sig)["pending"].append(ev)

    @classmethod
    def settled(cls) -> bool:
        """True when no workload seen so far is still being measured (bench.py: finish this before the timed region)."""
        if cls.mode != "auto":
            return True
        for st in cls._state.values():
            if st["choice"] is None:
                cls._harvest(st)
            if st["choice"] is None:
                return False
        return True

    @classmethod
    def choices(cls):
        return {str(k): v["choice"] for k, v in cls._state.items()}


class ReactionModelFn(torch.autograd.Function):
    """ReactionModel.forward (models/base_model.py:150-171) with an explicit backward."""

    @staticmethod
    def forward(ctx, st, *params):
        # params order: enc(Wi w,b, Wh w,b, Wo w,b), diff(Wi w,b, Wh w,b, Wo w,b), ffn (w,b)*
        ctx.plan = None
        if StepPlan.eligible(st, params):
            pg = st["p_graph"]
            n_out = params[-2].shape[0]
            out = torch.empty(pg.M, n_out, dtype=torch.float32, device=pg.device)
            M, S, keep = StepPlan.build(st, params, out)
            nbytes = int(lib().rr_reaction_workspace_bytes(C.byref(M), C.byref(S)))
            if nbytes == 0:
                raise RuntimeError("rr_reaction_workspace_bytes rejected the step (inconsistent model / batch shapes)")
            ws = _WorkspacePool.take(nbytes, pg.device)
            S.workspace, S.workspace_bytes = C.c_void_p(ws.data_ptr()), nbytes
            # the backward must lay the workspace out the same way; when one will come, the forward packs its weights too
            flags = StepPlan.flags(train=any(ctx.needs_input_grad))
            check(lib().rr_reaction_forward(C.byref(M), C.byref(S), flags, stream()), "rr_reaction_forward")
            ctx.plan = (M, S, keep, ws, out, flags)
            if StepPlan.keep_last:
                StepPlan.last = ctx.plan
            if not any(ctx.needs_input_grad):          # (e.g. under torch.no_grad()) no backward will come for this step
                _WorkspacePool.give(ws)
            ctx.st = st
            ctx.param_shapes = [None if q is None else q for q in params]
            ctx.present = [q is not None for q in params]
            return out.reshape(-1) if st["squeeze"] else out
        enc = [LinW(params[0], params[1]), LinW(params[2], params[3]) if params[2] is not None else None,
               LinW(params[4], params[5])]
        dif = [LinW(params[6], params[7]), LinW(params[8], params[9]) if params[8] is not None else None,
               LinW(params[10], params[11]) if params[10] is not None else None]
        ffn = [LinW(params[i], params[i + 1], big=False) for i in range(12, len(params), 2)]
        H, p, seed = st["H"], st["p"], st["seed"]
        rg, pg = st["r"], st["p_graph"]
        dev = rg.device
        main = torch.cuda.current_stream(dev)
        px = st.get("prefix")                                        # (distinct reactant graph, bmap, bmap_t) or None

        def enc_r():
            if px is not None:
                return mpn_forward_shared(px[0], rg, px[1], H, st["depth"], enc[0], enc[1], enc[2], p, _site_seed(seed, 1))
            return mpn_forward(rg, H, st["depth"], enc[0], enc[1], enc[2], p, _site_seed(seed, 1))
        if AuxStream.enabled:
            enc[0].pk(FBOND)                                        # shared packed weights: build once, on main
            if enc[1] is not None:
                enc[1].pk(H)
            enc[2].pk(ATOM_FDIM, H)
            aux = AuxStream.get(dev)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                r_h, r_saved = enc_r()
            p_h, p_saved = mpn_forward(pg, H, st["depth"], enc[0], enc[1], enc[2], p, _site_seed(seed, 2))
            main.wait_stream(aux)
            r_h.record_stream(main)
        else:
            r_h, r_saved = enc_r()
            p_h, p_saved = mpn_forward(pg, H, st["depth"], enc[0], enc[1], enc[2], p, _site_seed(seed, 2))
        dd = st.get("dedup")                                         # (unique reactant graph, amap, amap_t) or None
        vecs, d_saved = mpndiff_forward(pg, H, st["diff_depth"], dif[0], dif[1], dif[2], p, _site_seed(seed, 3),
                                        p_h, r_h, st["feat"], st["F"], out_drop_p=p, out_seed=_site_seed(seed, 4),
                                        x_sub_idx=None if dd is None else dd[1])
        out, f_saved = ffn_forward(vecs, ffn, p, _site_seed(seed, 5), st["head"])
        ctx.st, ctx.mods = st, (enc, dif, ffn)
        ctx.saved = (r_saved, p_saved, d_saved, f_saved, r_h, p_h)
        ctx.n_params = len(params)
        ctx.present = [q is not None for q in params]
        if st["squeeze"]:
            out = out.reshape(-1)
        return out

    @staticmethod
    def backward(ctx, dout):
        st = ctx.st
        if ctx.plan is not None:
            if ctx.plan == "done":
                raise RuntimeError("ReactionModelFn: backward ran twice; the saved activations are released after the "
                                   "first backward (retain_graph=True is not supported by the explicit backward)")
            M, S, keep, ws, out, flags = ctx.plan
            dout = _rowmajor(dout.reshape(out.shape), "grad_output")
            G = _lib.Grads()
            grads = []
            order = [0, 2, 4, 6, 8, 10] + list(range(12, len(ctx.param_shapes), 2))     # weights; bias = index + 1
            for gi, wi in enumerate(order):
                w, b = ctx.param_shapes[wi], ctx.param_shapes[wi + 1]
                gw = None if w is None else _grad_like(w)
                gb = None if b is None else _grad_like(b)
                G.w[gi], G.b[gi] = ptr(gw), ptr(gb)
                grads += [gw, gb]
            sig = (st["H"], st["depth"], st["diff_depth"], int(S.mode), int(st["p_graph"].nB) >> 14)
            early, ev = WgradOrder.begin(sig)
            if ev is not None:
                ev[1].record(torch.cuda.current_stream())
            check(lib().rr_reaction_backward(C.byref(M), C.byref(S), ptr(dout), C.byref(G),
                                             flags | (_lib.RR_PLAN_WGRAD_EARLY if early else 0), stream()), "rr_reaction_backward")
            if ev is not None:
                ev[2].record(torch.cuda.current_stream())
                WgradOrder.end(sig, ev)
            ctx.plan = "done"
            _WorkspacePool.give(ws)
            return (None, *grads)
        enc, dif, ffn = ctx.mods
        if ctx.saved is None:
            raise RuntimeError("ReactionModelFn: backward ran twice; the saved activations are released after the first "
                               "backward (retain_graph=True is not supported by the explicit backward)")
        r_saved, p_saved, d_saved, f_saved, r_h, p_h = ctx.saved
        H, p, seed = st["H"], st["p"], st["seed"]
        rg, pg = st["r"], st["p_graph"]
        dout = _rowmajor(dout.reshape(f_saved[1].shape), "grad_output")
        dvecs, fg = ffn_backward(ffn, p, st["head"], f_saved, dout, need_dx=True, dx_cols=(H if len(ffn) > 1 else None))
        dd = st.get("dedup")
        d_diff, gWi, gbi, gWh, gbh, gWo, gbo = mpndiff_backward(pg, H, st["diff_depth"], dif[0], dif[1], dif[2], p,
                                                                 d_saved, p_h, r_h, dvecs, st["F"], p,
                                                                 _site_seed(seed, 4),
                                                                 x_sub_idx=None if dd is None else dd[1])
        # de-duplicated reactants: d r_h[u] = -(sum over the copies of atom u of d_diff) (fixed-order segment sum)
        d_r = d_diff if dd is None else gather_sum(d_diff, dd[2], H)
        if AuxStream.enabled and AuxStream.backward and st.get("prefix") is None:
            dev = dout.device
            main = torch.cuda.current_stream(dev)
            aux = AuxStream.get(dev)
            if enc[1] is not None:
                enc[1].pk_t(0, H)                                   # shared packed transposes: build once, on main
            enc[2].pk_t(ATOM_FDIM, ATOM_FDIM + H)
            aux.wait_stream(main)
            d_r.record_stream(aux)
            with torch.cuda.stream(aux):
                gr = mpn_backward(rg, H, st["depth"], enc[0], enc[1], enc[2], p, r_saved, d_r, -1.0)
            gp = mpn_backward(pg, H, st["depth"], enc[0], enc[1], enc[2], p, p_saved, d_diff, 1.0)
            main.wait_stream(aux)
            for t in gr:
                if t is not None:
                    t.record_stream(main)
            SideStream.join(dout.device)
            genc = [None if a is None else axpby(1.0, a, 1.0, b, out=a) for a, b in zip(gp, gr)]
        else:
            # the two encoder passes share weights: the reactant pass accumulates into the product pass's buffers
            gp = mpn_backward(pg, H, st["depth"], enc[0], enc[1], enc[2], p, p_saved, d_diff, 1.0)
            px = st.get("prefix")
            if px is not None:
                mpn_backward_shared(px[0], rg, px[2], H, st["depth"], enc[0], enc[1], enc[2], p, r_saved, d_r, -1.0, into=gp)
            else:
                mpn_backward(rg, H, st["depth"], enc[0], enc[1], enc[2], p, r_saved, d_r, -1.0, into=gp)
            genc = list(gp)
        SideStream.join(dout.device)                                # weight gradients are complete from here on
        grads = list(genc) + [gWi, gbi, gWh, gbh, gWo, gbo]
        for gw, gb in fg:
            grads += [gw, gb]
        grads = [gq if present else None for gq, present in zip(grads, ctx.present)]
        ctx.saved = None
        SideStream.join(dout.device)
        return (None, *grads)


class MPNFn(torch.autograd.Function):
    """Standalone MPN encoder (atom hiddens)."""

    @staticmethod
    def forward(ctx, st, wi, bi, wh, bh, wo, bo):
        mods = [LinW(wi, bi), LinW(wh, bh) if wh is not None else None, LinW(wo, bo)]
        h, saved = mpn_forward(st["g"], st["H"], st["depth"], mods[0], mods[1], mods[2], st["p"], st["seed"])
        ctx.st, ctx.mods, ctx.saved = st, mods, saved
        ctx.present = [q is not None for q in (wi, bi, wh, bh, wo, bo)]
        return h

    @staticmethod
    def backward(ctx, dh):
        st, m = ctx.st, ctx.mods
        g = mpn_backward(st["g"], st["H"], st["depth"], m[0], m[1], m[2], st["p"], ctx.saved,
                         _rowmajor(dh, "grad"), 1.0)
        SideStream.join(dh.device)
        return (None, *[gq if pr else None for gq, pr in zip(g, ctx.present)])


class MPNDiffFn(torch.autograd.Function):
    """Standalone MPNDiff (readout vectors, no FFN-input dropout)."""

    @staticmethod
    def forward(ctx, st, x, wi, bi, wh, bh, wo, bo):
        mods = [LinW(wi, bi), LinW(wh, bh) if wh is not None else None, LinW(wo, bo) if wo is not None else None]
        xx = _rowmajor(x.detach(), "atom_features")
        vecs, saved = mpndiff_forward(st["g"], st["H"], st["depth"], mods[0], mods[1], mods[2], st["p"], st["seed"],
                                      xx, None, st["feat"], st["F"])
        ctx.st, ctx.mods, ctx.saved, ctx.x = st, mods, saved, xx
        ctx.present = [q is not None for q in (wi, bi, wh, bh, wo, bo)]
        return vecs

    @staticmethod
    def backward(ctx, dvecs):
        st, m = ctx.st, ctx.mods
        res = mpndiff_backward(st["g"], st["H"], st["depth"], m[0], m[1], m[2], st["p"], ctx.saved, ctx.x, None,
                               _rowmajor(dvecs, "grad"), st["F"], 0.0, 0)
        SideStream.join(dvecs.device)
        return (None, res[0], *[gq if pr else None for gq, pr in zip(res[1:], ctx.present)])


class FFNFn(torch.autograd.Function):
    """Standalone FFN (first Dropout included)."""

    @staticmethod
    def forward(ctx, st, x, *params):
        layers = [LinW(params[i], params[i + 1], big=False) for i in range(0, len(params), 2)]
        xx = _rowmajor(x.detach(), "ffn input")
        p = st["p"]
        xin = dropout(xx, p, _site_seed(st["seed"], 9)) if p > 0 else xx
        out, saved = ffn_forward(xin, layers, p, st["seed"], st["head"])
        ctx.st, ctx.layers, ctx.saved = st, layers, saved
        ctx.present = [q is not None for q in params]
        if st["squeeze"]:
            out = out.reshape(-1)
        return out

    @staticmethod
    def backward(ctx, dout):
        st = ctx.st
        dout = _rowmajor(dout.reshape(ctx.saved[1].shape), "grad_output")
        dx, fg = ffn_backward(ctx.layers, st["p"], st["head"], ctx.saved, dout, need_dx=True)
        if st["p"] > 0:
            dx = dropout(dx, st["p"], _site_seed(st["seed"], 9))
        grads = []
        for gw, gb in fg:
            grads += [gw, gb]
        SideStream.join(dout.device)
        return (None, dx, *[gq if pr else None for gq, pr in zip(grads, ctx.present)])
