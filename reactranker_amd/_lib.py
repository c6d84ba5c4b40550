"""ctypes binding of libreactranker_hip.so (the C-ABI in include/reactranker_hip.h).

This is the only place the shared library is loaded.  There is no fallback: if the
library is missing or a call returns a non-zero status, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RR_LIB_PATH") or os.path.join(_HERE, "csrc", "libreactranker_hip.so")   # override: A/B builds

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p
i64 = C.c_int64
i32 = C.c_int
u64 = C.c_uint64
f32 = C.c_float


class LinearArgs(C.Structure):
    _fields_ = [
        ("M", i64), ("N", i32),
        ("a1", c_f32p), ("lda1", i64), ("k1", i32), ("a1_idx", c_i32p),
        ("a1_sub", c_f32p), ("lda1_sub", i64), ("a1_sub_idx", c_i32p),
        ("a2", c_f32p), ("lda2", i64), ("k2", i32),
        ("a_mask", c_f32p), ("ld_mask", i64), ("mask_scale", f32),
        ("dz_out", c_f32p), ("ld_dz", i64), ("dz_accumulate", i32),
        ("w", c_f32p), ("ldw", i64), ("w_packed", i32),
        ("bias", c_f32p),
        ("residual", c_f32p), ("ldr", i64), ("residual_idx", c_i32p),
        ("act", i32), ("drop_p", f32), ("drop_seed", u64),
        ("c", c_f32p), ("ldc", i64),
        ("c_pre", c_f32p), ("ld_pre", i64),
        ("colsum_w", c_f32p), ("colsum_partial", c_f32p), ("ld_partial", i64),
        ("mask_bits_out", C.c_void_p), ("a_mask_bits", C.c_void_p),
        ("a1_amax", c_f32p), ("a1_sub_amax", c_f32p), ("a2_amax", c_f32p),
        ("c_amax_out", c_f32p), ("dz_amax_out", c_f32p),
    ]


class PackDesc(C.Structure):
    _fields_ = [
        ("src", c_f32p), ("ld_src", i64), ("transpose", i32), ("rows", i32), ("c0", i32), ("k1", i32), ("k2", i32),
        ("dst", C.c_void_p), ("split", i32),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("M", i64), ("N", i32),
        ("dy", c_f32p), ("ld_dy", i64),
        ("mask", c_f32p), ("ld_mask", i64), ("mask_scale", f32),
        ("x1", c_f32p), ("ldx1", i64), ("k1", i32), ("x1_idx", c_i32p),
        ("x1_sub", c_f32p), ("ldx1_sub", i64), ("x1_sub_idx", c_i32p),
        ("x2", c_f32p), ("ldx2", i64), ("k2", i32),
        ("dw", c_f32p), ("ld_dw", i64),
        ("dbias", c_f32p),
        ("accumulate", i32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("split", i32),
        ("dy_amax", c_f32p), ("x1_amax", c_f32p), ("x1_sub_amax", c_f32p), ("x2_amax", c_f32p),
    ]


RR_MAX_GATHER_ADDS = 15
RR_MAX_GATHER_SRCS = 4


class GatherEpi(C.Structure):
    _fields_ = [
        ("mask", c_f32p), ("ld_mask", i64), ("mask_bits", C.c_void_p), ("mask_scale", f32),
        ("n_adds", i32), ("ld_add", i64), ("adds", C.c_void_p * RR_MAX_GATHER_ADDS),
        ("amax_out", c_f32p),
    ]


RR_MAX_FFN = 8
RR_G_FFN0 = 6
RR_STEP_PLAIN, RR_STEP_DEDUP, RR_STEP_PREFIX = 0, 1, 2
RR_PLAN_NO_SIDE_STREAM, RR_PLAN_NO_AUX_STREAM, RR_PLAN_F32_GEMM, RR_PLAN_AUX_BACKWARD, RR_PLAN_TRAIN, RR_PLAN_F16X2_GEMM = 1, 2, 4, 8, 16, 32
RR_PLAN_NO_FFN_CHAIN = 64
RR_PLAN_TIME = 128
RR_PLAN_WGRAD_EARLY = 256


class PlanTiming(C.Structure):
    _fields_ = [
        ("kind", i32), ("mode", i32), ("M", i64), ("n_src", i64), ("N", i32), ("k1", i32), ("k2", i32),
        ("residual", i32), ("c_pre", i32), ("dz_out", i32), ("bits_out", i32), ("bits_in", i32), ("mask", i32), ("us", f32),
    ]


class FfnStage(C.Structure):
    _fields_ = [
        ("w", c_f32p), ("ldw", i64), ("bias", c_f32p), ("n_out", i32), ("n_in", i32),
        ("relu", i32), ("dropout", i32), ("rowdot", i32), ("drop_seed", u64),
        ("out", c_f32p), ("ld_out", i64), ("post_mask", c_f32p), ("ld_mask", i64),
    ]


class FfnChainArgs(C.Structure):
    _fields_ = [
        ("M", i64), ("n_stages", i32), ("x", c_f32p), ("ldx", i64), ("drop_p", f32), ("mask_scale", f32),
        ("stage", FfnStage * RR_MAX_FFN),
    ]


class Graph(C.Structure):
    _fields_ = [
        ("nA", i64), ("nB", i64), ("M", i64), ("K", i32), ("Kb", i32),
        ("f_atoms", c_f32p), ("ld_fa", i64), ("f_bonds", c_f32p), ("ld_fb", i64),
        ("a2b", c_i32p), ("b2a", c_i32p), ("b2revb", c_i32p), ("a2a", c_i32p), ("a_scope", c_i32p), ("b2t", c_i32p),
        ("a2a_t", c_i32p), ("atom2mol", c_i32p), ("b2b_t", c_i32p),
        ("npad", c_f32p), ("npad_b", c_f32p),
        ("fb_sum", c_f32p), ("ld_fbs", i64),
    ]


class LinearW(C.Structure):
    _fields_ = [("w", c_f32p), ("b", c_f32p), ("out", i32), ("in_", i32), ("ldw", i64)]


class Model(C.Structure):
    _fields_ = [
        ("H", i32), ("depth", i32), ("diff_depth", i32), ("n_ffn", i32), ("head", i32), ("atom_fdim", i32), ("bond_fdim", i32),
        ("enc_wi", LinearW), ("enc_wh", LinearW), ("enc_wo", LinearW),
        ("dif_wi", LinearW), ("dif_wh", LinearW), ("dif_wo", LinearW),
        ("ffn", LinearW * RR_MAX_FFN),
    ]


class Step(C.Structure):
    _fields_ = [
        ("p", Graph), ("r", Graph), ("u", Graph), ("mode", i32),
        ("amap", c_i32p), ("amap_t", c_i32p), ("amap_t_cols", i32),
        ("bmap", c_i32p), ("bmap_t", c_i32p), ("bmap_t_cols", i32),
        ("feat", c_f32p), ("F", i32),
        ("drop_p", f32), ("seed", u64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("out", c_f32p),
    ]


class Grads(C.Structure):
    _fields_ = [("w", c_f32p * (RR_G_FFN0 + RR_MAX_FFN)), ("b", c_f32p * (RR_G_FFN0 + RR_MAX_FFN))]


_SIGS = {
    "rr_strerror": (C.c_char_p, [i32]),
    "rr_version": (i32, []),
    "rr_abi_struct_sizes": (None, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rr_gather_sum_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, i32, c_f32p, i64, c_stream]),
    "rr_abi_gather_epi_size": (C.c_size_t, []),
    "rr_gather_sum_epi_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, i32, c_f32p, i64, i64, C.POINTER(GatherEpi), c_f32p, i64,
                                    c_stream]),
    "rr_segment_mean_bwd_masked_f32": (i32, [c_f32p, i64, c_i32p, c_i32p, i64, i32, i32, f32, u64, c_f32p, i64, C.c_void_p, f32,
                                             c_f32p, i64, c_stream]),
    "rr_segment_mean_bwd_masked_amax_f32": (i32, [c_f32p, i64, c_i32p, c_i32p, i64, i32, i32, f32, u64, c_f32p, i64, C.c_void_p, f32,
                                             c_f32p, i64, c_f32p, c_stream]),
    "rr_gather_sum_masked_f32": (i32, [c_f32p, c_f32p, i64, i64, c_i32p, i64, i32, i32, f32, c_f32p, i64, c_stream]),
    "rr_gather_sum_dropmask_f32": (i32, [c_f32p, i64, i64, c_f32p, i64, c_i32p, i64, i32, i32, f32, u64, f32, c_f32p, i64, c_stream]),
    "rr_gather_sum_csr_f32": (i32, [c_f32p, i64, i64, c_i32p, c_i32p, i64, i32, c_f32p, i64, c_stream]),
    "rr_build_fbonds_f32": (i32, [c_f32p, i64, i64, i32, c_i32p, c_f32p, i64, i32, i64, c_f32p, i64, c_stream]),
    "rr_gather_diff_f32": (i32, [c_f32p, i64, i64, c_i32p, c_f32p, i64, i64, c_i32p, i64, i32, c_f32p, i64, c_stream]),
    "rr_gather_dropout_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, f32, u64, c_f32p, i64, c_stream]),
    "rr_gather_dropout_amax_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, f32, u64, c_f32p, i64, c_f32p, c_stream]),
    "rr_colsum_workspace_bytes": (C.c_size_t, [i64, i32]),
    "rr_weighted_colsum_f32": (i32, [c_f32p, i64, i64, c_f32p, i32, c_f32p, i32, C.c_void_p, C.c_size_t, c_stream]),
    "rr_linear_f32": (i32, [C.POINTER(LinearArgs), c_stream]),
    "rr_linear_colsum_rows": (i64, [i64]),
    "rr_gather_sum_multi_f32": (i32, [C.POINTER(C.c_void_p), i32, i64, i64, c_i32p, i64, i32, i32, c_f32p, i64, c_stream]),
    "rr_gather_sum_padrow_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, i32, c_f32p, i64, i64, c_f32p, i64, c_stream]),
    "rr_gather_sum_amax_f32": (i32, [c_f32p, i64, i64, c_i32p, i64, i32, i32, c_f32p, i64, i64, c_f32p, i64, c_f32p, c_stream]),
    "rr_packed_weight_ld": (i64, [i32, i32]),
    "rr_mask_bits_row_bytes": (i64, [i32]),
    "rr_split_weight_bytes": (C.c_size_t, [i32, i32, i32]),
    "rr_pack_weight_f32": (i32, [c_f32p, i64, i32, i32, i32, i32, i32, c_f32p, c_stream]),
    "rr_pack_weights_f32": (i32, [C.c_void_p, i32, c_stream]),
    "rr_linear_wgrad_workspace_bytes": (C.c_size_t, [i64, i32, i32]),
    "rr_linear_wgrad_f32": (i32, [C.POINTER(WgradArgs), c_stream]),
    "rr_amax_f32": (i32, [c_f32p, i64, i32, i64, c_f32p, c_stream]),
    "rr_dropout_keep_host": (i32, [u64, u64, f32]),
    "rr_dropout_f32": (i32, [c_f32p, i64, f32, u64, c_f32p, c_stream]),
    "rr_relu_bwd_f32": (i32, [c_f32p, c_f32p, f32, c_f32p, c_f32p, i64, c_stream]),
    "rr_relu_bwd_sum_f32": (i32, [c_f32p, c_f32p, f32, C.POINTER(C.c_void_p), i32, c_f32p, i64, c_stream]),
    "rr_axpby_f32": (i32, [f32, c_f32p, f32, c_f32p, c_f32p, i64, c_stream]),
    "rr_head_fwd_f32": (i32, [c_f32p, i64, i32, i32, c_f32p, c_stream]),
    "rr_head_bwd_f32": (i32, [c_f32p, c_f32p, i64, i32, i32, c_f32p, c_stream]),
    "rr_segment_mean_fwd_f32": (i32, [c_f32p, i64, c_i32p, i64, i32, c_f32p, i32, f32, u64, c_f32p, i64, c_stream]),
    "rr_segment_mean_bwd_f32": (i32, [c_f32p, i64, c_i32p, c_i32p, i64, i32, i32, f32, u64, c_f32p, i64, c_stream]),
    "rr_listmle_fwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, c_stream]),
    "rr_listmle_bwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, i64, c_stream]),
    "rr_listmle_step_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, C.c_void_p, c_f32p, i64, c_stream]),
    "rr_listnet_step_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, i64, c_f32p, c_f32p, C.c_void_p, c_f32p, i64, c_stream]),
    "rr_evidential_ranking_step_f32": (i32, [c_f32p, c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, C.c_void_p, c_f32p,
                                             c_f32p, i64, c_stream]),
    "rr_listnet_fwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, i64, c_f32p, c_f32p, c_stream]),
    "rr_listnet_bwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, i64, c_f32p, c_f32p, i64, c_stream]),
    "rr_evidential_ranking_fwd_f32": (i32, [c_f32p, c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, c_stream]),
    "rr_evidential_ranking_bwd_f32": (i32, [c_f32p, c_f32p, i64, c_f32p, c_i32p, i32, i32, c_f32p, c_f32p, c_f32p,
                                            i64, c_stream]),
    "rr_ranknet_fwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, f32, c_f32p, C.c_void_p, c_f32p, c_stream]),
    "rr_ranknet_bwd_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, f32, i32, c_f32p, c_f32p, i64, c_stream]),
    "rr_pointwise_partial_count": (i64, [i64]),
    "rr_mse_fwd_f32": (i32, [c_f32p, i64, c_f32p, i64, c_f32p, c_f32p, c_stream]),
    "rr_mse_bwd_f32": (i32, [c_f32p, i64, c_f32p, i64, c_f32p, c_f32p, i64, c_stream]),
    "rr_gauss_nll_fwd_f32": (i32, [c_f32p, c_f32p, i64, c_f32p, i64, c_f32p, c_f32p, c_stream]),
    "rr_gauss_nll_bwd_f32": (i32, [c_f32p, c_f32p, i64, c_f32p, i64, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "rr_ranking_metrics_f32": (i32, [c_f32p, i64, c_f32p, c_i32p, i32, i32, C.c_double, C.c_double, c_i32p, C.c_void_p,
                               c_stream]),
    "rr_logcumsumexp_fwd_f32": (i32, [c_f32p, i32, c_f32p, c_stream]),
    "rr_logcumsumexp_bwd_f32": (i32, [c_f32p, c_f32p, c_f32p, i32, c_f32p, c_stream]),
    "rr_pack_sizes": (i32, [C.c_void_p, C.c_void_p, i64, C.c_void_p, i32, C.POINTER(i64), C.POINTER(i64),
                            C.POINTER(C.c_int32)]),
    "rr_pack_graphs": (i32, [C.c_void_p, C.c_void_p, i64, C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, i32, C.c_void_p, i64, C.c_void_p, i64, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "rr_abi_plan_struct_sizes": (None, [C.POINTER(C.c_size_t)] * 4),
    "rr_abi_ffn_chain_size": (C.c_size_t, []),
    "rr_plan_timing_take": (i32, [C.POINTER(PlanTiming), i32]),
    "rr_plan_timing_select": (i32, [i32, i32]),
    "rr_ffn_chain_f32": (i32, [C.POINTER(FfnChainArgs), c_stream]),
    "rr_reaction_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(Step)]),
    "rr_reaction_forward": (i32, [C.POINTER(Model), C.POINTER(Step), i32, c_stream]),
    "rr_reaction_backward": (i32, [C.POINTER(Model), C.POINTER(Step), c_f32p, C.POINTER(Grads), i32, c_stream]),
    "rr_reaction_saved_f32": (i32, [C.POINTER(Model), C.POINTER(Step), i32, i32, i32, C.POINTER(C.c_void_p), C.POINTER(i64),
                                    C.POINTER(i64)]),
    "rr_adam_step_f32": (i32, [C.c_void_p, i32, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_stream]),
    "rr_comm_backend": (i32, []),
    "rr_comm_unique_id": (i32, [C.c_void_p]),
    "rr_comm_init_rank": (i32, [C.POINTER(C.c_void_p), i32, C.c_void_p, i32]),
    "rr_comm_destroy": (i32, [C.c_void_p]),
    "rr_allreduce_f32": (i32, [c_f32p, i64, f32, C.c_void_p, c_stream]),
    "rr_allreduce_rsag_f32": (i32, [c_f32p, i64, f32, C.c_void_p, c_stream]),
    "rr_derive_bond_tables": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, i64, i64, i32, i32,
                                    C.c_void_p, C.c_void_p]),
    "rr_derive_tables": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, i64, i64, i32, C.c_void_p, i64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))
class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


RR_MAX_ADAM = 64
RR_AMAX_LANES, RR_AMAX_STRIDE = 16, 32
RR_AMAX_FLOATS = RR_AMAX_LANES * RR_AMAX_STRIDE
ABI_VERSION = 8
(RR_SAVED_R_MSG, RR_SAVED_R_H, RR_SAVED_P_MSG, RR_SAVED_P_H, RR_SAVED_D_MSG, RR_SAVED_D_HID, RR_SAVED_VECS, RR_SAVED_FFN_H,
 RR_SAVED_R_MSG0_U, RR_SAVED_R_Z1_U) = range(10)

_lib = None


def lib():
    """The loaded shared library; raises loudly (no fallback) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"reactranker_amd: native library not found at {LIB_PATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C reactranker_amd/csrc`. "
                "There is no CPU/eager fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)           # AttributeError here = ABI mismatch, also loud
            fn.restype = res
            fn.argtypes = args
        if l.rr_version() != ABI_VERSION:
            raise RuntimeError(f"reactranker_amd: ABI version mismatch ({l.rr_version()} != {ABI_VERSION}); rebuild")
        sl, sw = C.c_size_t(), C.c_size_t()
        l.rr_abi_struct_sizes(C.byref(sl), C.byref(sw))
        if sl.value != C.sizeof(LinearArgs) or sw.value != C.sizeof(WgradArgs) or l.rr_abi_gather_epi_size() != C.sizeof(GatherEpi):
            raise RuntimeError("reactranker_amd: ctypes struct layout differs from the compiled header")
        sz = [C.c_size_t() for _ in range(4)]
        l.rr_abi_plan_struct_sizes(*[C.byref(x) for x in sz])
        if [x.value for x in sz] != [C.sizeof(Graph), C.sizeof(Model), C.sizeof(Step), C.sizeof(Grads)]:
            raise RuntimeError("reactranker_amd: ctypes plan struct layout differs from the compiled header")
        if l.rr_abi_ffn_chain_size() != C.sizeof(FfnChainArgs):
            raise RuntimeError("reactranker_amd: ctypes rr_ffn_chain_args layout differs from the compiled header")
        _lib = l
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().rr_strerror(status).decode()
        raise RuntimeError(f"reactranker_hip {what} failed: {msg} (status {status})")


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def np_ptr(a):
    if a is None:
        return None
    return C.c_void_p(a.ctypes.data)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"reactranker_amd: `{name}` must live on the GPU (got {t.device}); "
                           "the HIP path has no CPU fallback")
