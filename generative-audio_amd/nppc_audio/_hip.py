"""ctypes binding of libnppc_hip.so (include/nppc_hip.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  Importing this module
without the built library, or calling an op without a HIP device, raises RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NPPC_HIP_LIB") or os.path.join(_HERE, "libnppc_hip.so")   # (NPPC_HIP_LIB: diagnostic builds, tools/diag)

PREC_BF16 = 0
PREC_F32 = 1
_ERR = {1: "bad argument", 2: "kernel launch failed", 3: "unsupported shape/configuration"}

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python generative-audio_amd/build_ext.py` "
                "(the NPPC-audio hot path has no non-HIP fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("nppc_audio (MI355X build) needs a HIP device: the hot path is HIP-only")


def ptr(t):
    if t is None:
        return c_p(0)
    assert t.is_cuda and t.is_contiguous(), "HIP ops take contiguous device tensors"
    return c_p(t.data_ptr())


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr_array(tensors):
    """host array of device pointers (for entry points that take `const float* const*`)"""
    for t in tensors:
        assert t.is_cuda and t.is_contiguous()
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def mid_bwd_part_elems(B, C, Tp, batch):
    n = c_l()
    call("nppc_tcn_mid_bwd_part_elems", B, C, Tp, batch, ctypes.byref(n))
    return n.value


def conv_thin_part_elems():
    n = c_l()
    call("nppc_conv_thin_part_elems", ctypes.byref(n))
    return n.value


def dtype_of(prec):
    return torch.bfloat16 if prec == PREC_BF16 else torch.float32


P, I, L, F = c_p, c_i, c_l, c_f
D = ctypes.c_double
PL = ctypes.POINTER(c_l)
PI = ctypes.POINTER(c_i)

# argtypes of every entry point of include/nppc_hip.h (tests assert the two stay in sync)
SIGS = {
    "nppc_lstm2_packed_elems": [I, I, PL, PL, PI],
    "nppc_lstm2_pack_weights": [I, P, P, P, P, P, P, P, P, I, I, P, P, P, P, P],
    "nppc_lstm2_fwd": [I, I, I, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, P],
    "nppc_lstm2_coop_plan": [I, I, L, I, I, PI, PI, PI],
    "nppc_lstm2_fwd_coop": [I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, I, P],
    "nppc_lstm2_fwd_coop_head": [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, I, P, P, I, I, P],
    "nppc_sb_head_finalize": [P, I, P, P, L, I, I, I, I, P],
    "nppc_lstm2_ws_plan": [I, L, I, I, I, PI, PI],
    "nppc_lstm2_ws_packed_elems": [PL, PL],
    "nppc_lstm2_ws_pack": [P, P, P, P, I, P, P, P],
    "nppc_lstm2_fwd_ws": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, L, I, I, I, P],
    "nppc_lstm2_coop_bwd_packed_elems": [PL],
    "nppc_lstm2_coop_bwd_pack": [P, P, P, P, I, P, P, P],
    "nppc_lstm2_bwd_coop": [P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, P],
    "nppc_lstm2_coop_bwd2_packed_elems": [PL],
    "nppc_lstm2_coop_bwd2_pack": [P, P, P, P, I, P, P, P],
    "nppc_lstm2_bwd_coop2": [P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, P],
    "nppc_lstm2_bwd_coop2_head": [P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, P],
    "nppc_lstm2_coop_bwd4_sizes": [L, PL, PL, PL],
    "nppc_lstm2_coop_bwd4_pack": [P, P, P, P, I, P, P, P],
    "nppc_lstm2_bwd_coop4": [P, P, P, P, P, P, P, P, P, P, P, P, P, L, P, L, I, I, P],
    "nppc_head_dy_gather": [P, P, L, I, I, I, I, P],
    "nppc_sb_head_bwd_w": [I, P, P, P, P, L, I, I, I, I, I, P],
    "nppc_lstm2_bwd_packed_elems": [I, I, PL, PL],
    "nppc_lstm2_pack_weights_bwd": [I, P, P, P, P, I, I, P, P, P],
    "nppc_lstm2_bwd": [I, P, P, P, P, P, P, P, P, P, P, L, I, I, I, P],
    "nppc_stft": [P, P, P, P, I, I, I, I, P],
    "nppc_dropband": [P, P, I, I, I, I, I, P],
    "nppc_cirm_build_compress": [P, P, P, P, P, I, I, I, I, F, P],
    "nppc_cirm_decompress_apply_conj": [P, P, P, P, P, P, P, I, I, I, P],
    "nppc_cirm_decompress_apply": [P, P, P, P, P, P, P, I, I, I, P],
    "nppc_istft": [P, P, P, I, I, I, I, I, P],
    "nppc_rowsum": [P, P, L, I, P],
    "nppc_tsse_fwd": [P, P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "nppc_tsse_fwd_maps": [I, P, I, P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, L, P, P, P, P, P, P, P, L, I, I, I, I, I, I, P],
    "nppc_tsse_bwd_maps": [I, P, L, P, I, P, P, P, P, I, I, I, P, P, P, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                           I, I, I, I, I, I, P],
    "nppc_scale_transpose": [I, P, P, P, I, I, I, I, I, I, P],
    "nppc_gemm_nt": [I, I, P, L, L, P, L, L, P, L, L, P, L, P, L, L, P, L, P, L, I, I, I, I, I, I, I, I, I, P],
    "nppc_tcn_mid_bwd_part_elems": [I, I, I, I, PL],
    "nppc_tcn_mid_bwd": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P,
                         I, I, I, I, I, F, L, L, L, I, I, P],
    "nppc_tcn_mid_bwd_finish": [P, L, P, L, I, I, I, P, P, P, P, P, P, P, P, P, P, I, I, L, L, I, I, P],
    "nppc_gemm_nt_colsum": [I, I, P, L, L, P, L, L, P, L, L, P, L, P, L, L, I, I, I, I, I, I, I, P, P],
    "nppc_gemm_nt_gn": [I, P, L, L, P, L, L, P, L, L, P, P, L, P, L, L, P, L, D, F, I, I, I, I, I, I, I, P],
    "nppc_tcn_pack_sconv": [I, P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, P],
    "nppc_gemm_nt_splitk": [I, P, L, P, L, P, L, I, I, L, I, P],
    "nppc_gemm_tn_splitk": [P, L, P, L, P, L, I, I, L, I, P],
    "nppc_gemm_tn_splitk_rowsum": [P, L, P, L, P, L, I, I, L, I, P, P],
    "nppc_gemm_tn_splitk2": [P, L, P, L, I, P, L, I, P, L, I, L, I, P, P],
    "nppc_gemm_tn_splitk_batched": [P, L, L, P, L, L, P, L, L, I, I, L, I, I, P],
    "nppc_reduce_slabs_t": [P, I, L, L, P, L, I, I, L, L, I, P],
    "nppc_gemm_tn_splitk_taps": [P, L, P, L, P, L, I, I, L, I, I, I, P],
    "nppc_tcn_dwconv": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, L, L, L, I, P],
    "nppc_tcn_gn_apply": [I, P, P, P, P, P, I, I, I, I, I, F, L, L, L, I, P],
    "nppc_pack_matrix": [I, P, P, I, I, I, I, I, P],
    "nppc_pack_matrix_batched": [I, P, P, I, I, I, I, I, I, I, L, L, L, L, P],
    "nppc_subband_mean": [I, P, I, P, I, L, P, P, P, I, I, I, I, I, P],
    "nppc_subband_stage": [I, P, I, P, I, L, P, P, I, I, I, I, I, I, I, I, P],
    "nppc_sb_head": [I, P, P, P, P, L, I, I, I, I, I, P],
    "nppc_gram": [P, P, P, P, P, I, I, L, P],
    "nppc_combine": [P, P, P, P, P, P, P, I, I, L, P],
    "nppc_gs_solve": [P, P, P, I, I, I, P],
    "nppc_gs_bwd_solve": [P, P, P, P, I, I, I, P],
    "nppc_loss_solve": [P, P, P, P, P, P, P, P, P, P, I, I, P],
    "nppc_loss_bwd_coef": [P, P, P, F, F, P, I, I, P],
    "nppc_loss_bwd_coef_dev": [P, P, P, P, F, F, P, I, I, P],
    "nppc_transpose": [I, P, P, I, I, L, L, L, L, I, I, P],
    "nppc_colsum": [I, P, P, I, I, L, L, L, I, P, L, P],
    "nppc_colsum_scratch_elems": [I, I, I, PL],
    "nppc_sb_head_bwd": [I, P, P, P, P, P, P, L, I, I, I, I, I, P],
    "nppc_subband_stage_bwd": [I, P, P, P, P, P, P, I, I, I, I, I, L, I, I, I, P],
    "nppc_reduce_slabs": [P, I, L, L, P, L, I, I, I, I, I, L, L, I, P],
    "nppc_adam_step": [P, P, P, P, L, D, D, D, D, D, I, D, P],
    "nppc_adam_step_guarded": [P, P, P, P, L, D, D, D, D, D, I, D, P, I, P, P],
    "nppc_loss_solve_eps": [P, P, P, P, P, P, P, P, P, P, I, I, D, I, P],
    "nppc_loss_solve_obj": [P, P, P, P, P, P, P, P, P, P, I, I, D, I, F, P, P],
    "nppc_mix_snr": [P, P, P, F, P, P, P, I, I, P],
    "nppc_inpaint_prepare": [P, P, P, I, F, P, P, P, I, I, I, I, I, P],
    "nppc_time_to_spec_mask": [P, P, I, I, I, I, I, I, P],
    "nppc_stft_pair": [P, P, P, P, I, I, I, I, P],
    "nppc_dropout": [I, P, L, L, I, F, L, I, P, P],
    "nppc_pca_work_elems": [I, I, I, PL],
    "nppc_pca_batch": [P, I, I, I, I, P, P, P, P, P, P, P],
    "nppc_pair_loss": [P, P, P, P, P, P, P, P, P, I, I, L, D, P],
    "nppc_pair_loss_bwd": [P, P, P, P, F, F, P, I, I, L, P],
    "nppc_metric_rows": [P, P, P, P, P, L, P],
    "nppc_rows_gram": [P, I, P, I, L, P, P],
    "nppc_sumsq": [P, L, P, P],
    "nppc_adam_step_clip": [P, P, P, P, L, D, D, D, D, D, I, D, P, D, P],
    "nppc_logmag": [P, P, L, I, L, P, P],
    "nppc_standardize": [P, P, L, I, L, P, P, P],
    "nppc_unet_stage_map": [I, P, L, P, L, I, I, I, I, P],
    "nppc_conv_pack": [I, P, P, P, I, I, I, I, I, I, I, P],
    "nppc_conv_fwd": [I, P, L, P, P, L, P, P, P, F, I, I, I, I, I, I, I, P],
    "nppc_conv_fwd_stats": [I, P, L, P, P, L, P, I, I, I, I, I, I, I, P, P],
    "nppc_bn_stats_from_parts": [P, I, I, I, I, I, P, P, P],
    "nppc_conv_wgrad": [I, P, L, P, L, P, I, I, I, I, I, I, I, P],
    "nppc_conv_thin_part_elems": [PL],
    "nppc_conv3x3_thin_fwd": [I, P, L, P, P, P, P, F, P, L, I, I, I, I, I, P],
    "nppc_conv3x3_thin_wgrad": [I, P, L, P, L, P, P, I, I, I, I, I, P],
    "nppc_conv1x1_thin_fwd": [I, P, L, P, P, P, L, I, I, I, I, I, P],
    "nppc_conv1x1_thin_bwd_data": [I, P, L, P, P, L, I, I, I, I, I, P],
    "nppc_conv1x1_thin_wgrad": [I, P, L, P, L, P, P, I, I, I, I, I, P],
    "nppc_conv_wgrad_transposed": [I, I],
    "nppc_conv_wgrad_reduce": [I, P, I, I, I, P, I, I, I, P],
    "nppc_bn_stats": [I, P, L, L, I, P, P],
    "nppc_bn_finalize": [P, P, P, P, P, P, I, D, F, F, I, P],
    "nppc_bn_act": [I, P, L, P, L, P, I, I, I, I, F, P],
    "nppc_bn_bwd": [I, P, L, P, L, P, L, P, L, P, P, P, L, P, P, I, I, I, I, F, P],
    "nppc_maxpool2": [I, P, L, P, L, P, I, I, I, I, P],
    "nppc_maxpool2_bwd": [I, P, L, P, P, L, I, I, I, I, P],
    "nppc_upsample2": [I, P, L, P, L, I, I, I, I, I, I, P],
    "nppc_upsample2_bwd": [I, P, L, P, L, I, I, I, I, I, I, P],
    "nppc_unet_out": [I, P, L, P, P, L, P, L, I, I, I, I, I, P],
    "nppc_unet_out_bwd": [I, P, L, P, P, L, I, I, I, I, P],
    "nppc_tcn_gn_bwd": [I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, L, L, L, I, P],
    "nppc_tcn_dwconv_bwd": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, L, L, L, I, P],
    "nppc_tsse_bwd_ws_elems": [I, I, I, I, I, I, PL],
    "nppc_tsse_bwd": [I, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                      I, I, I, I, I, I, I, P],
}
_bound = set()
# bench.py sets this to a list to collect (entry point, start_event, end_event) around EVERY launch of an untimed pass;
# the events are recorded on the current stream, i.e. the stream the kernel is enqueued on
PROFILE = None


def call(name, *args):
    """lib().<name>(*args); tensors -> device pointers; raises RuntimeError on a non-zero return."""
    fn = getattr(lib(), name)
    if name not in _bound:
        fn.argtypes = SIGS[name]
        fn.restype = c_i
        _bound.add(name)
    conv = [ptr(a) if (isinstance(a, torch.Tensor) or a is None) else a for a in args]
    if PROFILE is None:
        check(fn(*conv), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*conv), name)
    e1.record()
    PROFILE.append((name, e0, e1))


def colsum(prec, M, out, rows, cols, ld, sM, sOut, batch, scratch_for, s):
    """nppc_colsum with its scratch: scratch_for(n_floats) -> a float32 device tensor owned by the caller (one per
    concurrently running call)"""
    n = ctypes.c_long()
    call("nppc_colsum_scratch_elems", rows, cols, batch, ctypes.byref(n))
    sc = scratch_for(n.value)
    call("nppc_colsum", prec, M, out, rows, cols, ld, sM, sOut, batch, sc, sc.numel(), s)
