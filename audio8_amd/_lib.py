"""ctypes binding of liba8hip.so (the C ABI declared in include/a8hip.h).

The product path has no fallback: if the HIP library is missing or a call fails, a RuntimeError
is raised (never a silent PyTorch/CPU substitute).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("A8HIP_LIB") or os.path.join(_HERE, "liba8hip.so")      # (A8HIP_LIB: developer A/B of two builds)

F32, BF16 = 0, 1
NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_MUL_GELU_GRAD, ACT_ADD = 0, 1, 2, 3

_lib = None
_dev_index = None

# ctypes signatures of every entry point of include/a8hip.h (generated from the header): with argtypes declared,
# plain Python ints / floats / None are marshalled directly (no per-call c_void_p objects).
ARGTYPES = {
    "a8_version": [],
    "a8_gemm": [C.c_void_p, C.c_void_p],
    "a8_gemm_grouped": [C.c_void_p, C.c_int, C.c_void_p],
    "a8_encoder_layer_fwd": [C.c_void_p, C.c_void_p],
    "a8_encoder_layer_bwd": [C.c_void_p, C.c_void_p],
    "a8_encoder_layer_bwd_full": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "a8_set_tuning": [C.c_int, C.c_int],
    "a8_get_tuning": [C.c_int, C.c_void_p],
    "a8_delay_us": [C.c_float, C.c_void_p],
    "a8_prepare_weights": [C.c_int, C.c_void_p, C.c_int, C.c_void_p],
    "a8_prepare_weights_device": [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_conv0_gn_gelu_fwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "a8_conv0_gn_gelu_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_col2im_gelu_grad": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_layernorm_fwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p],
    "a8_layernorm_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p],
    "a8_layernorm_bwd_partial_rows": [C.c_int],
    "a8_colsum_partials_reduce": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "a8_cast": [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "a8_ctc_greedy_decode": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p],
    "a8_dropout": [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p],
    "a8_mask_fill_rows": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_mask_fill_rows_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_mask_zero_cols": [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_zero_rows": [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_gather_rows": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_scatter_add_rows": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_add": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "a8_posconv_fwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_posconv_wgrad": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_loss_combine": [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "a8_loss_combine_bwd": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p],
    "a8_weightnorm_pack": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_weightnorm_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_gelu_bwd_bias": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_attn_fwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint32, C.c_void_p],
    "a8_attn_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint32, C.c_void_p],
    "a8_vq_select": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "a8_zero": [C.c_void_p, C.c_int64, C.c_void_p],
    "a8_conv_wgrad_fold": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_vq_gather": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_vq_ppl": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "a8_vq_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_contrastive_fwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_contrastive_bwd": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_ctc_pack_targets": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_host_negative_indices": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_ctc_fwd_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_log_softmax_fwd": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_log_softmax_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "a8_sumsq": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "a8_adamw": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p],
}


class GemmDesc(C.Structure):
    _fields_ = [
        ("size", C.c_int),                       # sizeof(a8_gemm_desc): GEMM_DESC_SIZE below
        ("dtype", C.c_int), ("layout", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_rpb", C.c_int64), ("a_bs", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_rpb", C.c_int64), ("b_bs", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_rpb", C.c_int64), ("c_bs", C.c_int64),
        ("bias", C.c_void_p), ("bias_rows", C.c_int),
        ("act", C.c_int),
        ("zin", C.c_void_p), ("zout", C.c_void_p), ("ldz", C.c_int64), ("z_rpb", C.c_int64), ("z_bs", C.c_int64),
        ("out_f32", C.c_int), ("accumulate", C.c_int), ("splitk", C.c_int),
        ("alpha", C.c_float),
        ("colsum", C.c_void_p),
    ]


class EncoderLayerDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("size", "dtype", "B", "T", "H", "d", "dff")] + [("p", C.c_float), ("eps", C.c_float)] + \
               [(n, C.c_uint32) for n in ("seed_attn", "seed1", "seed2")] + [("b0", C.c_int)] + \
               [(n, C.c_void_p) for n in ("x", "keymask", "wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "g1", "be1", "g2", "be2",
                                          "qkv", "o", "ao", "x1", "z1", "h", "f", "x2", "lse", "mean2", "rstd2", "mean1", "rstd1")]


class EncoderLayerBwdDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("size", "dtype", "B", "T", "H", "d", "dff")] + [("p", C.c_float)] + \
               [(n, C.c_uint32) for n in ("seed_attn", "seed1", "seed2")] + [("b0", C.c_int)] + \
               [(n, C.c_void_p) for n in ("dx2", "x", "qkv", "o", "ao", "x1", "z1", "f", "keymask", "lse", "mean2", "rstd2", "mean1", "rstd1", "g1", "g2",
                                          "w2_t", "w1_t", "wo_t", "dg1", "dbe1", "dg2", "dbe2", "db2", "db1", "dbo",
                                          "df", "dx1_res", "dz1", "dx1_ffn", "dao", "dx_res", "dout_o", "dqkv", "delta", "ln1_partials", "ln2_partials")]


class EncoderLayerBwdFullDesc(C.Structure):
    _fields_ = [("size", C.c_int), ("nhalves", C.c_int), ("half", EncoderLayerBwdDesc * 2), ("M", C.c_int)] + \
               [(n, C.c_void_p) for n in ("h", "wqkv_t", "dx", "dw2", "dw1", "dwo", "dwqkv")] + \
               [(n, C.c_int) for n in ("acc_w2", "acc_w1", "acc_wo", "acc_wqkv")] + \
               [("dbqkv", C.c_void_p), ("ln1_ws", C.c_void_p), ("ln2_ws", C.c_void_p), ("ln_rows", C.c_int)]


class WPrepDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dstT", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("kw", C.c_int),
                ("tap", C.c_int), ("ld_dst", C.c_int), ("ld_dstT", C.c_int)]


GEMM_DESC_SIZE, ENC_DESC_SIZE, ENC_BWD_DESC_SIZE = C.sizeof(GemmDesc), C.sizeof(EncoderLayerDesc), C.sizeof(EncoderLayerBwdDesc)
ENC_BWD_FULL_DESC_SIZE = C.sizeof(EncoderLayerBwdFullDesc)
ABI_VERSION = 120            # include/a8hip.h: A8HIP_VERSION


def lib():
    """Load (once) and return the HIP library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"a8hip: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C audio8_amd/csrc`).  There is no CPU/PyTorch fallback for the hot path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.a8_last_error.restype = C.c_char_p
        _lib.a8_version.restype = C.c_int
        if _lib.a8_version() != ABI_VERSION:          # descriptor layouts are positional: never talk to a library of another layout
            got, _lib = _lib.a8_version(), None
            raise RuntimeError(f"a8hip: {LIB_PATH} has ABI version {got}, this binding was written for {ABI_VERSION} -- rebuild the library")
        for name, argtypes in ARGTYPES.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        if hasattr(_lib, "a8_debug_set_stamps"):          # developer build only (make STAMPS=1): in-kernel timestamps of the GEMM tiles
            _lib.a8_debug_set_stamps.argtypes = [C.c_void_p]
            _lib.a8_debug_set_stamps.restype = C.c_int
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().a8_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"a8hip {what} failed (status {status}): {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"a8hip: unsupported activation dtype {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on this process's device (one GPU per process)."""
    global _dev_index
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_dev_index)


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("a8hip: tensors must live on the GPU (no CPU fallback in the product path)")


EXPORTS = [
    "a8_version", "a8_last_error", "a8_gemm", "a8_gemm_grouped", "a8_set_tuning", "a8_get_tuning", "a8_delay_us", "a8_encoder_layer_fwd", "a8_encoder_layer_bwd", "a8_encoder_layer_bwd_full", "a8_prepare_weights", "a8_prepare_weights_device",
    "a8_conv0_gn_gelu_fwd", "a8_conv0_gn_gelu_bwd", "a8_col2im_gelu_grad",
    "a8_layernorm_fwd", "a8_layernorm_bwd", "a8_layernorm_bwd_partial_rows", "a8_colsum_partials_reduce",
    "a8_cast", "a8_dropout", "a8_mask_fill_rows", "a8_mask_fill_rows_bwd", "a8_mask_zero_cols", "a8_zero_rows",
    "a8_gather_rows", "a8_scatter_add_rows", "a8_add",
    "a8_posconv_fwd", "a8_posconv_wgrad", "a8_weightnorm_pack", "a8_weightnorm_bwd", "a8_gelu_bwd_bias",
    "a8_attn_fwd", "a8_attn_bwd",
    "a8_vq_select", "a8_vq_gather", "a8_vq_ppl", "a8_vq_bwd",
    "a8_contrastive_fwd", "a8_contrastive_bwd", "a8_loss_combine", "a8_loss_combine_bwd", "a8_ctc_greedy_decode",
    "a8_ctc_pack_targets", "a8_ctc_fwd_bwd", "a8_log_softmax_fwd", "a8_log_softmax_bwd",
    "a8_sumsq", "a8_adamw", "a8_zero", "a8_conv_wgrad_fold", "a8_host_negative_indices",
]
