"""ctypes binding of libmtp_hip.so (the C ABI declared in include/mtp_hip.h).

The product path has NO fallback: if the library is missing or a symbol does not resolve, importing the
ops raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C mtp_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTP_HIP_LIB") or os.path.join(_HERE, "libmtp_hip.so")   # MTP_HIP_LIB: A/B builds of the same ABI

MTP_F32, MTP_BF16, MTP_F64 = 0, 1, 2      # (MTP_F64: the DCNv3 entry points only)
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_BIAS_GELU_DG, EPI_MUL = 0, 1, 2, 3, 4, 5

p, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", p), ("B", p), ("C", p), ("M", i64), ("N", i64), ("K", i64),
                ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("in_dtype", i32), ("out_dtype", i32), ("epilogue", i32),
                ("bias", p), ("bias_mod", i64), ("res", p), ("res_ld", i64), ("res_mod", i64),
                ("rowscale", p), ("rows_per_sample", i64), ("aux", p), ("aux_ld", i64),
                ("split_k", i32), ("variant", i32), ("colsum", p), ("defer_sum", i32), ("pad_", i32),
                ("workspace", p), ("workspace_bytes", i64)]


class WimgDesc(C.Structure):
    """mtp_wimg_desc"""
    _fields_ = [("src", p), ("w", p), ("wt", p), ("R", i64), ("C", i64), ("tile0", i64), ("f32_out", C.c_int32), ("wd", C.c_float)]


class Dcnv3Geom(C.Structure):
    """mtp_dcnv3_geom"""
    _fields_ = [("N", i64), ("H", i64), ("W", i64), ("kernel_h", C.c_int32), ("kernel_w", C.c_int32), ("stride_h", C.c_int32), ("stride_w", C.c_int32),
                ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("dilation_h", C.c_int32), ("dilation_w", C.c_int32), ("group", C.c_int32),
                ("group_channels", C.c_int32), ("offset_scale", C.c_float), ("im2col_step", C.c_int32), ("remove_center", C.c_int32), ("variant", C.c_int32)]


# name -> (restype, argtypes); must list EVERY function declared in include/mtp_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "mtp_weight_images": (i32, [p, i32, i64, i32, p]),
    "mtp_dwconv_fwd": (i32, [p, p, p, p, i32, i64, i64, i64, i64, i32, p]),
    "mtp_dwconv_bwd_dx": (i32, [p, i32, p, p, i32, i64, i64, i64, i64, i32, p]),
    "mtp_dwconv_bwd_dw": (i32, [p, p, i32, p, p, i64, i64, i64, i64, i32, p]),
    "mtp_center_feature_scale_fwd": (i32, [p, p, p, i64, p, i32, i64, i64, i64, p]),
    "mtp_center_feature_scale_bwd": (i32, [p, p, p, p, i64, p, p, p, i32, i64, i64, i64, p]),
    "mtp_dcnv3_out_size": (i32, [C.POINTER(Dcnv3Geom), C.POINTER(i64), C.POINTER(i64)]),
    "mtp_dcnv3_fwd": (i32, [p, p, p, p, i32, C.POINTER(Dcnv3Geom), p]),
    "mtp_dcnv3_bwd": (i32, [p, p, p, p, i32, p, p, p, C.POINTER(Dcnv3Geom), p]),
    "mtp_dcnv3_bwd_act": (i32, [p, p, p, p, i32, p, p, p, p, i64, C.POINTER(Dcnv3Geom), p]),
    "mtp_gemm_nt": (i32, [C.POINTER(GemmArgs), p]),
    "mtp_gemm_nt_tile": (i32, [C.POINTER(GemmArgs)]),
    "mtp_gemm_nt_workspace_bytes": (i64, []),
    "mtp_gemm_tn": (i32, [C.POINTER(GemmArgs), p]),
    "mtp_gemm_tn_grouped": (i32, [C.POINTER(GemmArgs), i32, p]),
    "mtp_sum_partials_batch": (i32, [p, p, p, p, i32, p]),
    "mtp_layernorm_fwd": (i32, [p, i32, p, p, p, i32, p, p, i64, i64, f32, i32, p]),
    "mtp_layernorm_bwd_partial_rows": (i64, [i64]),
    "mtp_layernorm_bwd": (i32, [p, i32, p, i32, p, p, p, p, i32, p, p, p, i32, p, i32, p, i64, p, p, i64, i64, i64, p]),
    "mtp_reduce_rows_f32": (i32, [p, i64, p, i64, i64, i32, p]),
    "mtp_reduce_rows_batched_f32": (i32, [p, p, i32, i64, i64, i64, i32, p]),
    "mtp_reduce_rows_t_f32": (i32, [p, i64, p, i64, i64, i64, i32, p]),
    "mtp_reduce_rows_t_batched_f32": (i32, [p, p, i32, i64, i64, i64, i64, i32, p]),
    "mtp_copy_segments_f32": (i32, [p, p, p, i32, p]),
    "mtp_colsum": (i32, [p, i32, i64, p, i64, i64, p]),
    "mtp_colsum_acc": (i32, [p, i32, i64, p, i64, i64, p]),
    "mtp_patchify": (i32, [p, p, i32, i64, i64, i64, i64, i64, p]),
    "mtp_preprocess_patchify": (i32, [p, p, i32, i64, i64, i64, i64, i64, C.POINTER(f32), C.POINTER(f32), i32, f32, p]),
    "mtp_unpatchify": (i32, [p, i32, p, i64, i64, i64, i64, i64, p]),
    "mtp_cast": (i32, [p, i32, p, i32, i64, p]),
    "mtp_transpose_cast": (i32, [p, p, i32, i64, i64, p]),
    "mtp_convt_pack": (i32, [p, p, p, i32, i64, i64, p]),
    "mtp_convt_unpack_grad": (i32, [p, p, i64, i64, p]),
    "mtp_tokens_to_nchw": (i32, [p, i32, p, i32, i64, i64, i64, i64, i32, p]),
    "mtp_nchw_to_tokens": (i32, [p, i32, p, i32, i64, i64, i64, i64, i32, p]),
    "mtp_maxpool2_tokens_fwd": (i32, [p, p, i32, i64, i64, i64, i64, p]),
    "mtp_maxpool2_tokens_bwd": (i32, [p, p, i32, p, i32, i64, i64, i64, i64, p]),
    "mtp_axpy_f32": (i32, [p, p, f32, i64, p]),
    "mtp_scale_rows_cast": (i32, [p, p, i32, p, i64, i64, i64, p]),
    "mtp_im2col3x3": (i32, [p, i32, i64, i64, i64, i64, p, i32, i64, i64, i64, i64, i64, i64, p]),
    "mtp_col2im3x3": (i32, [p, i32, p, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i32, p]),
    "mtp_conv3x3_pack": (i32, [p, p, p, i32, i64, i64, i64, p]),
    "mtp_conv3x3_unpack_grad": (i32, [p, p, i64, i64, i64, p]),
    "mtp_pack_rows_padded": (i32, [p, p, p, i32, i64, i64, i64, p]),
    "mtp_cast_pad_rows": (i32, [p, i64, p, i32, i64, i64, p]),
    "mtp_copy_rows": (i32, [p, i64, p, i64, i32, i64, i64, p]),
    "mtp_dwconv3x3_fwd": (i32, [p, p, p, p, i32, i64, i64, i64, i64, p]),
    "mtp_dwconv3x3_bwd_dx": (i32, [p, i32, p, p, i32, i64, i64, i64, i64, p]),
    "mtp_dwconv3x3_bwd_dw_partial_rows": (i64, [i64, i64, i64]),
    "mtp_dwconv3x3_bwd_dw": (i32, [p, p, i32, p, i64, i64, i64, i64, p]),
    "mtp_softmax_groups_fwd": (i32, [p, i64, p, i32, i64, i64, i64, p]),
    "mtp_softmax_groups_bwd": (i32, [p, p, p, i64, i32, i64, i64, i64, p]),
    "mtp_scale_residual_fwd": (i32, [p, p, i32, p, p, i64, p, p, i64, i64, p]),
    "mtp_scale_residual_bwd_partial_rows": (i64, [i64]),
    "mtp_scale_residual_bwd": (i32, [p, p, i32, p, p, i64, p, p, i64, i64, p]),
    "mtp_comm_unique_id": (i32, [p]),
    "mtp_comm_init": (i32, [p, i32, i32, C.POINTER(C.c_void_p)]),
    "mtp_comm_allreduce_bucket": (i32, [p, p, i64, p]),
    "mtp_comm_allreduce_bucket_dt": (i32, [p, p, i64, i32, p]),
    "mtp_comm_reduce_scatter_bucket": (i32, [p, p, i64, i32, i32, p]),
    "mtp_comm_allgather_bucket": (i32, [p, p, i64, i32, i32, p]),
    "mtp_comm_info": (i32, [p, p]),
    "mtp_comm_destroy": (i32, [p]),
    "mtp_full_attn_fwd": (i32, [p, p, p, i32, p, p, i64, i64, i64, i64, i64, f32, p]),
    "mtp_full_attn_bwd_workspace_floats": (i64, [i64, i64, i64, i64]),
    "mtp_full_attn_bwd": (i32, [p, p, p, p, p, i32, p, p, p, p, i64, i64, i64, i64, i64, f32, p]),
    "mtp_rvsa_pool_fwd": (i32, [p, i32, p, p, i64, i64, i64, i64, p]),
    "mtp_rvsa_pool_bwd": (i32, [p, p, p, i32, i32, i64, i64, i64, i64, p]),
    "mtp_rvsa_sampling_fwd": (i32, [p, i32, p, p, p, p, p, i64, i64, i64, i64, i64, p]),
    "mtp_rvsa_sampling_bwd": (i32, [p, p, p, p, i32, i64, i64, i64, i64, i64, p]),
    "mtp_rvsa_sampling_bwd_win": (i32, [p, p, p, p, i64, i64, i64, p]),
    "mtp_layernorm_residual_fwd": (i32, [p, i32, p, p, p, p, p, i64, p, p, p, p, i64, i64, f32, p]),
    "mtp_layernorm_residual_bwd": (i32, [p, p, i32, p, p, p, p, p, p, i64, p, p, i64, i64, p]),
    "mtp_layernorm_bwd_win": (i32, [p, i32, p, i32, p, p, p, p, p, p, i32, p, i32, p, i64, p, p, i64, i64, i64, p, i64, i64, i64, p]),
    "mtp_small_linear_fwd": (i32, [p, p, p, p, i64, i64, i64, p]),
    "mtp_small_linear_bwd": (i32, [p, p, p, p, p, p, i64, i64, i64, p]),
    "mtp_small_linear_dw_segments": (i32, [p, p, i64, i64, i64, i32, p, p, p, p]),
    "mtp_small_linear_dw_segments_batched": (i32, [p, p, i32, i64, i64, i64, i32, p, p, p, p]),
    "mtp_rvsa_attn_fwd": (i32, [p, p, p, p, i32, p, p, p, i64, i64, i64, i64, i64, f32, p]),
    "mtp_rvsa_attn_bwd": (i32, [p, p, p, p, p, p, p, p, p, p, i32, p, p, p, i64, i64, i64, i64, i64, f32, p]),
    "mtp_zero_segments_f32": (i32, [p, p, p, i32, p]),
    "mtp_sqnorm_f32": (i32, [p, p, i64, p]),
    "mtp_sqnorm_segments_f32": (i32, [p, p, p, i32, p, p]),
    "mtp_adamw_flat": (i32, [p, p, p, p, i64, p, p, i32, p, p, f32, f32, p]),
    "mtp_adamw_weight_images": (i32, [p, i32, i64, i32, p, p, p, p, p, p, f32, f32, p]),
    "mtp_version": (C.c_char_p, []),
    "mtp_stream_create_low_priority": (i32, [p]),
    "mtp_stream_create_cu_mask": (i32, [p, i32, p]),
    "mtp_probe_placement": (i32, [p, i32, i64, p]),
    "mtp_stream_destroy": (i32, [p]),
}

_lib = None


class MtpHipError(RuntimeError):
    pass


def load():
    """Load libmtp_hip.so and bind every symbol; raises if the HIP extension is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtpHipError("libmtp_hip.so not found at %s -- build it first (__graft_entry__.build() or make -C mtp_amd/csrc); "
                          "mtp_amd has no CPU/PyTorch fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    ab = bool(os.environ.get("MTP_HIP_LIB"))
    for name, (res, args) in SIGNATURES.items():
        if ab and not hasattr(lib, name):
            continue              # an A/B build of an older tree (tools/_abl): entry points added since are simply absent
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, "hipError_t %d" % rc)
        raise MtpHipError("%s failed: %s" % (what, kind))
