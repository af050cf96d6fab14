"""CPU: the C-ABI library loads and exports every symbol include/mtp_hip.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mtp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported():
    from mtp_amd import _lib
    names = _declared()
    assert len(names) >= 25
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and include/mtp_hip.h disagree"
    lib = _lib.load()                     # raises if libmtp_hip.so is missing: there is no fallback
    for n in names:
        assert getattr(lib, n) is not None
    assert b"gfx950" in lib.mtp_version()


def test_gemm_args_struct_layout():
    import ctypes as C
    from mtp_amd._lib import GemmArgs
    # mirrors `mtp_gemm_args` in the header: 3 ptrs, 6 i64, 3 i32 (+pad), ptr, i64, ptr, 2 i64, ptr, i64, ptr, i64, 2 i32, ptr, 2 i32, ptr, i64
    assert C.sizeof(GemmArgs) == 3 * 8 + 6 * 8 + 3 * 4 + 4 + 8 + 8 + 8 + 16 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8
    assert GemmArgs.workspace.offset == 184 and GemmArgs.workspace_bytes.offset == 192
    assert GemmArgs.bias.offset == 88 and GemmArgs.split_k.offset == 160 and GemmArgs.colsum.offset == 168 and GemmArgs.defer_sum.offset == 176


def test_weight_image_descriptor_layout():
    import ctypes as C
    from mtp_amd._lib import WimgDesc
    # mtp_wimg_desc: 3 ptrs, 3 i64, 2 i32
    assert C.sizeof(WimgDesc) == 56 and WimgDesc.R.offset == 24 and WimgDesc.tile0.offset == 40 and WimgDesc.f32_out.offset == 48


def test_arg_checks_reject_without_gpu():
    """argument validation happens before any launch, so it is testable on CPU"""
    import ctypes as C
    from mtp_amd import _lib
    lib = _lib.load()
    g = _lib.GemmArgs()
    assert lib.mtp_gemm_nt(C.byref(g), None) == -1
    assert lib.mtp_gemm_tn(C.byref(g), None) == -1
    assert lib.mtp_layernorm_fwd(None, 0, None, None, None, 0, None, None, 4, 8, 1e-6, 0, None) == -1
    assert lib.mtp_full_attn_fwd(None, None, None, 0, None, None, 1, 14, 14, 2, 64, 0.125, None) == -1
    assert lib.mtp_layernorm_bwd_partial_rows(12544) == 512 and lib.mtp_layernorm_bwd_partial_rows(10) == 3


def test_every_entry_point_rejects_null_arguments_before_launching():
    """the whole ABI: all-NULL pointers / zero sizes must come back as MTP_ERR_ARG (-1) from the argument checks -- no launch,
    no dereference, no GPU needed.  (Pure query functions are exercised in the other tests.)"""
    import ctypes as C
    from mtp_amd import _lib
    lib = _lib.load()
    queries = {"mtp_version", "mtp_layernorm_bwd_partial_rows", "mtp_full_attn_bwd_workspace_floats", "mtp_dwconv3x3_bwd_dw_partial_rows",
               "mtp_scale_residual_bwd_partial_rows", "mtp_gemm_nt_workspace_bytes"}
    for name, (_, argtypes) in sorted(_lib.SIGNATURES.items()):
        if name in queries:
            continue
        args = [0.0 if a is C.c_float else (None if (a is C.c_void_p or hasattr(a, "contents")) else 0) for a in argtypes]
        assert getattr(lib, name)(*args) == -1, name
