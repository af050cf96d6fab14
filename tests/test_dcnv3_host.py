"""CPU: host side of the DCNv3 operator -- struct layout, output-size rule, argument checks (which run before any launch)
and the reference's error behaviour for tensors that are not on the device."""
import ctypes as C

import pytest
import torch

from oracle import dcnv3_oracle as D


def geom(N=2, H=8, W=8, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1, dh=1, dw=1, group=4, gc=16, os=2.0, step=256, rmc=0):
    from mtp_amd._lib import Dcnv3Geom
    g = Dcnv3Geom()
    g.N, g.H, g.W = N, H, W
    g.kernel_h, g.kernel_w, g.stride_h, g.stride_w, g.pad_h, g.pad_w, g.dilation_h, g.dilation_w = kh, kw, sh, sw, ph, pw, dh, dw
    g.group, g.group_channels, g.offset_scale, g.im2col_step, g.remove_center = group, gc, os, step, rmc
    return g


def test_geometry_struct_layout():
    from mtp_amd._lib import Dcnv3Geom
    # mtp_dcnv3_geom: 3 i64, 8 i32, 2 i32, f32, 3 i32
    assert C.sizeof(Dcnv3Geom) == 24 + 14 * 4 and Dcnv3Geom.kernel_h.offset == 24 and Dcnv3Geom.offset_scale.offset == 64 and Dcnv3Geom.remove_center.offset == 72


@pytest.mark.parametrize("cfg", [dict(), dict(H=9, sh=2, sw=2), dict(W=10, ph=2, pw=2, dh=2, dw=2), dict(ph=0, pw=0), dict(kh=5, kw=5, ph=2, pw=2, H=9, W=9, rmc=1),
                                 dict(kh=1, kw=3, H=6), dict(H=128, W=128, group=12)])
def test_output_size_rule_matches_the_oracle(cfg):
    from mtp_amd import _lib
    g = geom(**cfg)
    ho, wo = C.c_int64(), C.c_int64()
    assert _lib.load().mtp_dcnv3_out_size(C.byref(g), C.byref(ho), C.byref(wo)) == 0
    assert (ho.value, wo.value) == D.out_size(g.H, g.W, g.kernel_h, g.kernel_w, g.stride_h, g.stride_w, g.pad_h, g.pad_w, g.dilation_h, g.dilation_w)


def test_argument_checks_reject_before_any_launch():
    from mtp_amd import _lib
    lib = _lib.load()
    ho, wo = C.c_int64(), C.c_int64()
    for bad in (dict(N=3, step=2), dict(kh=4, kw=4, rmc=1), dict(kh=3, kw=5, rmc=1), dict(group=0), dict(sh=0), dict(H=1, W=1, ph=0, pw=0), dict(step=0)):
        assert lib.mtp_dcnv3_out_size(C.byref(geom(**bad)), C.byref(ho), C.byref(wo)) == -1, bad
    g = geom()
    assert lib.mtp_dcnv3_fwd(None, None, None, None, 0, C.byref(g), None) == -1
    assert lib.mtp_dcnv3_fwd(1, 1, 1, 1, 7, C.byref(g), None) == -1                      # unknown dtype
    assert lib.mtp_dcnv3_bwd(None, None, None, None, 0, None, None, None, C.byref(g), None) == -1
    assert lib.mtp_dcnv3_fwd(1, 1, 1, 1, 0, C.byref(geom(H=40000, W=40000)), None) == -1  # beyond the supported map size
    assert lib.mtp_dcnv3_fwd(1, 1, 1, 1, 0, C.byref(geom(H=16384, W=16384, group=8)), None) == -2   # 32-bit in-image offsets: unsupported, not wrong


def test_no_cpu_path_like_the_reference():
    """the reference's CPU entry points throw (src/cpu/dcnv3_cpu.cpp:25,36; dcnv3.h:37,58)"""
    from mtp_amd.ops_dcnv3 import DCNv3Function, dcnv3_backward, dcnv3_forward, ext
    x, off, m = torch.zeros(2, 8, 8, 64), torch.zeros(2, 8, 8, 72), torch.zeros(2, 8, 8, 36)
    args = (3, 3, 1, 1, 1, 1, 1, 1, 4, 16, 2.0)
    with pytest.raises(RuntimeError, match="device tensor"):
        dcnv3_forward(x, off, m, *args, 256, 0)
    with pytest.raises(RuntimeError, match="device tensor"):
        dcnv3_backward(x, off, m, *args, x, 256, 0)
    with pytest.raises(RuntimeError, match="device tensor"):
        DCNv3Function.apply(x, off, m, *args, 256, 0)
    assert ext.dcnv3_forward is dcnv3_forward and ext.dcnv3_backward is dcnv3_backward and float(ext.__version__) > 1.0
