"""DCNv3 core operator on MI355X behind the reference's own operator interface.

Mirrors Multi-Task_Pretrain/backbone/ops_dcnv3:
  * `dcnv3_forward` / `dcnv3_backward` -- the two functions of the reference's compiled `DCNv3` extension module
    (src/vision.cpp:14-17, src/dcnv3.h:20-59), same positional arguments, same return values, same errors
    (RuntimeError for non-contiguous / non-device tensors, channel mismatch and the im2col_step rule,
    src/cuda/dcnv3_cuda.cu:28-53);
  * `DCNv3Function` -- functions/dcnv3_func.py:22-77, same `apply(...)` signature.
so the reference's dcnv3_func.py runs unchanged with `import DCNv3` pointing at `mtp_amd.ops_dcnv3.ext`
(INTEGRATION.md).  All compute is libmtp_hip.so (mtp_dcnv3_fwd / mtp_dcnv3_bwd, mtp_amd/csrc/dcnv3.hip); there is no CPU
or torch fallback (the reference's CPU entry points throw too: src/cpu/dcnv3_cpu.cpp:25,36).

Dtypes (the reference dispatches float / double / half, dcnv3_cuda.cu:69): float32 and bfloat16 run on their own kernel instantiations; float16
(round 6) runs on the float32 kernels -- the casts at the boundary are torch's, the arithmetic is at least as exact as a native half kernel's, the
output comes back as float16; float64 (round 6) runs on plain double kernels with f64 atomics (MTP_F64: a validation path -- the reference's own
test-suite checks its gradients numerically in double -- not a fast one), gradients in float64.  Gradients of a bfloat16 / float16 call are
float32, as the reference promotes half to float (dcnv3_cuda.cu:125-128).
"""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from ..ops import _dt, _s, lib


def _geom(input, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step, remove_center):
    g = _lib.Dcnv3Geom()
    g.N, g.H, g.W = int(input.shape[0]), int(input.shape[1]), int(input.shape[2])
    g.kernel_h, g.kernel_w, g.stride_h, g.stride_w = int(kernel_h), int(kernel_w), int(stride_h), int(stride_w)
    g.pad_h, g.pad_w, g.dilation_h, g.dilation_w = int(pad_h), int(pad_w), int(dilation_h), int(dilation_w)
    g.group, g.group_channels, g.offset_scale = int(group), int(group_channels), float(offset_scale)
    g.im2col_step, g.remove_center = int(im2col_step), int(remove_center)
    g.variant = int(os.environ.get("MTP_DCNV3_VARIANT", "0"))     # A/B switch, see mtp_hip.h
    return g


def _dtc(t):
    """dtype code of the C ABI; float64 exists for the DCNv3 entry points only (MTP_F64)"""
    return _lib.MTP_F64 if t.dtype == torch.float64 else _dt(t)


def _as_f32(*ts):
    """float16 operands -> contiguous float32 copies (the half path of the reference, on the float32 kernels)"""
    return [t.float() if t.dtype == torch.float16 else t for t in ts]


def _check_inputs(names_tensors, input, group, group_channels, im2col_step):
    for name, t in names_tensors:                                     # dcnv3_cuda.cu:28-33, 96-105
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
        if not t.is_cuda:
            raise RuntimeError("%s must be a device tensor (mtp_amd has no CPU path; the reference's CPU build throws as well)" % name)
        if t.dtype != input.dtype:
            raise RuntimeError("%s must have the dtype of input (%s), got %s" % (name, input.dtype, t.dtype))
    if input.dim() != 4:
        raise RuntimeError("input must be (N, H, W, C) channels-last")
    batch, channels = input.shape[0], input.shape[3]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:                                # dcnv3_cuda.cu:46-49
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    if channels != group * group_channels:                            # dcnv3_cuda.cu:50-53
        raise RuntimeError("Input channels and group times group channels wont match: (%d vs %d)." % (channels, group * group_channels))


def out_size(g):
    ho, wo = C.c_int64(), C.c_int64()
    _lib.check(lib().mtp_dcnv3_out_size(C.byref(g), C.byref(ho), C.byref(wo)), "mtp_dcnv3_out_size")
    return ho.value, wo.value


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                  offset_scale, im2col_step, remove_center=0):
    """-> output (N, Ho, Wo, group*group_channels), dtype of input  (dcnv3.h:20-38)"""
    _check_inputs([("input", input), ("offset", offset), ("mask", mask)], input, group, group_channels, im2col_step)
    g = _geom(input, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step, remove_center)
    Ho, Wo = out_size(g)
    P = kernel_h * kernel_w - int(remove_center)
    N = input.shape[0]
    if tuple(offset.shape) != (N, Ho, Wo, group * P * 2) or tuple(mask.shape) != (N, Ho, Wo, group * P):
        raise RuntimeError("offset / mask must be (N, %d, %d, group*P*2) / (N, %d, %d, group*P) with P = %d" % (Ho, Wo, Ho, Wo, P))
    half = input.dtype == torch.float16
    input, offset, mask = _as_f32(input, offset, mask)
    output = torch.empty((N, Ho, Wo, group * group_channels), dtype=input.dtype, device=input.device)
    _lib.check(lib().mtp_dcnv3_fwd(input.data_ptr(), offset.data_ptr(), mask.data_ptr(), output.data_ptr(), _dtc(input), C.byref(g), _s()), "mtp_dcnv3_fwd")
    return output.half() if half else output


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                   offset_scale, grad_output, im2col_step, remove_center=0):
    """-> [grad_input, grad_offset, grad_mask]  (dcnv3.h:40-59); float32 for float32 and bfloat16 inputs"""
    _check_inputs([("input", input), ("offset", offset), ("mask", mask), ("grad_output", grad_output)], input, group, group_channels, im2col_step)
    g = _geom(input, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step, remove_center)
    Ho, Wo = out_size(g)
    if tuple(grad_output.shape) != (input.shape[0], Ho, Wo, group * group_channels):
        raise RuntimeError("grad_output must be (N, %d, %d, %d)" % (Ho, Wo, group * group_channels))
    kw = dict(dtype=torch.float64 if input.dtype == torch.float64 else torch.float32, device=input.device)
    input, offset, mask, grad_output = _as_f32(input, offset, mask, grad_output)
    grad_input, grad_offset, grad_mask = torch.empty(input.shape, **kw), torch.empty(offset.shape, **kw), torch.empty(mask.shape, **kw)
    _lib.check(lib().mtp_dcnv3_bwd(input.data_ptr(), offset.data_ptr(), mask.data_ptr(), grad_output.data_ptr(), _dtc(input), grad_input.data_ptr(),
                                   grad_offset.data_ptr(), grad_mask.data_ptr(), C.byref(g), _s()), "mtp_dcnv3_bwd")
    return [grad_input, grad_offset, grad_mask]


def dcnv3_backward_act(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                       offset_scale, grad_output, im2col_step, act_ld, remove_center=0):
    """dcnv3_backward plus grad_offset in the input dtype as (N * Ho * Wo, act_ld) rows (pad columns zero) -- the GEMM operand of the offset head's
    backward, written by the kernel that computes it.  -> [grad_input, grad_offset, grad_mask, grad_offset_act]; grad_offset_act is None when the
    geometry has no gather-form backward (the caller then casts grad_offset itself)."""
    _check_inputs([("input", input), ("offset", offset), ("mask", mask), ("grad_output", grad_output)], input, group, group_channels, im2col_step)
    g = _geom(input, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step, remove_center)
    Ho, Wo = out_size(g)
    if tuple(grad_output.shape) != (input.shape[0], Ho, Wo, group * group_channels):
        raise RuntimeError("grad_output must be (N, %d, %d, %d)" % (Ho, Wo, group * group_channels))
    kw = dict(dtype=torch.float32, device=input.device)
    grad_input, grad_offset, grad_mask = torch.empty(input.shape, **kw), torch.empty(offset.shape, **kw), torch.empty(mask.shape, **kw)
    act = torch.empty(input.shape[0] * Ho * Wo, act_ld, dtype=input.dtype, device=input.device)
    rc = lib().mtp_dcnv3_bwd_act(input.data_ptr(), offset.data_ptr(), mask.data_ptr(), grad_output.data_ptr(), _dt(input), grad_input.data_ptr(),
                                 grad_offset.data_ptr(), grad_mask.data_ptr(), act.data_ptr(), act_ld, C.byref(g), _s())
    if rc == -2:      # MTP_ERR_UNSUPPORTED: no gather-form backward for this geometry, nothing was launched
        return dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                              offset_scale, grad_output, im2col_step, remove_center) + [None]
    _lib.check(rc, "mtp_dcnv3_bwd_act")
    return [grad_input, grad_offset, grad_mask, act]


class DCNv3Function(Function):
    """functions/dcnv3_func.py:22-77 -- same apply() arguments; the saved tensors and the 13 `None` gradients too."""

    @staticmethod
    def forward(ctx, input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                offset_scale, im2col_step, remove_center):
        ctx.cfg = (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale)
        ctx.im2col_step, ctx.remove_center = im2col_step, remove_center
        output = dcnv3_forward(input, offset, mask, *ctx.cfg, im2col_step, remove_center)
        ctx.save_for_backward(input, offset, mask)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask = ctx.saved_tensors
        gi, go, gm = dcnv3_backward(input, offset, mask, *ctx.cfg, grad_output.contiguous(), ctx.im2col_step, ctx.remove_center)
        # autograd wants the gradient in the dtype of the input it belongs to
        return (gi.to(input.dtype), go.to(offset.dtype), gm.to(mask.dtype)) + (None,) * 13
