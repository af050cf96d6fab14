"""ORACLE (test infrastructure only) -- CPU restatement of the DCNv3 core operator (InternImage's deformable convolution).

Restates the algorithm of the reference's native extension, "IM2COL" =
/root/reference/Multi-Task_Pretrain/backbone/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh (forward loop :225-285, bilinear
sample :30-82, backward / col2im :84-147, 290-380) behind `dcnv3_forward` / `dcnv3_backward`
(ops_dcnv3/src/dcnv3.h:20-59), in plain vectorised torch on the CPU (fp32 or fp64): explicit sampling locations,
explicit four-corner bilinear blend, explicit scatter for the input gradient -- no grid_sample, no autograd.

PINNED: tests/golden/f11_dcnv3.npz holds outputs and gradients of the reference's own pure-torch core
`dcnv3_core_pytorch` (ops_dcnv3/functions/dcnv3_func.py:168-236 -- the function the reference's test,
ops_dcnv3/test.py, holds its CUDA kernels to) generated in the build container with that test's input recipe
(tests/golden/make_golden.py f11); tests/test_dcnv3_oracle.py holds this file to them (fp64: 2e-6 -- the float32 reference
grid inside dcnv3_core_pytorch -- and 1e-12 where that grid is exact; fp32: 1e-5).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Tensor layouts (the reference's): input (N, H, W, group*group_channels) channels-last; offset
(N, Ho, Wo, group*P*2) as (group, point, (x, y)); mask (N, Ho, Wo, group*P); P = kh*kw - remove_center; points are
ordered kernel-column-major: p = i*kh + j with i over kernel_w (x) and j over kernel_h (y)  (IM2COL:258-259).
"""
import torch


def out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    """ops_dcnv3/src/cuda/dcnv3_cuda.cu:40-45"""
    return (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1


def _points(kh, kw, remove_center):
    """kernel points in the reference's order; the centre is skipped when remove_center (IM2COL:254-260)"""
    return [(i, j) for i in range(kw) for j in range(kh) if not (remove_center and i == kw // 2 and j == kh // 2)]


def _locations(offset, H, W, kh, kw, sh, sw, ph, pw, dh, dw, group, offset_scale, remove_center):
    """sampling locations in UNPADDED input pixel coordinates, (N, Ho, Wo, group, P) each  (IM2COL:232-236, 249-266)"""
    N, Ho, Wo, _ = offset.shape
    pts = _points(kh, kw, remove_center)
    P = len(pts)
    off = offset.reshape(N, Ho, Wo, group, P, 2)
    dt = offset.dtype
    pi = torch.tensor([p[0] for p in pts], dtype=dt)
    pj = torch.tensor([p[1] for p in pts], dtype=dt)
    p0w = ((dw * (kw - 1)) >> 1) - pw + torch.arange(Wo, dtype=dt) * sw
    p0h = ((dh * (kh - 1)) >> 1) - ph + torch.arange(Ho, dtype=dt) * sh
    p0w_ = p0w - ((dw * (kw - 1)) >> 1) * offset_scale
    p0h_ = p0h - ((dh * (kh - 1)) >> 1) * offset_scale
    loc_w = p0w_.view(1, 1, Wo, 1, 1) + (pi.view(1, 1, 1, 1, P) * dw + off[..., 0]) * offset_scale
    loc_h = p0h_.view(1, Ho, 1, 1, 1) + (pj.view(1, 1, 1, 1, P) * dh + off[..., 1]) * offset_scale
    return loc_h, loc_w


def _corners(loc_h, loc_w, H, W):
    """validity of the point (IM2COL:268-269), the four corner indices / weights / in-range flags (IM2COL:38-75)"""
    valid = (loc_h > -1) & (loc_w > -1) & (loc_h < H) & (loc_w < W)
    h_low, w_low = torch.floor(loc_h), torch.floor(loc_w)
    lh, lw = loc_h - h_low, loc_w - w_low
    hh, hw = 1 - lh, 1 - lw
    h_low, w_low = h_low.long(), w_low.long()
    h_high, w_high = h_low + 1, w_low + 1
    corners = [(h_low, w_low, hh * hw, (h_low >= 0) & (w_low >= 0)),
               (h_low, w_high, hh * lw, (h_low >= 0) & (w_high <= W - 1)),
               (h_high, w_low, lh * hw, (h_high <= H - 1) & (w_low >= 0)),
               (h_high, w_high, lh * lw, (h_high <= H - 1) & (w_high <= W - 1))]
    return valid, corners, (lh, lw, hh, hw)


def _gather(inp5, n_idx, g_idx, hc, wc, ok, H, W):
    """inp5 (N, H, W, group, gc) -> (..., gc) values at the corner, zeros where the corner is out of range"""
    hcc, wcc = hc.clamp(0, H - 1), wc.clamp(0, W - 1)
    v = inp5[n_idx, hcc, wcc, g_idx]
    return v * ok.unsqueeze(-1).to(v.dtype)


def dcnv3_forward(input, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale, remove_center=0):
    """out[n, ho, wo, g, c] = sum_p mask[n, ho, wo, g, p] * bilinear(input[n, :, :, g, c], loc_h, loc_w)   (IM2COL:225-285)"""
    N, H, W, C = input.shape
    assert C == group * group_channels
    Ho, Wo = out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert offset.shape[1:3] == (Ho, Wo) and mask.shape[1:3] == (Ho, Wo)
    loc_h, loc_w = _locations(offset, H, W, kh, kw, sh, sw, ph, pw, dh, dw, group, offset_scale, remove_center)
    P = loc_h.shape[-1]
    valid, corners, _ = _corners(loc_h, loc_w, H, W)
    inp5 = input.reshape(N, H, W, group, group_channels)
    n_idx = torch.arange(N).view(N, 1, 1, 1, 1).expand_as(loc_h)
    g_idx = torch.arange(group).view(1, 1, 1, group, 1).expand_as(loc_h)
    val = 0
    for hc, wc, wgt, ok in corners:
        val = val + wgt.unsqueeze(-1) * _gather(inp5, n_idx, g_idx, hc, wc, ok & valid, H, W)
    m = mask.reshape(N, Ho, Wo, group, P)
    out = (val * m.unsqueeze(-1)).sum(4)
    return out.reshape(N, Ho, Wo, C)


def dcnv3_backward(input, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale, grad_output, remove_center=0):
    """(grad_input, grad_offset, grad_mask)  (IM2COL:84-147: per point, with top = grad_output[n,ho,wo,g,c],
    grad_mask = sum_c top*val;  grad_input[corner] += w_corner * top * mask;
    grad_offset = offset_scale * sum_c d(val)/d(loc) * top * mask, x first)"""
    N, H, W, C = input.shape
    Ho, Wo = out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    loc_h, loc_w = _locations(offset, H, W, kh, kw, sh, sw, ph, pw, dh, dw, group, offset_scale, remove_center)
    P = loc_h.shape[-1]
    valid, corners, (lh, lw, hh, hw) = _corners(loc_h, loc_w, H, W)
    inp5 = input.reshape(N, H, W, group, group_channels)
    n_idx = torch.arange(N).view(N, 1, 1, 1, 1).expand_as(loc_h)
    g_idx = torch.arange(group).view(1, 1, 1, group, 1).expand_as(loc_h)
    top = grad_output.reshape(N, Ho, Wo, group, 1, group_channels)
    m = mask.reshape(N, Ho, Wo, group, P)
    tg = top * m.unsqueeze(-1)                                   # top_grad_im (IM2COL:107)
    v = [_gather(inp5, n_idx, g_idx, hc, wc, ok & valid, H, W) for hc, wc, _, ok in corners]
    val = sum(c[2].unsqueeze(-1) * vi for c, vi in zip(corners, v))
    grad_mask = (top * val).sum(-1)
    u = lambda t: t.unsqueeze(-1)
    gw = -u(hh) * v[0] + u(hh) * v[1] - u(lh) * v[2] + u(lh) * v[3]          # d val / d loc_w (IM2COL:113-139)
    gh = -u(hw) * v[0] - u(lw) * v[1] + u(hw) * v[2] + u(lw) * v[3]          # d val / d loc_h
    grad_offset = torch.stack([offset_scale * (gw * tg).sum(-1), offset_scale * (gh * tg).sum(-1)], -1)
    grad_input = torch.zeros(N * H * W * group, group_channels, dtype=input.dtype)
    for hc, wc, wgt, ok in corners:
        okk = (ok & valid)
        flat = ((n_idx * H + hc.clamp(0, H - 1)) * W + wc.clamp(0, W - 1)) * group + g_idx
        contrib = u(wgt * okk.to(wgt.dtype)) * tg
        grad_input.index_add_(0, flat.reshape(-1), contrib.reshape(-1, group_channels))
    return grad_input.reshape(N, H, W, C), grad_offset.reshape(offset.shape), grad_mask.reshape(mask.shape)
