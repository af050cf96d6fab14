"""ORACLE (test infrastructure only) -- CPU restatement of the InternImage backbone around the DCNv3 core.

Groundwork for the next widening step: the HIP operator (mtp_amd/ops_dcnv3) covers the reference's native extension; the
layers around it are plain torch modules in the reference ("II" = /root/reference/Multi-Task_Pretrain/backbone/intern_image.py,
"DCNM" = .../backbone/ops_dcnv3/modules/dcnv3.py).  This file restates them channels-last, from a flat parameter dict with the
reference's state-dict keys, for the configuration family BASELINE config 5 uses (II:700-712: norm_layer='LN',
act_layer='GELU', layer_scale set, post_norm=True, no center_feature_scale / res_post_norm / level2_post_norm).
The DCNv3 core inside is oracle/dcnv3_oracle.py (explicit bilinear gather), so gradients flow through torch autograd.

PINNED: tests/golden/f12_internimage.npz = outputs and gradients of the reference's own `InternImage(core_op='DCNv3_pytorch')`
(tests/golden/make_golden.py f12); tests/test_internimage_oracle.py holds this file to them (fp32, 2e-5) and checks
`state_shapes()` against the reference's state_dict() keys / shapes / order stored in the fixture.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import torch
import torch.nn.functional as F

from . import dcnv3_oracle as D

EPS = 1e-6   # build_norm_layer default (II:39-43, DCNM:37-41)


def state_shapes(channels=192, depths=(5, 5, 24, 5), groups=(12, 24, 48, 96), mlp_ratio=4.0, kernel_size=3):
    """reference state-dict keys and shapes in the reference's order, for layer_scale + post_norm configurations"""
    s = {}
    c2 = channels // 2
    s["patch_embed.conv1.weight"] = (c2, 3, 3, 3)
    s["patch_embed.conv1.bias"] = (c2,)
    s["patch_embed.norm1.1.weight"] = (c2,)
    s["patch_embed.norm1.1.bias"] = (c2,)
    s["patch_embed.conv2.weight"] = (channels, c2, 3, 3)
    s["patch_embed.conv2.bias"] = (channels,)
    s["patch_embed.norm2.1.weight"] = (channels,)
    s["patch_embed.norm2.1.bias"] = (channels,)
    P = kernel_size * kernel_size
    for i, (depth, G) in enumerate(zip(depths, groups)):
        C = channels * 2 ** i
        hid = int(C * mlp_ratio)
        for j in range(depth):
            p = "levels.%d.blocks.%d." % (i, j)
            s[p + "gamma1"] = (C,)
            s[p + "gamma2"] = (C,)
            s[p + "norm1.0.weight"] = (C,)
            s[p + "norm1.0.bias"] = (C,)
            s[p + "dcn.dw_conv.0.weight"] = (C, 1, kernel_size, kernel_size)
            s[p + "dcn.dw_conv.0.bias"] = (C,)
            s[p + "dcn.dw_conv.1.1.weight"] = (C,)
            s[p + "dcn.dw_conv.1.1.bias"] = (C,)
            s[p + "dcn.offset.weight"] = (G * P * 2, C)
            s[p + "dcn.offset.bias"] = (G * P * 2,)
            s[p + "dcn.mask.weight"] = (G * P, C)
            s[p + "dcn.mask.bias"] = (G * P,)
            s[p + "dcn.input_proj.weight"] = (C, C)
            s[p + "dcn.input_proj.bias"] = (C,)
            s[p + "dcn.output_proj.weight"] = (C, C)
            s[p + "dcn.output_proj.bias"] = (C,)
            s[p + "norm2.0.weight"] = (C,)
            s[p + "norm2.0.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (hid, C)
            s[p + "mlp.fc1.bias"] = (hid,)
            s[p + "mlp.fc2.weight"] = (C, hid)
            s[p + "mlp.fc2.bias"] = (C,)
        if i < len(depths) - 1:
            p = "levels.%d.downsample." % i
            s[p + "conv.weight"] = (2 * C, C, 3, 3)
            s[p + "norm.1.weight"] = (2 * C,)
            s[p + "norm.1.bias"] = (2 * C,)
    return s


def _ln(x, p, key):
    return F.layer_norm(x, (x.shape[-1],), p[key + ".weight"], p[key + ".bias"], EPS)


def _conv_nhwc(x, w, b, stride, groups=1):
    """3x3 convolution, padding 1, on a channels-last tensor (the reference permutes to NCHW and back around nn.Conv2d)"""
    return F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=stride, padding=w.shape[-1] // 2, groups=groups).permute(0, 2, 3, 1)


def stem(img, p):
    """StemLayer II:239-276: conv3x3 s2 -> LN -> GELU -> conv3x3 s2 -> LN; NCHW image in, channels-last map out (stride 4)"""
    x = _conv_nhwc(img.permute(0, 2, 3, 1), p["patch_embed.conv1.weight"], p["patch_embed.conv1.bias"], 2)
    x = F.gelu(_ln(x, p, "patch_embed.norm1.1"))
    x = _conv_nhwc(x, p["patch_embed.conv2.weight"], p["patch_embed.conv2.bias"], 2)
    return _ln(x, p, "patch_embed.norm2.1")


PROBE = None   # tests set this to a dict: every DCNv3 call then records how close its sampling positions come to a cell edge


def dcnv3_module(x, p, pre, group, offset_scale, kernel_size=3):
    """DCNv3 module DCNM:187-218 (DCNv3_pytorch) == DCNM:316-353 (DCNv3): input_proj; depthwise conv -> LN -> GELU on the
    UNprojected input; offset / mask heads on that; softmax over the P points of each group; core; output_proj"""
    N, H, W, C = x.shape
    P = kernel_size * kernel_size
    xp = F.linear(x, p[pre + "input_proj.weight"], p[pre + "input_proj.bias"])
    x1 = _conv_nhwc(x, p[pre + "dw_conv.0.weight"], p[pre + "dw_conv.0.bias"], 1, groups=C)
    x1 = F.gelu(_ln(x1, p, pre + "dw_conv.1.1"))
    offset = F.linear(x1, p[pre + "offset.weight"], p[pre + "offset.bias"])
    mask = F.linear(x1, p[pre + "mask.weight"], p[pre + "mask.bias"]).reshape(N, H, W, group, P)
    mask = torch.softmax(mask, -1).reshape(N, H, W, group * P)
    pad = kernel_size // 2
    if PROBE is not None:
        # a 3 x 3 / stride 1 / dilation 1 sample sits at integer + offset * offset_scale (dcnv3_oracle._locations): bilinear interpolation is
        # only piecewise differentiable, the kinks are where that fractional part is 0 -- record the closest approach (in pixels)
        with torch.no_grad():
            fr = (offset.detach().double() * offset_scale) % 1.0
            d = float(torch.minimum(fr, 1.0 - fr).min())
        PROBE["min_edge_distance"] = min(PROBE.get("min_edge_distance", 1.0), d)
        PROBE["calls"] = PROBE.get("calls", 0) + 1
    y = D.dcnv3_forward(xp, offset, mask, kernel_size, kernel_size, 1, 1, pad, pad, 1, 1, group, C // group, offset_scale, 0)
    return F.linear(y, p[pre + "output_proj.weight"], p[pre + "output_proj.bias"])


def layer(x, p, pre, group, offset_scale):
    """InternImageLayer II:407-434, the layer_scale + post_norm branch (II:424-426): x += g1 * LN1(dcn(x)); x += g2 * LN2(mlp(x))"""
    x = x + p[pre + "gamma1"] * _ln(dcnv3_module(x, p, pre + "dcn.", group, offset_scale), p, pre + "norm1.0")
    h = F.gelu(F.linear(x, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))           # MLPLayer II:327-333, dropout p = 0
    h = F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    return x + p[pre + "gamma2"] * _ln(h, p, pre + "norm2.0")


def downsample(x, p, pre):
    """DownsampleLayer II:279-300: conv3x3 s2 without bias (C -> 2C) -> LN"""
    return _ln(_conv_nhwc(x, p[pre + "conv.weight"], None, 2), p, pre + "norm.1")


def backbone_forward(img, p, depths, groups, offset_scale=2.0, out_indices=(0, 1, 2, 3)):
    """InternImage.forward II:690-698 (+ InternImageBlock.forward II:509-525 with post_norm: no level norm): list of NCHW maps
    at strides 4, 8, 16, 32 taken before each level's downsample"""
    x = stem(img, p)
    outs = []
    for i, (depth, G) in enumerate(zip(depths, groups)):
        for j in range(depth):
            x = layer(x, p, "levels.%d.blocks.%d." % (i, j), G, offset_scale)
        if i in out_indices:
            outs.append(x.permute(0, 3, 1, 2).contiguous())
        if i < len(depths) - 1:
            x = downsample(x, p, "levels.%d.downsample." % i)
    return outs
