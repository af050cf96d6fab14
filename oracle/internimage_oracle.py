"""ORACLE (test infrastructure only) -- CPU restatement of the InternImage backbone around the DCNv3 core.

Groundwork for the next widening step: the HIP operator (mtp_amd/ops_dcnv3) covers the reference's native extension; the
layers around it are plain torch modules in the reference ("II" = /root/reference/Multi-Task_Pretrain/backbone/intern_image.py,
"DCNM" = .../backbone/ops_dcnv3/modules/dcnv3.py).  This file restates them channels-last, from a flat parameter dict with the
reference's state-dict keys, for norm_layer='LN', act_layer='GELU': the configuration family BASELINE config 5 uses (II:700-712:
layer_scale set, post_norm=True) and, since round 6, the other layer branches (pre-norm, no layer scale, res_post_norm, level-2 post
norms, center_feature_scale, dw_kernel_size).
The DCNv3 core inside is oracle/dcnv3_oracle.py (explicit bilinear gather), so gradients flow through torch autograd.

PINNED: tests/golden/f12_internimage.npz (and f15_internimage_variants.npz for the other branches) = outputs and gradients of the
reference's own `InternImage(core_op='DCNv3_pytorch')` (tests/golden/make_golden.py f12 / f15); tests/test_internimage_oracle.py holds this file to them (fp32, 2e-5) and checks
`state_shapes()` against the reference's state_dict() keys / shapes / order stored in the fixture.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import torch
import torch.nn.functional as F

from . import dcnv3_oracle as D

EPS = 1e-6   # build_norm_layer default (II:39-43, DCNM:37-41)


def state_shapes(channels=192, depths=(5, 5, 24, 5), groups=(12, 24, 48, 96), mlp_ratio=4.0, kernel_size=3, post_norm=True, layer_scale=True,
                 res_post_norm=False, level2_post_norm_block_ids=None, dw_kernel_size=None, center_feature_scale=False):
    """reference state-dict keys and shapes in the reference's order (own parameters of a module before its sub-modules'): the defaults are the
    layer_scale + post_norm family of BASELINE config 5; the other branches of II:407-427 / II:497-502 by the keyword flags (fixture f15)"""
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
            if layer_scale:
                s[p + "gamma1"] = (C,)
                s[p + "gamma2"] = (C,)
            s[p + "norm1.0.weight"] = (C,)
            s[p + "norm1.0.bias"] = (C,)
            if center_feature_scale:                        # DCNM:168-172: own parameters of the module, ahead of its sub-modules
                s[p + "dcn.center_feature_scale_proj_weight"] = (G, C)
                s[p + "dcn.center_feature_scale_proj_bias"] = (G,)
            s[p + "dcn.dw_conv.0.weight"] = (C, 1, dw_kernel_size or kernel_size, dw_kernel_size or kernel_size)
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
            if res_post_norm:
                for k in (1, 2):
                    s[p + "res_post_norm%d.0.weight" % k] = (C,)
                    s[p + "res_post_norm%d.0.bias" % k] = (C,)
        if not post_norm or center_feature_scale:           # II:497-498
            s["levels.%d.norm.0.weight" % i] = (C,)
            s["levels.%d.norm.0.bias" % i] = (C,)
        if level2_post_norm_block_ids and i == 2:           # II:499-502, 593-594
            for k in range(len(level2_post_norm_block_ids)):
                s["levels.2.post_norms.%d.0.weight" % k] = (C,)
                s["levels.2.post_norms.%d.0.bias" % k] = (C,)
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
    if (pre + "center_feature_scale_proj_weight") in p:      # DCNM:80-88, 209-215: a per-group gate between the sampled and the projected features
        sc = torch.sigmoid(F.linear(x1, p[pre + "center_feature_scale_proj_weight"], p[pre + "center_feature_scale_proj_bias"]))
        sc = sc[..., None].expand(N, H, W, group, C // group).reshape(N, H, W, C)
        y = y * (1 - sc) + xp * sc
    return F.linear(y, p[pre + "output_proj.weight"], p[pre + "output_proj.bias"])


def _mlp(x, p, pre):
    h = F.gelu(F.linear(x, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))           # MLPLayer II:327-333, dropout p = 0
    return F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


def layer(x, p, pre, group, offset_scale, post_norm=True):
    """InternImageLayer II:407-434, every branch (the reference picks it from `post_norm`, `res_post_norm` and whether the layer has gammas; here the
    presence of the gamma / res_post_norm parameters in `p` says the same):
      post_norm:      x += g1 * LN1(dcn(x));            x += g2 * LN2(mlp(x))             (II:409-411, 424-425; g = 1 without layer scale)
      res_post_norm:  x += RPN1(dcn(LN1(x)));           x += RPN2(mlp(LN2(x)))            (II:412-414)
      pre-norm:       x += g1 * dcn(LN1(x));            x += g2 * mlp(LN2(x))             (II:415-417, 426-427)"""
    g1 = p.get(pre + "gamma1", 1.0)
    g2 = p.get(pre + "gamma2", 1.0)
    if post_norm:
        x = x + g1 * _ln(dcnv3_module(x, p, pre + "dcn.", group, offset_scale), p, pre + "norm1.0")
        return x + g2 * _ln(_mlp(x, p, pre), p, pre + "norm2.0")
    rpn = (pre + "res_post_norm1.0.weight") in p
    h = dcnv3_module(_ln(x, p, pre + "norm1.0"), p, pre + "dcn.", group, offset_scale)
    x = x + g1 * (_ln(h, p, pre + "res_post_norm1.0") if rpn else h)
    h = _mlp(_ln(x, p, pre + "norm2.0"), p, pre)
    return x + g2 * (_ln(h, p, pre + "res_post_norm2.0") if rpn else h)


def downsample(x, p, pre):
    """DownsampleLayer II:279-300: conv3x3 s2 without bias (C -> 2C) -> LN"""
    return _ln(_conv_nhwc(x, p[pre + "conv.weight"], None, 2), p, pre + "norm.1")


def backbone_forward(img, p, depths, groups, offset_scale=2.0, out_indices=(0, 1, 2, 3), post_norm=True, level2_post_norm_block_ids=None):      # (the level norm is applied when its parameters exist)
    """InternImage.forward II:690-698 (+ InternImageBlock.forward II:509-525: the level-2 post norms behind their blocks, the level's closing norm of the
    pre-norm forms): list of NCHW maps at strides 4, 8, 16, 32 taken before each level's downsample"""
    x = stem(img, p)
    outs = []
    for i, (depth, G) in enumerate(zip(depths, groups)):
        ids = list(level2_post_norm_block_ids) if (level2_post_norm_block_ids and i == 2) else []
        for j in range(depth):
            x = layer(x, p, "levels.%d.blocks.%d." % (i, j), G, offset_scale, post_norm)
            if j in ids:
                x = _ln(x, p, "levels.2.post_norms.%d.0" % ids.index(j))
        if ("levels.%d.norm.0.weight" % i) in p:      # II:516-517: `not post_norm or center_feature_scale`
            x = _ln(x, p, "levels.%d.norm.0" % i)
        if i in out_indices:
            outs.append(x.permute(0, 3, 1, 2).contiguous())
        if i < len(depths) - 1:
            x = downsample(x, p, "levels.%d.downsample." % i)
    return outs
