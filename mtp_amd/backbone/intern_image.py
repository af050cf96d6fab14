"""InternImage backbone on the MI355X HIP operators behind the reference's class surface (BASELINE config 5).

Mirrors Multi-Task_Pretrain/backbone/intern_image.py ("II"):
  * class `InternImage(nn.Module)` with the constructor keywords of II:552-574, `forward(x)` -> list of NCHW maps taken before
    each level's downsample (II:690-698), attributes `num_levels`, `depths`, `channels`, `num_features`, `out_indices`,
    `out_channels`, `num_layers`, and `init_weights(pretrained)` with the reference's prefix-stripping rules (II:639-670);
  * state-dict keys, shapes and order identical to the reference (`levels.0.blocks.0.dcn.dw_conv.1.1.weight`, ...), pinned by
    fixture f12 (tests/test_hip_internimage.py);
  * initialisation rules of II:672-686 + DCNv3._reset_parameters (ops_dcnv3/modules/dcnv3.py:308-316): Linear trunc_normal(.02)
    / zero bias, LayerNorm 1 / 0, offset and mask heads zero, input / output projections xavier-uniform, layer scale
    `layer_scale * ones`, convolutions with nn.Conv2d's default init.
The configuration family MTP uses (models.py:92-104: norm 'LN', act 'GELU', layer_scale set, post_norm=True) is what the engine
schedules; the InternImage-H/G options (dw_kernel_size, level2_post_norm, res_post_norm, center_feature_scale), pre-norm and
`layer_scale=None` raise NotImplementedError instead of silently running something else.  `with_cp` (activation checkpointing,
II:429-430) is honoured since round 5: a layer keeps its input only and its forward is run again inside the backward (same results, bit for bit;
2 instead of 12 saved row tensors per layer, one more forward of time) -- `internimage_xl()` sets it, as models.py:92-104 does.
All compute runs in libmtp_hip.so (mtp_amd/engine_intern.py); there is no CPU / eager fallback.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from ..registry import BACKBONES, MODELS


def _trunc_normal_(t, std):
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)      # timm's default cut-offs (II:14)


class _Holder(nn.Module):
    """parameter container node: the module tree only exists to reproduce the reference's state-dict names"""

    def forward(self, *a, **k):
        raise RuntimeError("mtp_amd.InternImage sub-modules are parameter holders: call the backbone (the HIP engine schedules the kernels)")


def _put(root, dotted, tensor):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Holder())
        node = node._modules[p]
    node.register_parameter(parts[-1], nn.Parameter(tensor))


class _InternFn(torch.autograd.Function):
    """One autograd node for the whole backbone: forward / backward are the engine's explicit kernel schedules."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module._engine()
        need = any(ctx.needs_input_grad)
        feats, ectx = eng.forward(x, training=module.training, need_grad=need, feature_dtype=module._feature_dtype())
        ctx.module, ctx.ectx, ctx.x_grad = module, ectx, x.requires_grad
        ctx.names = [n for n, _ in module.named_parameters()]
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        if ctx.ectx is None:
            raise RuntimeError("backward through a forward that ran without saved activations")
        module = ctx.module
        P = dict(module.named_parameters())
        G = {n: torch.zeros_like(P[n], dtype=torch.float32) for n in ctx.names}
        dimg = module._engine().backward(ctx.ectx, list(dfeats), G, need_input_grad=ctx.x_grad)
        ctx.ectx = None
        return (None, dimg) + tuple(G[n] for n in ctx.names)


class InternImage(nn.Module):
    def __init__(self, core_op="DCNv3", channels=64, depths=[3, 4, 18, 5], groups=[3, 6, 12, 24], mlp_ratio=4., drop_rate=0.,
                 drop_path_rate=0.2, drop_path_type="linear", act_layer="GELU", norm_layer="LN", layer_scale=None, offset_scale=1.0,
                 post_norm=False, with_cp=False, dw_kernel_size=None, level2_post_norm=False, level2_post_norm_block_ids=None,
                 res_post_norm=False, center_feature_scale=False, out_indices=(0, 1, 2, 3), init_cfg=None,
                 precision="bf16", feature_dtype=None):
        super().__init__()
        if core_op not in ("DCNv3", "DCNv3_pytorch"):
            raise NotImplementedError("core_op %r" % (core_op,))
        # every option of II:367-416 / 499-515 is scheduled (round 6): post_norm or pre-norm, with or without layer scale, res_post_norm, the level-2 post norms,
        # and InternImage-H/G's two switches inside the DCNv3 module -- a depth-wise kernel of any odd size (DCNM:124, 146-151: plain k x k kernels; 3 x 3 keeps the
        # fast path) and center_feature_scale (DCNM:168-173, 209-215)
        if dw_kernel_size is not None and (int(dw_kernel_size) < 1 or int(dw_kernel_size) % 2 == 0 or int(dw_kernel_size) > 15):
            raise ValueError("dw_kernel_size must be odd and <= 15, got %r" % (dw_kernel_size,))
        if act_layer != "GELU" or norm_layer != "LN" or drop_rate != 0.0:
            raise NotImplementedError("the HIP path schedules act_layer='GELU', norm_layer='LN', drop_rate=0 (what MTP uses, models.py:92-104)")
        if res_post_norm and (post_norm or layer_scale is not None):
            # (II:408-417: res_post_norm is only looked at in the branch without layer scale and without post_norm; refusing the combinations the reference
            #  would silently build as something else keeps a configuration's meaning the same on both sides)
            raise ValueError("res_post_norm takes effect only with post_norm=False and layer_scale=None (II:408-417)")
        if level2_post_norm and not level2_post_norm_block_ids:
            raise ValueError("level2_post_norm needs level2_post_norm_block_ids")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.core_op = core_op
        self.num_levels = len(depths)
        self.depths = list(depths)
        self.groups = list(groups)
        self.channels = channels
        self.num_features = int(channels * 2 ** (self.num_levels - 1))
        self.post_norm = post_norm
        self.mlp_ratio = mlp_ratio
        self.init_cfg = init_cfg
        self.out_indices = tuple(out_indices)
        self.level2_post_norm_block_ids = level2_post_norm_block_ids
        self.level2_post_norm = bool(level2_post_norm)
        self.res_post_norm = bool(res_post_norm)
        self.dw_kernel_size = int(dw_kernel_size) if dw_kernel_size is not None else 3
        self.center_feature_scale = bool(center_feature_scale)
        self.offset_scale = float(offset_scale)
        self.has_level_norm = (not post_norm) or bool(center_feature_scale)      # II:497, 516
        self.has_layer_scale = layer_scale is not None
        self.layer_scale = float(layer_scale) if layer_scale is not None else None
        self.kernel_size = 3
        self.with_cp = with_cp
        self.precision = precision
        self.feature_dtype = feature_dtype
        for i, (c, g) in enumerate(zip([channels * 2 ** i for i in range(self.num_levels)], groups)):
            if c % g or (c // g) % 4:
                raise ValueError("level %d: channels %d must split into groups of a multiple of 4 channels (got %d groups)" % (i, c, g))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]        # II:602-607
        if drop_path_type == "uniform":
            dpr = [drop_path_rate] * len(dpr)
        self.drop_path_rates = dpr
        self._build_parameters()
        self.num_layers = len(depths)
        self.out_channels = [192, 384, 768, 1536]      # (II:637: hard-coded in the reference whatever `channels` is)
        self._eng = None

    # ------------------------------------------------------------------ parameters (reference names / shapes / order / init)
    def _build_parameters(self):
        ch, P = self.channels, self.kernel_size ** 2

        def conv(cout, cin, k, groups=1, bias=True):
            w = torch.empty(cout, cin // groups, k, k)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            b = None
            if bias:
                bound = 1.0 / math.sqrt((cin // groups) * k * k)
                b = torch.empty(cout).uniform_(-bound, bound)
            return w, b

        def linear(r, c, kind):
            w = torch.empty(r, c)
            if kind == "zero":
                w.zero_()
            elif kind == "xavier":
                nn.init.xavier_uniform_(w)
            else:
                _trunc_normal_(w, 0.02)
            return w, torch.zeros(r)

        def ln(pre, c):
            _put(self, pre + ".weight", torch.ones(c))
            _put(self, pre + ".bias", torch.zeros(c))

        c2 = ch // 2
        w, b = conv(c2, 3, 3)
        _put(self, "patch_embed.conv1.weight", w)
        _put(self, "patch_embed.conv1.bias", b)
        ln("patch_embed.norm1.1", c2)
        w, b = conv(ch, c2, 3)
        _put(self, "patch_embed.conv2.weight", w)
        _put(self, "patch_embed.conv2.bias", b)
        ln("patch_embed.norm2.1", ch)
        for i, (depth, G) in enumerate(zip(self.depths, self.groups)):
            C = ch * 2 ** i
            hid = int(C * self.mlp_ratio)
            for j in range(depth):
                p = "levels.%d.blocks.%d." % (i, j)
                if self.has_layer_scale:
                    _put(self, p + "gamma1", self.layer_scale * torch.ones(C))
                    _put(self, p + "gamma2", self.layer_scale * torch.ones(C))
                ln(p + "norm1.0", C)
                if self.center_feature_scale:      # (own parameters of the DCNv3 module: ahead of its sub-modules in the state dict; DCNM:168-172: zeros)
                    _put(self, p + "dcn.center_feature_scale_proj_weight", torch.zeros(G, C))
                    _put(self, p + "dcn.center_feature_scale_proj_bias", torch.zeros(G))
                w, b = conv(C, C, self.dw_kernel_size, groups=C)
                _put(self, p + "dcn.dw_conv.0.weight", w)
                _put(self, p + "dcn.dw_conv.0.bias", b)
                ln(p + "dcn.dw_conv.1.1", C)
                for name, rows, kind in (("offset", G * P * 2, "zero"), ("mask", G * P, "zero"), ("input_proj", C, "xavier"), ("output_proj", C, "xavier")):
                    w, b = linear(rows, C, kind)
                    _put(self, p + "dcn.%s.weight" % name, w)
                    _put(self, p + "dcn.%s.bias" % name, b)
                ln(p + "norm2.0", C)
                w, b = linear(hid, C, "trunc")
                _put(self, p + "mlp.fc1.weight", w)
                _put(self, p + "mlp.fc1.bias", b)
                w, b = linear(C, hid, "trunc")
                _put(self, p + "mlp.fc2.weight", w)
                _put(self, p + "mlp.fc2.bias", b)
                if self.res_post_norm:
                    ln(p + "res_post_norm1.0", C)
                    ln(p + "res_post_norm2.0", C)
            if self.has_level_norm:                      # II:497-498: the level's closing norm of the pre-norm forms (and of center_feature_scale models)
                ln("levels.%d.norm.0" % i, C)
            for k in range(len(self.post_norm_ids(i))):   # II:499-502
                ln("levels.%d.post_norms.%d.0" % (i, k), C)
            if i < self.num_levels - 1:
                p = "levels.%d.downsample." % i
                w, _ = conv(2 * C, C, 3, bias=False)
                _put(self, p + "conv.weight", w)
                ln(p + "norm.1", 2 * C)

    def post_norm_ids(self, level):
        """the blocks of `level` that are followed by an extra LayerNorm (II:593-594: level 2 only, when level2_post_norm is set)"""
        return list(self.level2_post_norm_block_ids) if (self.level2_post_norm and level == 2) else []

    def init_weights(self, pretrained):
        """II:639-670: checkpoint dict -> 'state_dict' / 'model' / itself, strip 'backbone.' and 'module.' prefixes, non-strict load"""
        ckpt = torch.load(pretrained, map_location="cpu")
        sd = ckpt.get("state_dict", ckpt.get("model", ckpt)) if isinstance(ckpt, dict) else ckpt
        out = OrderedDict((k[9:] if k.startswith("backbone.") else k, v) for k, v in sd.items())
        if out and next(iter(out)).startswith("module."):
            out = OrderedDict((k[7:], v) for k, v in out.items())
        return self.load_state_dict(out, strict=False)

    # ------------------------------------------------------------------ data-parallel training hooks (mtp_amd.parallel)
    _unused_params = frozenset()

    def _flat_param_order(self):
        """(names in reverse execution order, name -> group id, number of layer groups): group = global layer index; a level's
        downsample completes together with the level's last layer; the stem is group -1 (mtp_amd.parallel.FlatParams)"""
        base = [sum(self.depths[:i]) for i in range(self.num_levels + 1)]
        groups = {}
        for n, _ in self.named_parameters():
            parts = n.split(".")
            if parts[0] == "levels":
                i = int(parts[1])
                if parts[2] == "blocks":
                    groups[n] = base[i] + int(parts[3])
                elif parts[2] == "post_norms":      # its gradient is written right before the backward of the block it follows
                    groups[n] = base[i] + self.post_norm_ids(i)[int(parts[3])]
                else:                               # downsample, the level's closing norm
                    groups[n] = base[i + 1] - 1
            else:
                groups[n] = -1
        names = sorted(groups, key=lambda k: -groups[k])        # stable: the module's order inside a group
        return names, groups, base[-1]

    @staticmethod
    def _overwritten_grads(name):
        """gradients the engine writes with a full overwrite whenever it writes them (mtp_amd.parallel.FlatParams skips clearing them; 98 % of the buffer): the Linear
        and dense-convolution weights -- grouped TN GEMM outputs.  A level no cotangent reaches is skipped by the backward, which then clears these itself
        (InternEngine.backward), so an unused tap still leaves zeros."""
        return name.endswith((".dcn.input_proj.weight", ".dcn.output_proj.weight", ".dcn.offset.weight", ".dcn.mask.weight", ".mlp.fc1.weight", ".mlp.fc2.weight",
                              ".downsample.conv.weight", "patch_embed.conv1.weight", "patch_embed.conv2.weight"))

    # ------------------------------------------------------------------ execution
    def _engine(self):
        from ..engine_intern import InternEngine
        act = torch.bfloat16 if self.precision == "bf16" else torch.float32
        if self._eng is None or self._eng.act != act:
            self._eng = InternEngine(self, act)
        return self._eng

    def _feature_dtype(self):
        if self.feature_dtype is not None:
            return self.feature_dtype
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    def forward(self, x):
        """II:690-698: (N, 3, H, W) -> [ (N, C_i, H / 2^(i+2), W / 2^(i+2)) for i in out_indices ]"""
        if not x.is_cuda:
            raise RuntimeError("mtp_amd.InternImage runs on an MI355X (gfx950) device only: there is no CPU / eager PyTorch fallback.  "
                               "Move the module and the input to 'cuda'.")
        params = [p for _, p in self.named_parameters()]
        return list(_InternFn.apply(self, x, *params))


MODELS.register_module(module=InternImage, force=True)
BACKBONES.register_module(module=InternImage, force=True)


def internimage_xl(**kw):
    """the backbone MTP builds for --backbone internimage_xl (models.py:92-104)"""
    cfg = dict(core_op="DCNv3", channels=192, depths=[5, 5, 24, 5], groups=[12, 24, 48, 96], mlp_ratio=4., drop_path_rate=0.2, norm_layer="LN",
               layer_scale=1e-5, offset_scale=2.0, post_norm=True, with_cp=True, out_indices=(0, 1, 2, 3))
    cfg.update(kw)
    return InternImage(**cfg)
