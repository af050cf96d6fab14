"""ViT + RVSA backbone with the reference's Python surface, running on the gfx950 HIP kernels.

Drop-in for `Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py` ("VIT") of ViTAE-Transformer/MTP:
  * class `ViT_Win_RVSA_V3_WSZ7(nn.Module)` with the constructor kwargs of VIT:590-594, `forward_features(x)` returning
    the list of four NCHW feature maps (VIT:787-813), `forward = forward_features`, `init_weights(pretrained)`
    (VIT:693-778), `get_num_layers()` (VIT:780), `no_weight_decay()` (VIT:784), `out_channels` (VIT:674);
  * identical state-dict keys and shapes, including the unused `norm.*` (VIT:638) and the int64 buffer
    `attn.relative_position_index` (VIT:282), so released MTP encoder checkpoints load unchanged;
  * factories `vit_b_rvsa(args)` / `vit_l_rvsa(args)` (VIT:819-865);
  * registry names `ViT_Win_RVSA_V3_WSZ7`, `RVSA_MTP`, `RVSA_MTP_branches` (RS_Tasks_Finetune/*/backbones/vit_rvsa_mtp*.py).

The sub-modules (nn.Linear, nn.LayerNorm, nn.Conv2d, nn.ConvTranspose2d) are PARAMETER CONTAINERS only: they give the
reference's names and default initialisation, but are never called.  All compute goes through
mtp_amd.engine.BackboneEngine -> libmtp_hip.so; there is no PyTorch/CPU fallback (CPU inputs raise).
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ..registry import BACKBONES, MODELS


def _to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _relative_position_index(ws):
    """VIT:271-281, closed form (bit-exact, tests/test_host.py): (a_n - a_m + ws-1)*(2ws-1) + (b_n - b_m + ws-1)."""
    a = torch.arange(ws).repeat_interleave(ws)
    b = torch.arange(ws).repeat(ws)
    return (a[:, None] - a[None, :] + ws - 1) * (2 * ws - 1) + (b[:, None] - b[None, :] + ws - 1)


def window_partition(x, window_size):
    """VIT:113-124: (B,H,W,C) -> (num_windows*B, ws, ws, C).  Pure index permutation (dead code in the reference's
    forward, kept for API parity)."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """VIT:127-140: inverse of window_partition."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class _NoCall(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("this sub-module is a parameter container; the backbone computes through mtp_amd's HIP engine")


class Mlp(_NoCall):
    """VIT:45-62 parameter layout: fc1, fc2."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(_NoCall):
    """VIT:65-88 parameter layout (full MHSA with decomposed rel-pos)."""

    def __init__(self, dim, num_heads, qkv_bias, window_size, rel_pos=True):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.window_size = window_size
        rel_sp_dim = 2 * window_size[0] - 1
        if rel_pos:   # (the ViTDet-style fine-tune copies have these two lines commented out, mmdet vit_rvsa_mtp.py:73-74)
            self.full_attn_rel_pos_h = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
            self.full_attn_rel_pos_w = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
        self.proj = nn.Linear(dim, dim)


class RotatedVariedSizeWindowAttention(_NoCall):
    """VIT:195-285 parameter layout.  The three sampling heads keep nn.Conv2d's default init (VIT:440-445 is never
    called in the reference)."""

    def __init__(self, dim, num_heads, qkv_bias, window_size=7):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.ws = window_size
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * window_size - 1, head_dim))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * window_size - 1, head_dim))
        self.sampling_offsets = nn.Sequential(nn.AvgPool2d(window_size, window_size), nn.LeakyReLU(), nn.Conv2d(dim, num_heads * 2, 1))
        self.sampling_scales = nn.Sequential(nn.AvgPool2d(window_size, window_size), nn.LeakyReLU(), nn.Conv2d(dim, num_heads * 2, 1))
        self.sampling_angles = nn.Sequential(nn.AvgPool2d(window_size, window_size), nn.LeakyReLU(), nn.Conv2d(dim, num_heads, 1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class Block(_NoCall):
    """VIT:479-504 parameter layout."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer, window_size, window, drop_path=0.0, full_rel_pos=True, init_values=None):
        super().__init__()
        if init_values is not None:     # layer scale (VIT:500-502); a module's own parameters precede its children's in state_dict(), as in the reference
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self.norm1 = norm_layer(dim)
        if window:
            self.attn = RotatedVariedSizeWindowAttention(dim, num_heads, qkv_bias, window_size[0])
        else:
            self.attn = Attention(dim, num_heads, qkv_bias, window_size, rel_pos=full_rel_pos)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.drop_path_prob = drop_path


class PatchEmbed(_NoCall):
    """VIT:515-529 parameter layout + shape attributes."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = _to_2tuple(img_size), _to_2tuple(patch_size)
        self.patch_shape = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Norm2d(_NoCall):
    """VIT:576-584 parameter layout."""

    def __init__(self, embed_dim):
        super().__init__()
        self.ln = nn.LayerNorm(embed_dim, eps=1e-6)


class _BackboneFn(torch.autograd.Function):
    """One autograd node for the whole backbone: forward/backward are the engine's explicit kernel schedules."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module._engine()
        # grad mode is off inside Function.forward; needs_input_grad is all-False under torch.no_grad()
        need = any(ctx.needs_input_grad)
        feats, ectx = eng.forward(x, training=module.training, need_grad=need, feature_dtype=module._feature_dtype(x))
        ctx.module, ctx.ectx, ctx.n = module, ectx, len(params)
        ctx.x_grad = x.requires_grad
        ctx.names = module._param_names
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        if ctx.ectx is None:
            raise RuntimeError("backward through a forward that ran without saved activations")
        module = ctx.module
        eng = module._engine()
        P = dict(module.named_parameters())
        G = {n: torch.zeros_like(P[n], dtype=torch.float32) for n in ctx.names}
        dimg = eng.backward(ctx.ectx, list(dfeats), G, need_input_grad=ctx.x_grad)
        ctx.ectx = None
        grads = []
        for n in ctx.names:
            # parameters that never receive a gradient (`norm.*`, VIT:638) report None like the reference's autograd
            grads.append(None if n in module._unused_params else G[n])
        return (None, dimg) + tuple(grads)


class ViT_Win_RVSA_V3_WSZ7(nn.Module):
    """Vision Transformer with RVSA window attention (VIT:587-817), MI355X-native."""

    _vitdet = False   # True in RVSA_MTP_det: the mmdet / mmrotate fine-tune copies' ViTDet-style forward
    _taps_only = False   # True in RVSA_MTP_taps: the mmpretrain / opencd copies return the taps without the fpn ops

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=None, init_values=None, use_checkpoint=False,
                 use_abs_pos_emb=False, use_rel_pos_bias=False, use_shared_rel_pos_bias=False,
                 out_indices=[11], interval=3, pretrained=None, restart_regression=True,
                 precision="bf16", feature_dtype=None):
        super().__init__()
        if hybrid_backbone is not None:
            # the reference accepts the keyword but cannot run it: with a full-attention block in the schedule the constructor reads
            # `self.patch_embed.patch_shape` (VIT:628), which HybridEmbed (VIT:543-573) does not define -> AttributeError; without one, forward_features
            # unpacks `x, (Hp, Wp) = self.patch_embed(x)` (VIT:790) from HybridEmbed.forward's single tensor -> ValueError.  Checked against the reference
            # itself; there is no behaviour to be a drop-in for.
            raise NotImplementedError("hybrid_backbone: the reference's HybridEmbed (VIT:543-573) cannot be constructed or run with this class (VIT:628 reads "
                                      "patch_embed.patch_shape, VIT:790 unpacks a tuple HybridEmbed.forward does not return) -- nothing to reproduce")
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("dropout is p=0 in both MTP factories (VIT:833-834)")
        if (embed_dim // num_heads) != 64:
            raise NotImplementedError("the HIP attention kernels are built for head_dim == 64 (ViT-B/L)")
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.in_chans = in_chans
        self.num_heads = num_heads
        self.qk_scale = qk_scale
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.out_indices = out_indices
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim)) if use_abs_pos_emb else None
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # VIT:619
        self.drop_path_rates = dpr
        self.use_rel_pos_bias = use_rel_pos_bias
        self.use_checkpoint = use_checkpoint
        self.window_blocks = [((i + 1) % interval != 0) for i in range(depth)]   # VIT:629
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer,
                  window_size=(7, 7) if self.window_blocks[i] else self.patch_embed.patch_shape,
                  window=self.window_blocks[i], drop_path=dpr[i], full_rel_pos=not self._vitdet, init_values=init_values)
            for i in range(depth)])
        self.interval = interval
        if self.pos_embed is not None:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.norm = norm_layer(embed_dim)        # present in the state dict, never applied (VIT:638, 813)
        if patch_size == 16:
            self.fpn1 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=2, stride=2), Norm2d(embed_dim), nn.GELU(),
                                      nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=2, stride=2))
            self.fpn2 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=2, stride=2))
            self.fpn3 = nn.Identity()
            self.fpn4 = nn.MaxPool2d(kernel_size=2, stride=2)
        elif patch_size == 8:       # VIT:656-670 (unused by MTP's factories; fixture f14)
            self.fpn1 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, kernel_size=2, stride=2))
            self.fpn2 = nn.Identity()
            self.fpn3 = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=2))
            self.fpn4 = nn.Sequential(nn.MaxPool2d(kernel_size=4, stride=4))
        else:
            raise NotImplementedError("patch_size must be 16 or 8 (the reference defines no FPN tail for anything else, VIT:640-670)")
        self.init_values = init_values
        self.apply(self._init_weights)
        self.fix_init_weight()
        self.pretrained = pretrained
        self.out_channels = [embed_dim, embed_dim, embed_dim, embed_dim]
        self.precision = precision
        self.feature_dtype = feature_dtype
        self._eng = None
        self.data_preprocessor = None
        self._param_names = [n for n, _ in self.named_parameters()]
        if self._vitdet:      # last block -> norm -> fpn1-4: every parameter is on the path
            self._unused_params = set()
        else:
            self._unused_params = {"norm.weight", "norm.bias"}
            last = max(out_indices)
            self._unused_params |= {n for n in self._param_names if n.startswith("blocks.") and int(n.split(".")[1]) > last}
            if self._taps_only:   # the fpn modules exist (state-dict parity) but their ops are not applied
                self._unused_params |= {n for n in self._param_names if n.startswith("fpn")}

    # ---- reference API -------------------------------------------------------------------------------------------
    def fix_init_weight(self):
        """VIT:676-682."""
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    def _init_weights(self, m):
        """VIT:684-691."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def init_weights(self, pretrained=None):
        """VIT:693-778 (and the no-argument form of the fine-tune copies, vit_rvsa_mtp.py:684): load a checkpoint given
        as `state_dict` / `model` / raw mapping, strip `module.`, keep `encoder.*`, drop `patch_embed.proj` when
        in_chans != 3, bicubic-resize `pos_embed` assuming one extra (cls) token."""
        return self._load_pretrained(pretrained or self.pretrained, finetune=False)

    def _load_pretrained(self, pretrained, finetune):
        """finetune=True: the loader shared by all nine fine-tune copies (e.g. RS_Tasks_Finetune/Semantic_Segmentation/mmseg/
        models/backbones/vit_rvsa_mtp.py:684-805), which differs from the pretrain-side one (VIT:693-778) in three places:
        `patch_embed.proj` is kept whatever in_chans is (:727-733 commented out); the `full_attn_rel_pos_h/w` tables are
        bicubically resized to this model's (2*Hp-1, head_dim) (:735-765); one extra token is stripped from `pos_embed`
        only when the checkpoint has a `cls_token` key (:775-778)."""
        if isinstance(pretrained, str):
            self.apply(self._init_weights)
            # weights_only=False as the reference's torch.load (VIT:711) effectively was: its own checkpoints carry a numpy
            # array ('loss_pretrain', MAIN:826) and optimizer / scheduler dicts next to 'state_dict'
            checkpoint = torch.load(pretrained, map_location="cpu", weights_only=False)
            if "state_dict" in checkpoint:
                state_dict = checkpoint["state_dict"]
            elif "model" in checkpoint:
                state_dict = checkpoint["model"]
            else:
                state_dict = checkpoint
            if list(state_dict.keys())[0].startswith("module."):
                state_dict = {k[7:]: v for k, v in state_dict.items()}
            if sorted(list(state_dict.keys()))[0].startswith("encoder"):
                state_dict = {k.replace("encoder.", ""): v for k, v in state_dict.items() if k.startswith("encoder.")}
            if self.in_chans != 3 and not finetune:
                for k in list(state_dict.keys()):
                    if "patch_embed.proj" in k:
                        del state_dict[k]
            if finetune:
                own = next((p for n, p in self.named_parameters() if "attn.full_attn_rel_pos_h" in n), None)
                if own is not None:
                    for k in list(state_dict.keys()):
                        if "full_attn_rel_pos_h" in k or "full_attn_rel_pos_w" in k:
                            old = state_dict[k]
                            new = torch.nn.functional.interpolate(old.reshape(1, 1, old.shape[0], old.shape[1]), size=tuple(own.shape),
                                                                  mode="bicubic", align_corners=False)
                            state_dict[k] = new.squeeze()
            if "pos_embed" in state_dict:
                pe = state_dict["pos_embed"]
                emb = pe.shape[-1]
                H, W = self.patch_embed.patch_shape
                extra = (1 if "cls_token" in state_dict else 0) if finetune else 1
                orig = int((pe.shape[-2] - extra) ** 0.5)
                new = int(self.patch_embed.num_patches ** 0.5)
                if orig != new:
                    tok = pe[:, extra:].reshape(-1, orig, orig, emb).permute(0, 3, 1, 2)
                    tok = torch.nn.functional.interpolate(tok, size=(H, W), mode="bicubic", align_corners=False)
                    state_dict["pos_embed"] = tok.permute(0, 2, 3, 1).flatten(1, 2)
                else:
                    state_dict["pos_embed"] = pe[:, extra:]
            return self.load_state_dict(state_dict, False)
        elif pretrained is None:
            self.apply(self._init_weights)
        else:
            raise TypeError("pretrained must be a str or None")

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    # ---- HIP path ----------------------------------------------------------------------------------------------------
    def set_precision(self, precision):
        """'bf16' (bf16 MFMA, f32 accumulate/statistics/residual stream) or 'fp32' (exact-f32 MFMA: the 1e-3 parity mode)."""
        assert precision in ("bf16", "fp32")
        self.precision = precision
        self._eng = None
        return self

    def set_data_preprocessor(self, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), bgr_to_rgb=True,
                              pad_size_divisor=32, pad_value=0.0):
        """Optional input side: with this set, forward_features also accepts the raw (B, H, W, 3) uint8 batch
        and applies MTP_DataPreprocessor's image path (defaults = its configuration at Multi-Task_Pretrain/models.py:37-41:
        BGR->RGB, (x - mean) / std, pad bottom/right to a multiple of 32) inside the patch-embed im2col kernel."""
        assert len(mean) == 3 and len(std) == 3 and all(float(s) != 0.0 for s in std)
        self.data_preprocessor = dict(mean=tuple(float(v) for v in mean), std=tuple(float(v) for v in std), bgr_to_rgb=bool(bgr_to_rgb),
                                      pad_size_divisor=int(pad_size_divisor), pad_value=float(pad_value))
        return self

    @staticmethod
    def _overwritten_grads(name):
        """gradients the engine writes with a full overwrite in EVERY backward (mtp_amd.parallel.FlatParams skips clearing them): the four
        Linear weights of each block and the patch embedding -- TN GEMM outputs, 97 % of the buffer.  (Not the FPN weights: a loss that
        ignores a feature map leaves them unwritten, and they must then read zero.)"""
        if name.endswith((".attn.qkv.weight", ".attn.proj.weight", ".mlp.fc1.weight", ".mlp.fc2.weight")) and name.startswith("blocks."):
            return True
        return name == "patch_embed.proj.weight"

    def _engine(self):
        from ..engine import BackboneEngine
        act = torch.bfloat16 if self.precision == "bf16" else torch.float32
        if self._eng is None or self._eng.act != act:
            self._eng = BackboneEngine(self, act)
        return self._eng

    def _feature_dtype(self, x):
        if self.feature_dtype is not None:
            return self.feature_dtype
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    def forward_features(self, x):
        """VIT:787-813: (B,3,H,W) -> [ (B,C,4Hp,4Wp), (B,C,2Hp,2Wp), (B,C,Hp,Wp), (B,C,Hp/2,Wp/2) ]."""
        if not x.is_cuda:
            raise RuntimeError("mtp_amd.ViT_Win_RVSA_V3_WSZ7 runs on an MI355X (gfx950) device only: there is no CPU / eager "
                               "PyTorch fallback.  Move the module and the input to 'cuda'.")
        params = [p for _, p in self.named_parameters()]
        return list(_BackboneFn.apply(self, x, *params))

    def forward(self, x):
        return self.forward_features(x)


def _factory(args, inchannels, **cfg):
    return ViT_Win_RVSA_V3_WSZ7(img_size=args.image_size, in_chans=inchannels, patch_size=16, drop_path_rate=0.1, mlp_ratio=4,
                                qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                                use_checkpoint=(getattr(args, "use_ckpt", "False") == "True"), use_abs_pos_emb=True,
                                use_rel_pos_bias=True, precision=getattr(args, "precision", "bf16"), **cfg)


def vit_b_rvsa(args, inchannels=3):
    """VIT:819-841."""
    return _factory(args, inchannels, out_indices=[3, 5, 7, 11], embed_dim=768, depth=12, num_heads=12, interval=3)


def vit_l_rvsa(args, inchannels=3):
    """VIT:843-865."""
    return _factory(args, inchannels, out_indices=[7, 11, 15, 23], embed_dim=1024, depth=24, num_heads=16, interval=6)


class RVSA_MTP_branches(ViT_Win_RVSA_V3_WSZ7):
    """Fine-tune registry name (RS_Tasks_Finetune/*/backbones/vit_rvsa_mtp_branches.py): the pretrain-style multi-tap
    backbone; returns a tuple and takes `init_weights()` without arguments."""

    def forward(self, x):
        return tuple(self.forward_features(x))

    def init_weights(self, pretrained=None):
        """no-argument form of the fine-tune copies (uses `self.pretrained`, vit_rvsa_mtp.py:684-691) with their loader rules"""
        return self._load_pretrained(pretrained or self.pretrained, finetune=True)


class RVSA_MTP_det(RVSA_MTP_branches):
    """`RVSA_MTP` as registered in mmdet / mmrotate (RS_Tasks_Finetune/Horizontal_Detection/mmdet/models/backbones/
    vit_rvsa_mtp.py:577-844, Rotated_Detection/mmrotate*/...): ViTDet style -- full attention WITHOUT decomposed rel-pos
    (:73-74, 93), no taps: the last block's output goes through the final `norm` (:835) and all four fpn ops are applied to
    that one map (:841).  `out_indices` is accepted and ignored, as there.  Pinned by fixture f9 (generated from that file)."""

    _vitdet = True


class RVSA_MTP_taps(RVSA_MTP_branches):
    """`RVSA_MTP` as registered in mmpretrain / open-cd (RS_Tasks_Finetune/Scene_Classification/mmpretrain/models/backbones/
    vit_rvsa_mtp.py:822-842, Change_Detection/opencd/...): the block outputs at `out_indices` (any number of them) are
    returned as NCHW maps, the fpn ops are commented out there (:838-840); `norm.*` and `fpn*` stay in the state dict and get
    no gradient.  open-cd's extra `frozen_stages` kwarg and `_freeze_stages()` (opencd copy :585, 820-835; never called there,
    :669) are kept with the same meaning.  Pinned by fixture f10 (generated from the mmpretrain file)."""

    _taps_only = True

    def __init__(self, *args, frozen_stages=-1, **kw):
        super().__init__(*args, **kw)
        self.frozen_stages = frozen_stages

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
            self.pos_embed.requires_grad = False
            for i in range(self.frozen_stages):
                self.blocks[i].eval()
                for p in self.blocks[i].parameters():
                    p.requires_grad = False


class RVSA_MTP(RVSA_MTP_branches):
    """`RVSA_MTP` as registered in mmseg (RS_Tasks_Finetune/Semantic_Segmentation/mmseg/models/backbones/vit_rvsa_mtp.py:577):
    multi-level taps + fpn1-4, tuple output.  (The mmdet/mmrotate ViTDet-style last-layer variant is SURVEY 8f-4.)"""


for _cls in (ViT_Win_RVSA_V3_WSZ7, RVSA_MTP, RVSA_MTP_branches, RVSA_MTP_det, RVSA_MTP_taps):
    MODELS.register_module(module=_cls, force=True)
    BACKBONES.register_module(module=_cls, force=True)
