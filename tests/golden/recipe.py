"""Deterministic parameter / input recipes shared by make_golden.py (build container, with the reference)
and the tests (anywhere, without it).  Fixtures store only outputs; inputs are re-created from seeds."""
import zlib

import numpy as np
import torch


# Input seed of fixture f13 (ViT-L, batch 2).  Bilinear sampling is only piecewise differentiable: an input with a sample within f32
# rounding of a cell edge makes every f32 implementation pick one of two one-sided derivatives by its last bit (round 2: seed 2023 had
# one 3.3e-6 px from an edge and moved the gradients upstream of that block by 1e-3).  tests/golden/find_f13_seed.py evaluates the
# reference in float64 for candidate seeds; 2074 is the first whose closest sample is >= 1e-5 px (1.24e-5) from every edge, and
# make_golden.f13 asserts that again.
F13_INPUT_SEED = 2074
F13_MIN_EDGE_DISTANCE = 1e-5
F13_HARD_INPUT_SEED = 2023      # round 2's input: one sample 3.3e-6 px from an edge (fixture f13_vitl_hard.npz, looser gradient bound)


def state_shapes(embed_dim=768, depth=12, heads=12, interval=3, img_size=224, mlp_ratio=4, patch_size=16, layer_scale=False):
    """Reference state-dict keys and shapes (float tensors only), in the reference's order
    (checked against the reference's own state_dict() in make_golden.py -> f0_state_keys.json; patch_size = 8 / layer_scale: fixture f14)."""
    C, hd = embed_dim, embed_dim // heads
    Hp = img_size // patch_size
    s = {"pos_embed": (1, Hp * Hp, C), "patch_embed.proj.weight": (C, 3, patch_size, patch_size), "patch_embed.proj.bias": (C,)}
    for i in range(depth):
        p = "blocks.%d." % i
        window = (i + 1) % interval != 0
        if layer_scale:     # a module's own parameters precede its children's in state_dict(): gamma_1 / gamma_2 (VIT:500-502) come first
            s[p + "gamma_1"] = (C,)
            s[p + "gamma_2"] = (C,)
        s[p + "norm1.weight"] = (C,)
        s[p + "norm1.bias"] = (C,)
        if window:
            s[p + "attn.rel_pos_h"] = (13, hd)
            s[p + "attn.rel_pos_w"] = (13, hd)
            s[p + "attn.relative_position_bias_table"] = (169, heads)
            s[p + "attn.sampling_offsets.2.weight"] = (2 * heads, C, 1, 1)
            s[p + "attn.sampling_offsets.2.bias"] = (2 * heads,)
            s[p + "attn.sampling_scales.2.weight"] = (2 * heads, C, 1, 1)
            s[p + "attn.sampling_scales.2.bias"] = (2 * heads,)
            s[p + "attn.sampling_angles.2.weight"] = (heads, C, 1, 1)
            s[p + "attn.sampling_angles.2.bias"] = (heads,)
        else:
            s[p + "attn.full_attn_rel_pos_h"] = (2 * Hp - 1, hd)
            s[p + "attn.full_attn_rel_pos_w"] = (2 * Hp - 1, hd)
        s[p + "attn.qkv.weight"] = (3 * C, C)
        s[p + "attn.qkv.bias"] = (3 * C,)
        s[p + "attn.proj.weight"] = (C, C)
        s[p + "attn.proj.bias"] = (C,)
        s[p + "norm2.weight"] = (C,)
        s[p + "norm2.bias"] = (C,)
        s[p + "mlp.fc1.weight"] = (mlp_ratio * C, C)
        s[p + "mlp.fc1.bias"] = (mlp_ratio * C,)
        s[p + "mlp.fc2.weight"] = (C, mlp_ratio * C)
        s[p + "mlp.fc2.bias"] = (C,)
    s["norm.weight"] = (C,)
    s["norm.bias"] = (C,)
    s["fpn1.0.weight"] = (C, C, 2, 2)
    s["fpn1.0.bias"] = (C,)
    if patch_size == 8:      # VIT:656-670: fpn1 = one ConvT; fpn2 identity; fpn3 / fpn4 max pools
        return s
    s["fpn1.1.ln.weight"] = (C,)
    s["fpn1.1.ln.bias"] = (C,)
    s["fpn1.3.weight"] = (C, C, 2, 2)
    s["fpn1.3.bias"] = (C,)
    s["fpn2.0.weight"] = (C, C, 2, 2)
    s["fpn2.0.bias"] = (C,)
    return s


def _std_for(name):
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("ln.weight") or name == "norm.weight":
        return None  # 1 + 0.1 N
    if "sampling_" in name:
        return 0.04
    if name.endswith("gamma_1") or name.endswith("gamma_2"):
        return None    # layer scale: 1 + 0.1 N (init_values * ones in the reference; randomised so that its gradient is exercised)
    if "rel_pos" in name or "relative_position_bias_table" in name:
        return 0.1     # zero at init in the reference; randomised so the branch is exercised
    if name.endswith(".bias"):
        return 0.02
    if name == "pos_embed":
        return 0.02
    if name.startswith("fpn") or name.startswith("patch_embed"):
        return 0.03
    return 0.02


def make_params(shapes, seed=2023, dtype=torch.float32):
    """name -> tensor, drawn per-name from a generator seeded by (seed, crc32(name)) so that subsets agree."""
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()) % 1000003)
        t = torch.randn(*shape, generator=g, dtype=torch.float32)
        std = _std_for(name)
        t = 1.0 + 0.1 * t if std is None else std * t
        out[name] = t.to(dtype)
    return out


def make_input(B, H, W, seed=2023):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, H, W, generator=g)


def loss_weights(shape, seed):
    """Fixed pseudo-random cotangent for a feature map (so gradients exercise every element)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) / float(np.prod(shape[1:])) ** 0.5


def sample_indices(numel, n=512, seed=7):
    rng = np.random.RandomState(seed)
    return np.sort(rng.choice(numel, size=min(n, numel), replace=False))


def summarize(t, n=512, seed=7):
    """(sum, abs-sum, sampled values) of a tensor -- compact pin for big outputs."""
    a = t.detach().double().reshape(-1).numpy()
    idx = sample_indices(a.size, n, seed)
    return np.array([a.sum(), np.abs(a).sum()]), a[idx].astype(np.float32)


# ---- InternImage (fixture f12): tiny configuration of the BASELINE config-5 family (16-channel groups, layer scale, post-norm)
II_CFG = dict(channels=32, depths=[1, 1, 2, 1], groups=[2, 4, 8, 16], offset_scale=2.0, layer_scale=0.5)
# f15 (the other InternImageLayer branches): reference constructor keywords on top of II_CFG, and the input seed of each variant -- chosen among 15 .. 59 as
# the one whose DCNv3 samples stay farthest from a bilinear cell edge with the unscaled f12 parameters (4e-5 .. 8e-5 px)
II_VARIANTS = {
    "prenorm_ls": dict(post_norm=False, layer_scale=0.5),                                      # the InternImage-T/S/B family (II:426-427)
    "prenorm": dict(post_norm=False, layer_scale=None),                                        # II:415-417
    "postnorm_nols": dict(post_norm=True, layer_scale=None),                                   # II:409-411
    "respostnorm_l2": dict(post_norm=False, layer_scale=None, res_post_norm=True, level2_post_norm=True, level2_post_norm_block_ids=[0]),   # II:412-414, 512-515 (H/G)
}
II_VARIANTS["hg_full"] = dict(post_norm=False, layer_scale=None, res_post_norm=True, level2_post_norm=True, level2_post_norm_block_ids=[0], dw_kernel_size=5,
                              center_feature_scale=True)                                     # every InternImage-H/G switch together (DCNM:124, 168-173, 209-215)
II_VARIANTS["postnorm_cfs"] = dict(post_norm=True, layer_scale=0.5, center_feature_scale=True, dw_kernel_size=7)      # the level norm that center_feature_scale adds to post-norm models (II:497, 516)
II_VARIANT_SEEDS = {"prenorm_ls": 26, "prenorm": 33, "postnorm_nols": 32, "respostnorm_l2": 50, "hg_full": 41, "postnorm_cfs": 42}


def internimage_variant_params(shapes, seed=77):
    """f15's parameters: f12's recipe with the offset heads at a quarter of its scale (offsets of +-3 px instead of +-12).  With LayerNorms right behind the
    branches and no layer scale < 1 (postnorm_nols, respostnorm_l2) the unscaled recipe is ill-conditioned on the 4 x 4 and 2 x 2 maps -- the oracle's own float32
    run is 2e-3 away from its float64 run in single gradients, 1e-5 with this scale -- and a fixture that cannot tell 2e-3 from right is no check"""
    out = {k: (0.25 * v if ".dcn.offset." in k else v) for k, v in internimage_params(shapes, seed).items()}
    for k in out:      # the gate's Linear (zero-initialised in the reference): scaled like the mask head so that the gates really differ
        if k.endswith("center_feature_scale_proj_weight"):
            out[k] = 0.2 * torch.randn(out[k].shape, generator=torch.Generator().manual_seed(seed * 7919 + zlib.crc32(k.encode()) % 7919))
    return out


def internimage_params(shapes, seed=77):
    """seeded parameters for f12: the reference zero-initialises the offset / mask heads (dcnv3.py:176-179) -- randomised so the
    sampling really deforms; LayerNorm weights and layer-scale gammas around their init values"""
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()) % 1000003)
        t = torch.randn(*shape, generator=g)
        if name.endswith("gamma1") or name.endswith("gamma2"):
            t = 0.5 + 0.1 * t
        elif ("norm" in name or "dw_conv.1.1" in name) and name.endswith(".weight"):
            t = 1.0 + 0.1 * t
        elif "offset.weight" in name:
            t = 0.3 * t
        elif "mask.weight" in name:
            t = 0.2 * t
        elif name.endswith(".bias"):
            t = 0.05 * t
        elif "conv" in name:
            t = 0.15 * t
        else:
            t = t / (shape[-1] ** 0.5)
        out[name] = t
    return out
