"""CPU: the InternImage oracle (oracle/internimage_oracle.py) against fixture f12 = the reference's own
InternImage(core_op='DCNv3_pytorch') on the seeded parameters of tests/golden/recipe.py (make_golden.py f12)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import internimage_oracle as IO

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import recipe  # noqa: E402

FIX = np.load(os.path.join(ROOT, "tests", "golden", "f12_internimage.npz"))
FIX15 = np.load(os.path.join(ROOT, "tests", "golden", "f15_internimage_variants.npz"))
CFG = recipe.II_CFG
# fixture f15 (make_golden.py F15_VARIANTS): the reference's other InternImageLayer branches (II:407-427) and block norms (II:497-517)
VARIANTS = recipe.II_VARIANTS


def variant_shapes(kw):
    return IO.state_shapes(CFG["channels"], CFG["depths"], CFG["groups"], post_norm=kw["post_norm"], layer_scale=kw["layer_scale"] is not None,
                           res_post_norm=kw.get("res_post_norm", False), level2_post_norm_block_ids=kw.get("level2_post_norm_block_ids"),
                           dw_kernel_size=kw.get("dw_kernel_size"), center_feature_scale=kw.get("center_feature_scale", False))


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def test_state_dict_keys_shapes_and_order_match_the_reference():
    shapes = IO.state_shapes(CFG["channels"], CFG["depths"], CFG["groups"])
    assert list(shapes.keys()) == [str(k) for k in FIX["keys"]]
    assert [str(tuple(v)) for v in shapes.values()] == [str(s) for s in FIX["shapes"]]
    xl = IO.state_shapes()                                  # InternImage-XL, BASELINE config 5: 39 DCNv3 layers
    assert sum(k.endswith("dcn.offset.weight") for k in xl) == 39 and xl["levels.3.blocks.4.mlp.fc1.weight"] == (6144, 1536)
    assert sum(int(np.prod(v)) for v in xl.values()) == 328_971_228      # the parameter count SURVEY 8c reports for the reference


@pytest.mark.parametrize("dtype,tol_f,tol_g", [(torch.float64, 2e-5, 2e-4), (torch.float32, 3e-4, 2e-3)])
def test_forward_and_gradients_vs_reference(dtype, tol_f, tol_g):
    """the fixture is the reference run in float64; its DCNv3 core still builds the sampling grid in float32
    (dcnv3_func.py:118-158), a ~1e-7 location perturbation that four levels of 3x3 deformable sampling on maps down to 2x2
    amplify to ~1e-5 -- that, not the oracle, sets the float64 tolerance (tests/test_dcnv3_oracle.py isolates the effect)"""
    shapes = IO.state_shapes(CFG["channels"], CFG["depths"], CFG["groups"])
    p = {k: v.to(dtype).requires_grad_(True) for k, v in recipe.internimage_params(shapes).items()}
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(12)).to(dtype).requires_grad_(True)
    feats = IO.backbone_forward(img, p, CFG["depths"], CFG["groups"], CFG["offset_scale"])
    assert [tuple(f.shape) for f in feats] == [(2, 32, 16, 16), (2, 64, 8, 8), (2, 128, 4, 4), (2, 256, 2, 2)]
    for i, f in enumerate(feats):
        assert rel(f.detach().double(), torch.from_numpy(FIX["feat%d" % i])) < tol_f, i
    gs = [torch.randn(f.shape, generator=torch.Generator().manual_seed(100 + i)).to(dtype) for i, f in enumerate(feats)]
    sum((f * g).sum() for f, g in zip(feats, gs)).backward()
    assert rel(img.grad.double(), torch.from_numpy(FIX["grad_img"])) < tol_g
    for k in FIX.files:
        if k.startswith("grad."):
            assert rel(p[k[5:]].grad.double(), torch.from_numpy(FIX[k])) < tol_g, k


def test_hip_backbone_class_has_the_reference_state_dict_and_init_rules():
    """mtp_amd.InternImage (the HIP-side class, constructed on CPU): state-dict keys / shapes / order of the reference (fixture f12),
    the XL factory's parameter count, the init rules of II:672-686 + DCNv3._reset_parameters, and loud refusal of what is not built"""
    import mtp_amd
    net = mtp_amd.InternImage(channels=CFG["channels"], depths=CFG["depths"], groups=CFG["groups"], layer_scale=CFG["layer_scale"],
                              offset_scale=CFG["offset_scale"], post_norm=True)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in FIX["keys"]] and [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in FIX["shapes"]]
    assert float(sd["levels.0.blocks.0.dcn.offset.weight"].abs().max()) == 0 and float(sd["levels.1.blocks.0.dcn.mask.bias"].abs().max()) == 0
    assert torch.equal(sd["levels.2.blocks.1.gamma2"], torch.full((128,), CFG["layer_scale"])) and torch.equal(sd["levels.0.blocks.0.norm1.0.weight"], torch.ones(32))
    assert float(sd["levels.0.blocks.0.mlp.fc1.weight"].abs().max()) <= 2.0 and 0.01 < float(sd["levels.0.blocks.0.mlp.fc1.weight"].std()) < 0.03
    assert len(net.drop_path_rates) == sum(CFG["depths"]) and net.drop_path_rates[0] == 0.0 and abs(net.drop_path_rates[-1] - 0.2) < 1e-6
    assert mtp_amd.MODELS.get("InternImage") is mtp_amd.InternImage
    with pytest.raises(ValueError):
        mtp_amd.InternImage(layer_scale=1.0, post_norm=True, dw_kernel_size=4)       # even depth-wise kernels have no "same" padding (DCNM:150)
    with pytest.raises(ValueError):
        mtp_amd.InternImage(layer_scale=1.0, post_norm=False, res_post_norm=True)      # (the reference would silently ignore res_post_norm here, II:408-427)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))          # no CPU path


def test_flat_gradient_layout_follows_the_backward_order_of_the_levels():
    """mtp_amd.parallel.FlatParams over InternImage: parameters in reverse execution order (last level first, a level's downsample
    with its last layer, the stem last), so that what the backward has finished is always a prefix of the flat gradient buffer"""
    import mtp_amd
    from mtp_amd.parallel import FlatParams
    net = mtp_amd.InternImage(channels=CFG["channels"], depths=CFG["depths"], groups=CFG["groups"], layer_scale=CFG["layer_scale"],
                              offset_scale=CFG["offset_scale"], post_norm=True)
    flat = FlatParams(net, unused=net._unused_params)
    gids = [flat.groups[n] for n in flat.names]
    assert gids == sorted(gids, reverse=True) and gids[0] == sum(CFG["depths"]) - 1 and gids[-1] == -1
    assert flat.groups["levels.1.downsample.conv.weight"] == flat.groups["levels.1.blocks.0.mlp.fc1.weight"] == 1
    assert flat.names[-1].startswith("patch_embed") and flat.reduced == flat.total
    b = flat.buckets(1 << 12)
    assert b[0][1] == 0 and b[-1][2] == flat.total and all(x[2] == y[1] for x, y in zip(b, b[1:]))
    # parameters now live in the flat buffer (views), values unchanged
    assert net.state_dict()["levels.0.blocks.0.gamma1"].data_ptr() == flat.view(flat.data, "levels.0.blocks.0.gamma1").data_ptr()


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_layer_variants_oracle_and_class_surface_vs_reference(name):
    """fixture f15 = the reference's InternImage(core_op='DCNv3_pytorch') built with the other layer branches: pre-norm with / without layer scale
    (InternImage-T/S/B), post-norm without layer scale, res_post_norm + level-2 post norms (the H/G layer form).  The oracle reproduces features and
    gradients; mtp_amd.InternImage has the same state-dict keys / shapes / order and a flat-gradient order that follows its backward."""
    import mtp_amd
    from mtp_amd.parallel import FlatParams
    kw = VARIANTS[name]
    shapes = variant_shapes(kw)
    assert list(shapes.keys()) == [str(k) for k in FIX15[name + ".keys"]]
    assert [str(tuple(v)) for v in shapes.values()] == [str(s) for s in FIX15[name + ".shapes"]]
    p = {k: v.double().requires_grad_(True) for k, v in recipe.internimage_variant_params(shapes).items()}
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(recipe.II_VARIANT_SEEDS[name])).double().requires_grad_(True)
    feats = IO.backbone_forward(img, p, CFG["depths"], CFG["groups"], CFG["offset_scale"], post_norm=kw["post_norm"],
                                level2_post_norm_block_ids=kw.get("level2_post_norm_block_ids"))
    tol_f, tol_g = 2e-5, 2e-4      # (as for f12: what is left is the float32 sampling grid of the reference's dcnv3_core_pytorch)
    for i, f in enumerate(feats):
        assert rel(f.detach(), torch.from_numpy(FIX15["%s.feat%d" % (name, i)]).double()) < tol_f, i
    gs = [torch.randn(f.shape, generator=torch.Generator().manual_seed(200 + i)).double() for i, f in enumerate(feats)]
    sum((f * g).sum() for f, g in zip(feats, gs)).backward()
    assert rel(img.grad, torch.from_numpy(FIX15[name + ".grad_img"]).double()) < tol_g
    n = 0
    for k in FIX15.files:
        if k.startswith(name + ".grad."):
            assert rel(p[k[len(name) + 6:]].grad, torch.from_numpy(FIX15[k]).double()) < tol_g, k
            n += 1
    assert n >= 9
    net = mtp_amd.InternImage(channels=CFG["channels"], depths=CFG["depths"], groups=CFG["groups"], offset_scale=CFG["offset_scale"], **kw)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in FIX15[name + ".keys"]] and [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in FIX15[name + ".shapes"]]
    flat = FlatParams(net, unused=net._unused_params)
    gids = [flat.groups[m] for m in flat.names]
    assert gids == sorted(gids, reverse=True) and gids[-1] == -1
    if not kw["post_norm"] or kw.get("center_feature_scale"):
        assert flat.groups["levels.1.norm.0.weight"] == flat.groups["levels.1.blocks.0.mlp.fc1.weight"]
    if kw.get("level2_post_norm"):
        assert flat.groups["levels.2.post_norms.0.0.weight"] == flat.groups["levels.2.blocks.0.mlp.fc1.weight"]
