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
CFG = recipe.II_CFG


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
