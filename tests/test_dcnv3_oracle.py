"""CPU: the DCNv3 oracle (oracle/dcnv3_oracle.py) against fixture f11, generated from the reference's own pure-torch core
`dcnv3_core_pytorch` with the input recipe of the reference's test (ops_dcnv3/test.py) -- tests/golden/make_golden.py f11."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import dcnv3_oracle as D

FIX = np.load(os.path.join(ROOT, "tests", "golden", "f11_dcnv3.npz"))
CASES = [str(c) for c in FIX["cases"]]


def load_case(name, dtype=torch.float64):
    N, H, W, M, Dg, kh, kw, st, pad, dil, rmc = [int(v) for v in FIX[name + ".cfg"]]
    t = {k: torch.from_numpy(FIX[name + "." + k]).to(dtype) for k in ("input", "offset", "mask", "grad_output", "output", "grad_input", "grad_offset", "grad_mask")}
    args = (kh, kw, st, st, pad, pad, dil, dil, M, Dg, float(FIX[name + ".offset_scale"]))
    return t, args, rmc


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_fp64(name):
    # dcnv3_core_pytorch builds its reference points / dilation grid with float32 linspace and divides by the padded size
    # (dcnv3_func.py:118-158) before the double offsets are added, so even its float64 run carries ~1e-7 of location rounding;
    # the native kernel (and this oracle) use exact integer arithmetic for that part.  Where the division is exact in
    # float32 ("nopad": 8x8 map) the two agree to 1e-15.
    tol = 1e-12 if name == "nopad" else 2e-6
    t, args, rmc = load_case(name)
    y = D.dcnv3_forward(t["input"], t["offset"], t["mask"], *args, rmc)
    assert y.shape == t["output"].shape and rel(y, t["output"]) < tol
    gi, go, gm = D.dcnv3_backward(t["input"], t["offset"], t["mask"], *args, t["grad_output"], rmc)
    assert rel(gi, t["grad_input"]) < tol and rel(go, t["grad_offset"]) < tol and rel(gm, t["grad_mask"]) < tol


@pytest.mark.parametrize("name", ["base", "stride2", "rmc"])
def test_fp32(name):
    t, args, rmc = load_case(name, torch.float32)
    y = D.dcnv3_forward(t["input"], t["offset"], t["mask"], *args, rmc)
    gi, go, gm = D.dcnv3_backward(t["input"], t["offset"], t["mask"], *args, t["grad_output"], rmc)
    assert rel(y, t["output"]) < 1e-5 and rel(gi, t["grad_input"]) < 1e-5 and rel(go, t["grad_offset"]) < 1e-4 and rel(gm, t["grad_mask"]) < 1e-5


def test_fixture_exercises_the_border_rules():
    """the recipe must hit every branch: points outside the map, valid points with out-of-range corners, interior points"""
    t, args, rmc = load_case("dil2")
    kh, kw, sh, sw, ph, pw, dh, dw, M, Dg, osc = args
    N, H, W, _ = t["input"].shape
    lh, lw = D._locations(t["offset"], H, W, kh, kw, sh, sw, ph, pw, dh, dw, M, osc, rmc)
    valid, corners, _ = D._corners(lh, lw, H, W)
    assert 0.02 < (~valid).double().mean() < 0.9
    partial = valid & ~(corners[0][3] & corners[1][3] & corners[2][3] & corners[3][3])
    assert partial.any() and (valid & ~partial).any()


def test_backward_is_the_adjoint_of_forward():
    """size-independent property: <dcnv3(x), G> is linear in x and in mask, so the gradients must reproduce it"""
    t, args, rmc = load_case("base")
    y = D.dcnv3_forward(t["input"], t["offset"], t["mask"], *args, rmc)
    gi, _, gm = D.dcnv3_backward(t["input"], t["offset"], t["mask"], *args, t["grad_output"], rmc)
    s = (y * t["grad_output"]).sum()
    assert abs(((gi * t["input"]).sum() - s) / s) < 1e-10 and abs(((gm * t["mask"]).sum() - s) / s) < 1e-10
