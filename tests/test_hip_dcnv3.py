"""GPU: the DCNv3 core operator (mtp_dcnv3_fwd / mtp_dcnv3_bwd through mtp_amd.ops_dcnv3) against the reference-generated
fixture f11 and the oracle, laid out like the reference's own test (ops_dcnv3/test.py: forward float, backward float for
several channel counts), plus bf16, the autograd Function and InternImage-sized properties."""
import pytest
import torch

from test_dcnv3_oracle import CASES, load_case, rel

pytestmark = pytest.mark.gpu


def dev(t, dtype=torch.float32):
    return t.to(dtype).cuda().contiguous()


@pytest.mark.parametrize("name", CASES)
def test_forward_and_backward_fp32_vs_reference_fixture(name):
    """test.py asserts rtol 1e-2 / atol 1e-3 for its float kernels; the bar here is 1e-5 of the tensor maximum (the fixture itself
    carries ~1e-6 from the float32 grid inside dcnv3_core_pytorch, tests/test_dcnv3_oracle.py)"""
    from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
    t, args, rmc = load_case(name)
    x, off, m, G = dev(t["input"]), dev(t["offset"]), dev(t["mask"]), dev(t["grad_output"])
    y = dcnv3_forward(x, off, m, *args, 256, rmc)
    assert y.dtype == torch.float32 and rel(y.double().cpu(), t["output"]) < 1e-5
    gi, go, gm = dcnv3_backward(x, off, m, *args, G, 256, rmc)
    assert rel(gi.double().cpu(), t["grad_input"]) < 1e-5
    assert rel(go.double().cpu(), t["grad_offset"]) < 2e-5
    assert rel(gm.double().cpu(), t["grad_mask"]) < 1e-5


@pytest.mark.parametrize("name", ["base", "odd", "stride2", "rmc", "wide"])
def test_bf16_vs_oracle_on_the_same_rounded_inputs(name):
    from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
    from oracle import dcnv3_oracle as D
    t, args, rmc = load_case(name)
    r = {k: t[k].to(torch.bfloat16) for k in ("input", "offset", "mask", "grad_output")}
    ref_y = D.dcnv3_forward(r["input"].double(), r["offset"].double(), r["mask"].double(), *args, rmc)
    ref_g = D.dcnv3_backward(r["input"].double(), r["offset"].double(), r["mask"].double(), *args, r["grad_output"].double(), rmc)
    x, off, m, G = (r[k].cuda().contiguous() for k in ("input", "offset", "mask", "grad_output"))
    y = dcnv3_forward(x, off, m, *args, 256, rmc)
    assert y.dtype == torch.bfloat16 and rel(y.double().cpu(), ref_y) < 6e-3          # one bf16 rounding of the output
    grads = dcnv3_backward(x, off, m, *args, G, 256, rmc)
    for g, rg in zip(grads, ref_g):
        assert g.dtype == torch.float32 and rel(g.double().cpu(), rg) < 2e-5          # f32 math on identical inputs


@pytest.mark.parametrize("name", ["base", "rmc"])
def test_float16_runs_on_the_f32_kernels_and_float64_on_double_kernels(name):
    """the reference dispatches float / double / half (dcnv3_cuda.cu:69): float16 operands are accepted -- float32 arithmetic on the same rounded inputs, the
    output back in float16, float32 gradients as for the reference's promoted half -- and through DCNv3Function with half leaves; float64 runs on double kernels
    (MTP_F64): output and the three gradients equal the float64 oracle to 1e-12"""
    from mtp_amd.ops_dcnv3 import DCNv3Function, dcnv3_backward, dcnv3_forward
    from oracle import dcnv3_oracle as D
    t, args, rmc = load_case(name)
    r = {k: t[k].to(torch.float16) for k in ("input", "offset", "mask", "grad_output")}
    ref_y = D.dcnv3_forward(r["input"].double(), r["offset"].double(), r["mask"].double(), *args, rmc)
    ref_g = D.dcnv3_backward(r["input"].double(), r["offset"].double(), r["mask"].double(), *args, r["grad_output"].double(), rmc)
    x, off, m, G = (r[k].cuda().contiguous() for k in ("input", "offset", "mask", "grad_output"))
    y = dcnv3_forward(x, off, m, *args, 256, rmc)
    assert y.dtype == torch.float16 and rel(y.double().cpu(), ref_y) < 1e-3          # one float16 rounding of the output
    for g, rg in zip(dcnv3_backward(x, off, m, *args, G, 256, rmc), ref_g):
        assert g.dtype == torch.float32 and rel(g.double().cpu(), rg) < 2e-5
    xa, oa, ma = x.clone().requires_grad_(True), off.clone().requires_grad_(True), m.clone().requires_grad_(True)
    DCNv3Function.apply(xa, oa, ma, *args, 256, rmc).backward(G)
    assert xa.grad.dtype == torch.float16 and rel(xa.grad.double().cpu(), ref_g[0]) < 2e-3
    d = {k: t[k].double() for k in ("input", "offset", "mask", "grad_output")}
    ref_y = D.dcnv3_forward(d["input"], d["offset"], d["mask"], *args, rmc)
    ref_g = D.dcnv3_backward(d["input"], d["offset"], d["mask"], *args, d["grad_output"], rmc)
    xd, od, md, Gd = (d[k].cuda().contiguous() for k in ("input", "offset", "mask", "grad_output"))
    yd = dcnv3_forward(xd, od, md, *args, 256, rmc)
    assert yd.dtype == torch.float64 and rel(yd.cpu(), ref_y) < 1e-12
    for g, rg in zip(dcnv3_backward(xd, od, md, *args, Gd, 256, rmc), ref_g):
        assert g.dtype == torch.float64 and rel(g.cpu(), rg) < 1e-12


def test_double_gradients_pass_torch_gradcheck():
    """what the reference's own test-suite does with its double dispatch (ops_dcnv3/test.py: check_gradient_numerical): torch.autograd.gradcheck of
    DCNv3Function in float64 -- numerical against analytical gradients of input, offset and mask (offsets kept away from the bilinear cell edges, where the
    operator is not differentiable)"""
    from mtp_amd.ops_dcnv3 import DCNv3Function
    g = torch.Generator().manual_seed(3)
    N, H, W, G, GC, P = 1, 4, 5, 2, 4, 9
    x = (0.5 * torch.randn(N, H, W, G * GC, generator=g)).double().cuda().requires_grad_(True)
    off = (0.25 + 0.2 * torch.rand(N, H, W, G * P * 2, generator=g)).double().cuda().requires_grad_(True)      # locations at integer + 0.25 .. 0.45: clear of the edges
    m = torch.softmax(torch.randn(N, H, W, G, P, generator=g), -1).reshape(N, H, W, G * P).double().cuda().requires_grad_(True)

    def f(a, b, c):
        return DCNv3Function.apply(a, b, c, 3, 3, 1, 1, 1, 1, 1, 1, G, GC, 1.0, 256, 0)
    assert torch.autograd.gradcheck(f, (x, off, m), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)


def test_autograd_function_matches_the_extension_calls():
    from mtp_amd.ops_dcnv3 import DCNv3Function, dcnv3_backward
    t, args, rmc = load_case("base")
    x, off, m, G = dev(t["input"]).requires_grad_(True), dev(t["offset"]).requires_grad_(True), dev(t["mask"]).requires_grad_(True), dev(t["grad_output"])
    y = DCNv3Function.apply(x, off, m, *args, 2, rmc)
    y.backward(G)
    gi, go, gm = dcnv3_backward(x.detach(), off.detach(), m.detach(), *args, G, 2, rmc)
    # the scatter uses f32 atomics: run-to-run differences are rounding-order only
    assert rel(x.grad, gi) < 1e-6 and torch.equal(off.grad, go) and torch.equal(m.grad, gm)
    assert rel(x.grad.double().cpu(), t["grad_input"]) < 1e-5


def test_internimage_sized_level_properties():
    """InternImage-XL level 2 at 512^2 (64 x 64 map, 24 groups x 16 channels, 3x3, offset_scale 2; config cited in SURVEY 8f-3):
    <dcnv3(x), G> is linear in x and in mask, so the gradients must reproduce it; identity sampling (zero offsets, one-hot
    centre mask) must return the input bit-for-bit"""
    from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
    torch.manual_seed(5)
    N, H, W, M, Dg, P = 2, 64, 64, 24, 16, 9
    args = (3, 3, 1, 1, 1, 1, 1, 1, M, Dg, 2.0)
    x = torch.randn(N, H, W, M * Dg, device="cuda")
    off = (torch.rand(N, H, W, M * P * 2, device="cuda") - 0.5) * 6
    m = torch.softmax(torch.randn(N, H, W, M, P, device="cuda"), -1).reshape(N, H, W, M * P)
    G = torch.randn(N, H, W, M * Dg, device="cuda")
    y = dcnv3_forward(x, off, m, *args, 256, 0)
    gi, go, gm = dcnv3_backward(x, off, m, *args, G, 256, 0)
    s = (y.double() * G.double()).sum()
    assert abs(((gi.double() * x.double()).sum() - s) / s) < 1e-5
    assert abs(((gm.double() * m.double()).sum() - s) / s) < 1e-5
    assert torch.isfinite(go).all() and go.abs().max() > 0
    for dt in (torch.float32, torch.bfloat16):
        ident = torch.zeros(N, H, W, M, P, device="cuda", dtype=dt)
        ident[..., 4] = 1
        yi = dcnv3_forward(x.to(dt), torch.zeros_like(off, dtype=dt), ident.reshape(N, H, W, M * P), *args, 256, 0)
        assert torch.equal(yi, x.to(dt))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,osc,amp,rmc", [((2, 32, 32, 12), 2.0, 0.0, 0), ((2, 32, 32, 12), 2.0, 0.6, 0), ((1, 37, 21, 5), 2.0, 3.0, 0), ((2, 16, 16, 6), 1.0, 8.0, 0),
                                               ((1, 20, 13, 3), 1.0, 1.5, 1), ((3, 7, 5, 2), 2.0, 1.0, 0), ((1, 50, 50, 4), 0.5, 2.0, 0)])
def test_gather_form_backward_equals_the_scatter_form(shape, osc, amp, rmc, dtype, monkeypatch):
    """the default backward on InternImage's geometry (3x3, stride 1, pad 1, 16-channel groups: grad_input summed per input pixel from the
    output pixels around it + a wave-cooperative atomic path for samples beyond its reach) against the per-corner atomic scatter
    (MTP_DCNV3_VARIANT=2, the kernel the fixture tests pinned in rounds 1-2): zero offsets (all samples on the fast path), small, and
    large offsets (mostly the atomic path), map sizes that are not multiples of the 16 x 16 tile, remove_center, three offset scales"""
    from mtp_amd.ops_dcnv3 import dcnv3_backward
    torch.manual_seed(11)
    N, H, W, M = shape
    P = 9 - rmc
    args = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, osc)
    x = torch.randn(N, H, W, M * 16, device="cuda").to(dtype)
    off = ((torch.rand(N, H, W, M * P * 2, device="cuda") - 0.5) * 2 * amp).to(dtype)
    m = torch.softmax(torch.randn(N, H, W, M, P, device="cuda"), -1).reshape(N, H, W, M * P).to(dtype)
    G = torch.randn(N, H, W, M * 16, device="cuda").to(dtype)
    monkeypatch.setenv("MTP_DCNV3_VARIANT", "2")
    ref = dcnv3_backward(x, off, m, *args, G, 256, rmc)
    for variant in ("0", "4"):       # 0: the window form; 4: the 3 x 3 form where offset_scale is 1 or 2 (else the window form)
        monkeypatch.setenv("MTP_DCNV3_VARIANT", variant)
        got = dcnv3_backward(x, off, m, *args, G, 256, rmc)
        for a, b, name in zip(got, ref, ("grad_input", "grad_offset", "grad_mask")):
            assert torch.isfinite(a).all() and rel(a, b) < 2e-6, (variant, name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,rmc,ld", [(12, 0, 216), (12, 0, 224), (6, 1, 104)])
def test_backward_writes_grad_offset_as_gemm_operand(dtype, M, rmc, ld):
    """mtp_dcnv3_bwd_act: grad_offset once more in the input dtype as rows of `ld` elements, pad columns zero -- equal to casting the f32
    grad_offset (what the engine did in a separate pass before round 4); the three f32 gradients unchanged.  A geometry without a gather-form
    backward (stride 2) answers None for the copy and still returns the gradients."""
    from mtp_amd.ops_dcnv3.functions import dcnv3_backward, dcnv3_backward_act
    torch.manual_seed(3)
    N, H, W = 2, 19, 23
    P = 9 - rmc
    args = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, 2.0)
    x = torch.randn(N, H, W, M * 16, device="cuda").to(dtype)
    off = ((torch.rand(N, H, W, M * P * 2, device="cuda") - 0.5) * 3).to(dtype)
    m = torch.softmax(torch.randn(N, H, W, M, P, device="cuda"), -1).reshape(N, H, W, M * P).to(dtype)
    G = torch.randn(N, H, W, M * 16, device="cuda").to(dtype)
    ref = dcnv3_backward(x, off, m, *args, G, 256, rmc)
    gi, go, gm, act = dcnv3_backward_act(x, off, m, *args, G, 256, ld, rmc)
    assert rel(gi, ref[0]) < 2e-6 and torch.equal(go, ref[1]) and torch.equal(gm, ref[2])
    assert act is not None and tuple(act.shape) == (N * H * W, ld) and act.dtype == dtype
    n = M * P * 2
    assert torch.equal(act[:, :n], ref[1].view(-1, n).to(dtype)) and float(act[:, n:].abs().max() if ld > n else 0.0) == 0.0
    args2 = (3, 3, 2, 2, 1, 1, 1, 1, M, 16, 2.0)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    off2, m2, G2 = off[:, :Ho, :Wo].contiguous(), m[:, :Ho, :Wo].contiguous(), G[:, :Ho, :Wo].contiguous()
    out = dcnv3_backward_act(x, off2, m2, *args2, G2, 256, ld, rmc)
    assert out[3] is None and all(torch.isfinite(t).all() for t in out[:3])


def test_bf16_matches_fp32_at_internimage_size():
    from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
    torch.manual_seed(6)
    N, H, W, M, Dg, P = 1, 32, 32, 48, 16, 9
    args = (3, 3, 1, 1, 1, 1, 1, 1, M, Dg, 2.0)
    x = torch.randn(N, H, W, M * Dg, device="cuda").bfloat16()
    off = ((torch.rand(N, H, W, M * P * 2, device="cuda") - 0.5) * 4).bfloat16()
    m = torch.softmax(torch.randn(N, H, W, M, P, device="cuda"), -1).reshape(N, H, W, M * P).bfloat16()
    G = torch.randn(N, H, W, M * Dg, device="cuda").bfloat16()
    y16 = dcnv3_forward(x, off, m, *args, 256, 0)
    y32 = dcnv3_forward(x.float(), off.float(), m.float(), *args, 256, 0)
    assert rel(y16.float(), y32) < 6e-3
    for a, b in zip(dcnv3_backward(x, off, m, *args, G, 256, 0), dcnv3_backward(x.float(), off.float(), m.float(), *args, G.float(), 256, 0)):
        assert rel(a, b) < 1e-5


def test_reference_error_behaviour_on_device_tensors():
    from mtp_amd.ops_dcnv3 import dcnv3_forward
    x, off, m = torch.zeros(3, 8, 8, 64, device="cuda"), torch.zeros(3, 8, 8, 72, device="cuda"), torch.zeros(3, 8, 8, 36, device="cuda")
    args = (3, 3, 1, 1, 1, 1, 1, 1, 4, 16, 1.0)
    with pytest.raises(RuntimeError, match="contiguous"):
        dcnv3_forward(x.permute(0, 2, 1, 3), off, m, *args, 256, 0)
    with pytest.raises(RuntimeError, match="im2col_step"):
        dcnv3_forward(x, off, m, *args, 2, 0)                       # 3 % min(3, 2) != 0   (dcnv3_cuda.cu:46-49)
    with pytest.raises(RuntimeError, match="wont match"):
        dcnv3_forward(x, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 1.0, 256, 0)
    with pytest.raises(RuntimeError):
        dcnv3_forward(x, off, m, 4, 4, 1, 1, 1, 1, 1, 1, 4, 16, 1.0, 256, 1)   # remove_center needs a square odd kernel
    assert dcnv3_forward(x, off, m, *args, 3, 0).shape == (3, 8, 8, 64)


def test_offset_gradient_differences_are_confined_to_samples_on_a_cell_edge():
    """VERDICT r04 weak #1 / next #4d: on the data-dependent InternImage-XL recipe `levels.2.blocks.7.dcn.offset.weight` is 2.8e-2 off in fp32 mode, and the
    explanation -- bilinear sampling is only piecewise differentiable, a sample within f32 rounding of a cell edge takes the other one-sided derivative -- was
    an argument.  Here it is a test, at the operator, on a level-sized problem (64 x 64 map, 24 groups x 16 channels, 3 x 3, offset_scale 2: level 1 of XL at
    512^2) with offsets of the benchmark's spread and ~40 samples planted within 2e-6 px of an edge (an f32 ulp at these coordinates is 4e-6 ... 8e-6 px):
      (1) the offset gradient of every sample farther than 1e-4 px from an edge agrees with the float64 oracle to 1e-5 of the tensor's maximum;
      (2) whatever difference remains sits in the near-edge samples (recorded in the parity table: the oracle itself evaluated in f32 differs by 2.7e-2 there);
      (3) with the near-edge samples taken out of BOTH sides -- their modulation mask set to zero, which removes their contribution to the output and to
          every gradient -- output, grad_input, grad_offset and grad_mask all agree to 1e-5."""
    from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
    from oracle import dcnv3_oracle as D
    g = torch.Generator().manual_seed(17)
    N, H, W, G, gc, P = 1, 64, 64, 24, 16, 9
    args = (3, 3, 1, 1, 1, 1, 1, 1, G, gc, 2.0)
    x = torch.randn(N, H, W, G * gc, generator=g)
    off = 0.4 * torch.randn(N, H, W, G * P * 2, generator=g)
    mask = torch.softmax(torch.randn(N, H, W, G, P, generator=g), -1).reshape(N, H, W, G * P)
    dy = torch.randn(N, H, W, G * gc, generator=g)
    # plant samples on (next to) cell edges: move the x offset of every 20000th sample so that its location is an integer + 2e-6
    loc_h, loc_w = D._locations(off.double(), H, W, *args[:8], G, 2.0, 0)
    flat = off.reshape(-1, 2).clone()
    lw = loc_w.reshape(-1)
    idx = torch.arange(0, lw.numel(), 20000)
    sign = torch.where(torch.arange(idx.numel()) % 2 == 0, 1.0, -1.0).double()           # alternately just right / just left of the edge
    flat[idx, 0] += ((lw[idx].round() + 2e-6 * sign - lw[idx]) / 2.0).float()
    off = flat.reshape(off.shape)
    loc_h, loc_w = D._locations(off.double(), H, W, *args[:8], G, 2.0, 0)
    dist = torch.minimum((loc_h - loc_h.round()).abs(), (loc_w - loc_w.round()).abs())      # (N, Ho, Wo, G, P)
    near = dist < 1e-4
    assert int(near.sum()) >= idx.numel() and int(near.sum()) < 2000
    ref = D.dcnv3_backward(x.double(), off.double(), mask.double(), *args, dy.double(), 0)
    gi, go, gm = dcnv3_backward(dev(x), dev(off), dev(mask), *args, dev(dy), 256, 0)
    go_ref = ref[1].reshape(N, H, W, G, P, 2)
    go_hip = go.double().cpu().reshape(N, H, W, G, P, 2)
    scale = go_ref.abs().max()
    far_err = ((go_hip - go_ref).abs() * (~near).unsqueeze(-1)).max() / scale
    near_err = ((go_hip - go_ref).abs() * near.unsqueeze(-1)).max() / scale
    assert far_err < 1e-5, float(far_err)                       # (1)
    assert rel(gi.double().cpu(), ref[0]) < 1e-5 and rel(gm.double().cpu(), ref[2]) < 1e-5       # value paths are continuous across an edge
    print("offset-gradient difference: far samples %.2e, near-edge samples %.2e of the maximum (%d near-edge samples)" % (float(far_err), float(near_err), int(near.sum())))
    from conftest import record_parity
    record_parity("dcnv3_offset_gradient_kinks", "far_samples", float(far_err))       # (2) what difference there is, is theirs: the f32 location of a sample planted
    record_parity("dcnv3_offset_gradient_kinks", "near_edge_samples", float(near_err))  #     2e-6 px from an edge may round across it (the oracle evaluated in f32 shows 2.7e-2 here)
    m2 = (mask.reshape(N, H, W, G, P) * (~near)).reshape(mask.shape)
    ref2 = D.dcnv3_backward(x.double(), off.double(), m2.double(), *args, dy.double(), 0)
    y2 = dcnv3_forward(dev(x), dev(off), dev(m2), *args, 256, 0)
    assert rel(y2.double().cpu(), D.dcnv3_forward(x.double(), off.double(), m2.double(), *args, 0)) < 1e-5
    for a, b, nm in zip(dcnv3_backward(dev(x), dev(off), dev(m2), *args, dev(dy), 256, 0), ref2, ("grad_input", "grad_offset", "grad_mask")):
        if nm == "grad_mask":     # the mask gradient of a removed sample is the sampled value . dy: continuous across the edge, still compared
            assert rel(a.double().cpu(), b) < 1e-5, nm
        else:
            assert rel(a.double().cpu(), b) < 1e-5, nm           # (3)
