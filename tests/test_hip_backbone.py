"""GPU: blocks and whole backbones through the drop-in Python surface (mtp_amd.ViT_Win_RVSA_V3_WSZ7) against the golden
fixtures generated from the reference itself, plus oracle comparisons at BASELINE config shapes.
fp32 mode: north_star tolerance 1e-3 relative; bf16 mode: compared with the fp32 fixture AND the reference's own
bf16-autocast fixture (the reference under bf16 differs from fp32 by ~5e-3)."""
import numpy as np
import pytest
import torch

import mtp_amd
import recipe
from conftest import record_parity, rel_err
from oracle import vit_rvsa_oracle as O

pytestmark = pytest.mark.gpu
t = torch.from_numpy


def build(embed_dim, depth, heads, interval, out_indices, precision, **kw):
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=embed_dim, depth=depth, num_heads=heads, interval=interval, qkv_bias=True,
                                       use_abs_pos_emb=True, out_indices=out_indices, precision=precision, feature_dtype=torch.float32, **kw)
    net.load_state_dict(recipe.make_params(recipe.state_shapes(embed_dim, depth, heads, interval)), strict=False)
    return net.cuda()


def _check_summary(tensor, gsum, gsamples, tol, n=512, what=""):
    """max-abs error relative to the largest sample, plus relative L2 over the samples (robust for bf16 noise)"""
    s, v = recipe.summarize(tensor.float().cpu(), n)
    scale = np.abs(gsamples).max() + 1e-30
    err = np.abs(v - gsamples).max() / scale
    l2 = np.linalg.norm(v - gsamples) / (np.linalg.norm(gsamples) + 1e-30)
    assert err < tol and l2 < tol, (what, float(err), float(l2))
    assert abs(s[1] - gsum[1]) < tol * gsum[1], what
    return float(err)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_small_model_forward_and_all_gradients_vs_reference(golden, precision, tol):
    """fixture f8: 6-block C=128 model, train mode (drop_path 0), every output and every parameter gradient."""
    g = golden("f8_small.npz")
    net = build(128, 6, 2, 3, [1, 2, 3, 5], precision).train()
    img = recipe.make_input(2, 224, 224, seed=99).cuda().requires_grad_(True)
    feats = net(img)
    assert [tuple(f.shape) for f in feats] == [(2, 128, 56, 56), (2, 128, 28, 28), (2, 128, 14, 14), (2, 128, 7, 7)]
    loss = 0
    for i, f in enumerate(feats):
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], tol, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 200 + i).cuda()).sum()
    assert rel_err(feats[2].cpu(), g["f2"]) < tol and rel_err(feats[3].cpu(), g["f3"]) < tol
    loss.backward()
    # gradients: fp32 mode 1e-3 = north_star's bar (measured: <= 4.4e-6 on all 123 tensors, profiles/r02_parity_errors.json group small_fp32_vs_fp32_maxabs).
    # bf16 mode, WORST SINGLE ENTRY relative to the tensor's largest (a max-abs metric: one unlucky element decides): measured <= 0.23
    # on the ordinary gradients and <= 0.42 on the sampling heads (group small_bf16_vs_fp32_maxabs) -- the same tensors are within
    # 4.3e-2 / 0.30 relative L2 of the reference's own bf16-autocast gradients (test_small_model_bf16_gradients_vs_reference_bf16_autocast,
    # which is the meaningful bf16 bound); the reference's bf16 run itself is 6e-2 / 0.58 away from its fp32 run.
    gt = tol if precision == "fp32" else 0.35
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], gt, 2048, "dimg")
    errs = {}
    for n, p in net.named_parameters():
        if "nograd_" + n in g:
            assert p.grad is None
        elif "g_" + n in g:
            errs[n] = rel_err(p.grad.cpu(), g["g_" + n])
        else:
            errs[n] = _check_summary(p.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], gt, 1024, n)
    for n, v in errs.items():
        record_parity("small_%s_vs_fp32_maxabs" % precision, n, v)
    # the RVSA sampling heads' gradients go through d(bilinear)/d(coord) = DIFFERENCES of neighbouring K/V rows: rounding K/V to
    # bf16 is amplified there (same numbers with the f32-math VALU kernels on bf16 data), so they get their own bf16 bound
    for n, e in errs.items():
        assert e < (0.6 if (precision == "bf16" and "sampling" in n) else gt), (n, e)


def test_small_model_bf16_vs_reference_bf16_autocast(golden):
    g = golden("f8_small.npz")
    net = build(128, 6, 2, 3, [1, 2, 3, 5], "bf16").eval()
    with torch.no_grad():
        feats = net(recipe.make_input(2, 224, 224, seed=99).cuda())
    for i, f in enumerate(feats):
        _check_summary(f, g["bf16_f%d_sum" % i], g["bf16_f%d_samples" % i], 4e-2, 2048)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_vit_b_config1_forward_and_gradients(golden, precision, tol):
    """BASELINE config 1 (ViT-B/16, batch 2, 224x224) through the factory, vs fixture f7 (reference run)."""
    g = golden("f7_vitb.npz")

    class A:
        image_size = 224
        use_ckpt = "False"
    A.precision = precision
    net = mtp_amd.vit_b_rvsa(A)
    net.feature_dtype = torch.float32
    net.load_state_dict(recipe.make_params(recipe.state_shapes(768, 12, 12, 3)), strict=False)
    net = net.cuda().eval()
    img = recipe.make_input(2, 224, 224).cuda().requires_grad_(True)
    feats = net.forward_features(img)
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["f%d_shape" % i])
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], tol)
    loss = sum(f.mean() for f in feats)
    assert abs(loss.item() - float(g["loss"])) < tol * max(1.0, abs(float(g["loss"])))
    loss.backward()
    gt = tol if precision == "fp32" else 5 * tol          # fp32 mode: north_star's 1e-3 on the gradients as well
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], gt)
    P = dict(net.named_parameters())
    for k in g:
        if k.startswith("g_") and k.endswith("_samples"):
            n = k[2:-len("_samples")]
            _check_summary(P[n].grad, g["g_%s_sum" % n], g[k], gt)
    assert P["norm.weight"].grad is None


def test_checkpointing_and_eval_paths_agree():
    net = build(128, 6, 2, 3, [1, 2, 3, 5], "fp32").train()
    img = recipe.make_input(2, 224, 224, seed=3).cuda()
    f0 = net(img)
    loss = sum((f * f).mean() for f in f0)
    loss.backward()
    g0 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    net.zero_grad()
    net.use_checkpoint = True
    f1 = net(img)
    sum((f * f).mean() for f in f1).backward()
    for a, b in zip(f0, f1):
        assert torch.equal(a, b)
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad.cpu(), g0[n].cpu()) < 1e-5, n
    with torch.no_grad():
        f2 = net.eval()(img)
    for a, b in zip(f0, f2):
        assert torch.equal(a, b)


def test_drop_path_training_matches_oracle_with_same_masks():
    """stochastic depth (VIT:31-42): run the HIP path, read back the per-sample factors it drew, replay them in the oracle."""
    net = build(128, 4, 2, 3, [0, 1, 2, 3], "fp32", drop_path_rate=0.5).train()
    torch.manual_seed(5)
    img = recipe.make_input(4, 224, 224, seed=4).cuda()
    eng = net._engine()
    feats, ctx = eng.forward(img, training=True, need_grad=True, feature_dtype=torch.float32)
    scales = [None if a is None else (a.cpu(), b.cpu()) for a, b in ctx["dps"]]
    assert scales[0] is None and scales[2] is not None
    p = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.backbone_forward(img.cpu(), p, 4, 2, 3, [0, 1, 2, 3], dp_scales=scales)
    for a, b in zip(feats, ref):
        assert rel_err(a.cpu(), b) < 1e-3


def test_padded_resolution_512_forward_and_gradients_vs_oracle():
    """512x512 input: 32x32 tokens -> RVSA pads to 35x35 (25 windows, VIT:298-310).  Window blocks only (full attention at N=1024 is
    SURVEY 8f-4); features, the input gradient and EVERY parameter gradient against the oracle's autograd at north_star's 1e-3 (round 5, VERDICT r04 #4b:
    the backward of the padded geometry was covered at op level only).  Input seed 6 has no sample within 4e-5 px of a cell edge of the bilinear
    interpolation (tests/golden/kinks.py: 6.7e-5 px over 19600 coordinates; the test asserts it)."""
    import kinks
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=512, embed_dim=128, depth=4, num_heads=2, interval=5, qkv_bias=True, use_abs_pos_emb=True,
                                       out_indices=[0, 1, 2, 3], precision="fp32", feature_dtype=torch.float32, drop_path_rate=0.0)
    sd = recipe.make_params({k: v.shape for k, v in net.state_dict().items() if v.dtype.is_floating_point}, seed=11)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().train()
    img = recipe.make_input(1, 512, 512, seed=6)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    d, _ = kinks.min_edge_distance(img, {k: v.detach() for k, v in p.items()}, 4, 2, 5, [0, 1, 2, 3])
    assert d >= 4e-5, d
    x = img.cuda().requires_grad_(True)
    feats = net(x)
    xr = img.clone().requires_grad_(True)
    ref = O.backbone_forward(xr, p, 4, 2, 5, [0, 1, 2, 3])
    for i, (a, b) in enumerate(zip(feats, ref)):
        v = rel_err(a.detach().cpu(), b.detach())
        record_parity("vit_512_fp32_bwd", "f%d" % i, v)
        assert v < 1e-3
    ws = [recipe.loss_weights(f.shape, 900 + i) for i, f in enumerate(ref)]
    sum((f * w.cuda()).sum() for f, w in zip(feats, ws)).backward()
    sum((f * w).sum() for f, w in zip(ref, ws)).backward()
    v = rel_err(x.grad.cpu(), xr.grad)
    record_parity("vit_512_fp32_bwd", "dimg", v)
    assert v < 1e-3
    for n, q in net.named_parameters():
        if p[n].grad is None:
            assert q.grad is None, n
        else:
            v = rel_err(q.grad.cpu(), p[n].grad)
            record_parity("vit_512_fp32_bwd", n, v)
            assert v < 1e-3, (n, v)


def test_full_size_roundtrip_properties_vit_l_shapes():
    """BASELINE config-3 sizes (T = 64*196 tokens, C = 1024): size-independent properties of the hot kernels where a full CPU
    reference would take minutes.  Linearity: inputs and weights on a coarse binary grid (multiples of 1/8 resp. 1/64, |.| <= 2),
    so every product and every partial sum of 1024 of them is exact in f32 -- y(a) + y(b) == y(a + b) must hold BIT FOR BIT in
    f32 output, for the kernel family the dispatcher picks at this size (the pipelined 224-row tiles) and for the 128-wide one."""
    from mtp_amd import ops
    T, C = 64 * 196, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.randint(-8, 9, (T, C), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
    b = (torch.randint(-8, 9, (T, C), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
    w = (torch.randint(-64, 65, (3 * C, C), device="cuda", generator=g).float() / 64).to(torch.bfloat16)
    s_bf = (a.float() + b.float()).to(torch.bfloat16)
    assert torch.equal(s_bf.float(), a.float() + b.float())           # the sum itself is exact in bf16 on this grid
    for variant in (0, 1024):
        ya = ops.gemm_nt(a, w, torch.empty(T, 3 * C, device="cuda"), variant=variant)
        yb = ops.gemm_nt(b, w, torch.empty(T, 3 * C, device="cuda"), variant=variant)
        ys = ops.gemm_nt(s_bf, w, torch.empty(T, 3 * C, device="cuda"), variant=variant)
        assert torch.equal(ya + yb, ys), variant
        # a few rows against an f64 host dot product: exact, too
        idx = torch.tensor([0, 1, 777, 6000, T - 1])
        ref = a[idx].double().cpu() @ w.double().cpu().t()
        assert torch.equal(ya[idx].double().cpu(), ref), variant
    assert ops.gemm_nt_tile(a, w, torch.empty(T, 3 * C, device="cuda")) == 256
    f = ops.tokens_to_nchw(a, torch.empty(64, C, 14, 14, device="cuda", dtype=torch.bfloat16), 64, 14, 14, 0)
    assert torch.equal(ops.nchw_to_tokens(f, torch.empty_like(a), 64, 14, 14, 0), a)


def _l2(v, ref):
    v, ref = np.asarray(v, dtype=np.float64).ravel(), np.asarray(ref, dtype=np.float64).ravel()
    return float(np.linalg.norm(v - ref) / (np.linalg.norm(ref) + 1e-30))


def _sampled(tensor, n):
    return recipe.summarize(tensor.detach().float().cpu(), n)[1]


def _vit_l(precision):
    class A:
        image_size = 224
        use_ckpt = "False"
    A.precision = precision
    net = mtp_amd.vit_l_rvsa(A)
    net.feature_dtype = torch.float32
    net.load_state_dict(recipe.make_params(recipe.state_shapes(1024, 24, 16, 6)), strict=False)
    return net.cuda().eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_vit_l_headline_model_vs_reference(golden, precision):
    """fixture f13 = the reference's own vit_l_rvsa (1024 / 24 blocks / 16 heads, the model of BASELINE configs 3 and 4), batch 2:
    four feature maps, input gradient and twelve parameter gradients from every part of the network.
    fp32 mode: north_star's 1e-3, features AND gradients (the fixture's input has no bilinear sample within 1e-5 px of a cell edge:
    recipe.F13_INPUT_SEED).  bf16 mode: relative L2 against the reference's OWN bf16-autocast run
    (two bf16 roundings of one computation), and -- looser -- max-abs against the fp32 run."""
    g = golden("f13_vitl.npz")
    net = _vit_l(precision)
    assert list(net.out_indices) == list(g["out_indices"]) and sum(p.numel() for p in net.parameters()) == int(g["n_params"][0])
    assert int(g["input_seed"][0]) == recipe.F13_INPUT_SEED
    img = recipe.make_input(2, 224, 224, seed=recipe.F13_INPUT_SEED).cuda().requires_grad_(True)
    feats = net.forward_features(img)
    loss = 0
    errs = {}
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["f%d_shape" % i])
        v = _sampled(f, 4096)
        errs["f%d_vs_fp32_maxabs" % i] = np.abs(v - g["f%d_samples" % i]).max() / np.abs(g["f%d_samples" % i]).max()
        errs["f%d_vs_fp32_l2" % i] = _l2(v, g["f%d_samples" % i])
        errs["f%d_vs_bf16ref_l2" % i] = _l2(v, g["bf16_f%d_samples" % i])
        loss = loss + (f * recipe.loss_weights(f.shape, 600 + i).cuda()).sum()
    loss.backward()
    P = dict(net.named_parameters())
    grads = {"dimg": img.grad}
    for k in g:
        if k.startswith("g_") and k.endswith("_samples"):
            grads[k[2:-len("_samples")]] = P[k[2:-len("_samples")]].grad
    for n, gr in grads.items():
        key = "dimg" if n == "dimg" else "g_" + n
        v = _sampled(gr, 4096 if n == "dimg" else 2048)
        errs[n + "_vs_fp32_maxabs"] = np.abs(v - g[key + "_samples"]).max() / np.abs(g[key + "_samples"]).max()
        errs[n + "_vs_fp32_l2"] = _l2(v, g[key + "_samples"])
        errs[n + "_vs_bf16ref_l2"] = _l2(v, g["bf16_" + key + "_samples"])
    for k, v in errs.items():
        record_parity("vit_l_b2_" + precision, k, v)
    record_parity("vit_l_b2_reference_itself", "bf16_autocast_vs_fp32_f2_l2", _l2(g["bf16_f2_samples"], g["f2_samples"]))
    record_parity("vit_l_b2_reference_itself", "bf16_autocast_vs_fp32_dimg_l2", _l2(g["bf16_dimg_samples"], g["dimg_samples"]))
    assert P["norm.weight"].grad is None and not bool(g["norm_has_grad"][0])
    if precision == "fp32":
        for k, v in errs.items():
            if k.endswith("_vs_fp32_maxabs"):
                assert v < 1e-3, (k, v)          # north_star: 1e-3 rel fp32, forward and gradients
    else:
        for k, v in errs.items():
            if k.endswith("_vs_bf16ref_l2"):
                assert v < VITL_BF16_L2[_bf16_class(k)], (k, v)
            if k.endswith("_vs_fp32_maxabs"):
                assert v < VITL_BF16_MAXABS[_bf16_class(k)], (k, v)


def test_vit_l_near_edge_input_vs_reference(golden):
    """fixture f13_vitl_hard = the same ViT-L on round 2's input (seed 2023): one bilinear sample 3.3e-6 px from a cell edge, i.e. within f32
    rounding of a kink of the interpolation.  The features are still held to 1e-3 (a kink is a kink of the DERIVATIVE); the gradients upstream
    of that RVSA block may pick the other one-sided derivative than the reference's fp32 run did and are held to 3e-3 (measured in round 2:
    up to 1.3e-3).  Keeps the kernels' behaviour on near-edge samples under test next to the kink-free f13."""
    g = golden("f13_vitl_hard.npz")
    assert int(g["input_seed"][0]) == recipe.F13_HARD_INPUT_SEED and float(g["min_edge_distance_px"][0]) < recipe.F13_MIN_EDGE_DISTANCE
    net = _vit_l("fp32")
    img = recipe.make_input(2, 224, 224, seed=recipe.F13_HARD_INPUT_SEED).cuda().requires_grad_(True)
    feats = net.forward_features(img)
    loss = 0
    for i, f in enumerate(feats):
        v = _sampled(f, 4096)
        err = np.abs(v - g["f%d_samples" % i]).max() / np.abs(g["f%d_samples" % i]).max()
        record_parity("vit_l_b2_hard_fp32", "f%d_maxabs" % i, err)
        assert err < 1e-3, (i, err)
        loss = loss + (f * recipe.loss_weights(f.shape, 600 + i).cuda()).sum()
    loss.backward()
    P = dict(net.named_parameters())
    grads = {"dimg": img.grad}
    for k in g:
        if k.startswith("g_") and k.endswith("_samples"):
            grads[k[2:-len("_samples")]] = P[k[2:-len("_samples")]].grad
    for n, gr in grads.items():
        key = "dimg" if n == "dimg" else "g_" + n
        v = _sampled(gr, 4096 if n == "dimg" else 2048)
        err = np.abs(v - g[key + "_samples"]).max() / np.abs(g[key + "_samples"]).max()
        record_parity("vit_l_b2_hard_fp32", n + "_maxabs", err)
        assert err < 3e-3, (n, err)


def _bf16_class(k):
    if k[0] == "f" and k[1].isdigit():
        return "fwd"
    if "sampling" in k:
        return "sampling"
    return "grad"


# bf16-mode bounds of the ViT-L test, each ~2x the measured value (profiles/r02_parity_errors.json, group vit_l_b2_bf16):
#   forward maps   measured 6.4e-3 .. 7.9e-3 relative L2 vs the reference's bf16-autocast run (5.3e-3 .. 6.5e-3 vs its fp32 run; the
#                  reference's own bf16 run is 7.1e-3 away from its fp32 run)
#   gradients      measured <= 9.7e-2 (typically 6e-2 .. 8e-2: 24 blocks of bf16 rounding; the reference's own bf16 input gradient
#                  is 8.0e-2 away from its fp32 one, ours 6.4e-2)
#   sampling heads measured <= 0.22: d(bilinear)/d(position) is a DIFFERENCE of neighbouring K / V rows, which bf16 rounding of K / V
#                  hits hardest (the reference's own bf16 run moves these gradients by up to 0.58 on the small model, group
#                  small_reference_bf16_vs_fp32_l2)
#   max-abs (WORST SINGLE sampled entry vs the largest): a noisy statistic for bf16 gradients -- pos_embed measured 0.086 and 0.204 in
#                  two builds whose relative L2 on that tensor differs by < 5 %; kept as a coarse guard only, at 0.3 / 0.6
VITL_BF16_L2 = {"fwd": 2e-2, "grad": 0.15, "sampling": 0.45}
VITL_BF16_MAXABS = {"fwd": 2e-2, "grad": 0.3, "sampling": 0.6}


def test_small_model_bf16_gradients_vs_reference_bf16_autocast(golden):
    """f8 (6 blocks, C = 128), EVERY parameter gradient of the bf16 mode against the reference's bf16-autocast gradients,
    relative L2 over the whole tensor (or its 1024 samples)"""
    g = golden("f8_small.npz")
    net = build(128, 6, 2, 3, [1, 2, 3, 5], "bf16").train()
    img = recipe.make_input(2, 224, 224, seed=99).cuda().requires_grad_(True)
    feats = net(img)
    sum((f * recipe.loss_weights(f.shape, 200 + i).cuda()).sum() for i, f in enumerate(feats)).backward()
    errs = {"dimg": _l2(_sampled(img.grad, 2048), g["bf16_dimg_samples"])}
    ref_noise = {"dimg": _l2(g["bf16_dimg_samples"], g["dimg_samples"])}
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        if "bf16_g_" + n in g:
            errs[n] = _l2(p.grad.float().cpu().numpy(), g["bf16_g_" + n])
            ref_noise[n] = _l2(g["bf16_g_" + n], g["g_" + n])
        else:
            errs[n] = _l2(_sampled(p.grad, 1024), g["bf16_gs_%s_samples" % n])
            ref_noise[n] = _l2(g["bf16_gs_%s_samples" % n], g["gs_%s_samples" % n])
    for n, v in errs.items():
        record_parity("small_bf16_vs_bf16ref_l2", n, v)
        record_parity("small_reference_bf16_vs_fp32_l2", n, ref_noise[n])
    for n, v in errs.items():
        assert v < SMALL_BF16_L2["sampling" if "sampling" in n else "grad"], (n, v, ref_noise[n])


# measured (profiles/r02_parity_errors.json, group small_bf16_vs_bf16ref_l2): <= 4.3e-2 on the 97 ordinary gradients, <= 0.30 on the
# sampling heads, where the reference's own bf16 run is up to 0.58 away from its fp32 run (group small_reference_bf16_vs_fp32_l2)
SMALL_BF16_L2 = {"grad": 0.09, "sampling": 0.6}


def test_uint8_input_through_fused_preprocessor_equals_preprocessed_f32_input():
    """raw (B,H,W,3) uint8 batch -> same feature maps as feeding the oracle-preprocessed f32 NCHW batch (the
    patch rows are bit-identical, so everything downstream is too); 200x150 is padded to 224x160 like MTP_DataPreprocessor does."""
    net = build(128, 4, 2, 2, [0, 1, 2, 3], "fp32").eval()
    net.set_data_preprocessor()          # models.py:37-41 defaults
    for H, W in ((224, 224), (200, 150)):
        Hp, Wp = -(-H // 32) * 2, -(-W // 32) * 2
        g = torch.Generator().manual_seed(H)
        raw = torch.randint(0, 256, (2, H, W, 3), generator=g, dtype=torch.uint8)
        pp = net.data_preprocessor
        x = O.preprocess(raw, pp["mean"], pp["std"], pp["bgr_to_rgb"], pp["pad_size_divisor"], pp["pad_value"])
        with torch.no_grad():
            net.pos_embed.data = torch.randn(1, Hp * Wp, 128, generator=torch.Generator().manual_seed(1)).cuda() * 0.02
            a = net(raw.cuda())
            b = net(x.cuda())
        assert a[2].shape == (2, 128, Hp, Wp)
        for fa, fb in zip(a, b):
            assert torch.equal(fa, fb)
    net.data_preprocessor = None
    with pytest.raises(ValueError):
        net(raw.cuda())


def test_host_batches_through_the_double_buffered_prefetcher():
    """host uint8 batches -> pinned staging -> side-stream H2D -> fused preprocess + patch-embed; every batch must
    give exactly what a plain .cuda() copy of it gives, with two device buffers recycled under a running stream of work"""
    from mtp_amd.data import HostBatchPrefetcher
    net = build(128, 4, 2, 2, [0, 1, 2, 3], "fp32").eval()
    net.set_data_preprocessor()
    g = torch.Generator().manual_seed(11)
    host = [torch.randint(0, 256, (2, 224, 224, 3), generator=g, dtype=torch.uint8) for _ in range(5)]
    pf = HostBatchPrefetcher(iter(host), device="cuda", depth=2)
    outs = []
    with torch.no_grad():
        for dev_batch in pf:
            assert dev_batch.is_cuda and dev_batch.dtype == torch.uint8
            outs.append([f.clone() for f in net(dev_batch)])
        torch.cuda.synchronize()
        for b, got in zip(host, outs):
            for fa, fb in zip(got, net(b.cuda())):
                assert torch.equal(fa, fb)
    assert len(outs) == 5 and len(pf.slots) == 2 and pf.bytes_copied == 5 * 2 * 224 * 224 * 3


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_vitdet_style_finetune_variant_vs_reference(golden, precision, tol):
    """fixture f9 = the reference's mmdet `RVSA_MTP`: full attention without rel-pos, last block -> final norm ->
    fpn1-4 on that one map; every parameter (norm.* too) gets a gradient.  Here: mtp_amd.RVSA_MTP_det, tuple output."""
    g = golden("f9_vitdet.npz")
    net = mtp_amd.RVSA_MTP_det(img_size=224, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True,
                               out_indices=[1, 2, 3, 3], precision=precision, feature_dtype=torch.float32)
    shapes = {k: v for k, v in recipe.state_shapes(128, 4, 2, 2).items() if "full_attn_rel_pos" not in k}
    assert [k for k, v in net.state_dict().items() if v.dtype.is_floating_point] == [str(k) for k in g["keys"]]
    msg = net.load_state_dict(recipe.make_params(shapes), strict=False)
    assert not msg.unexpected_keys and all(k.endswith("relative_position_index") for k in msg.missing_keys)
    net = net.cuda().train()
    img = recipe.make_input(2, 224, 224, seed=77).cuda().requires_grad_(True)
    feats = net(img)
    assert isinstance(feats, tuple) and [tuple(f.shape) for f in feats] == [(2, 128, 56, 56), (2, 128, 28, 28), (2, 128, 14, 14), (2, 128, 7, 7)]
    loss = 0
    for i, f in enumerate(feats):
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], tol, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 300 + i).cuda()).sum()
    assert rel_err(feats[2].cpu(), g["f2"]) < tol and rel_err(feats[3].cpu(), g["f3"]) < tol
    loss.backward()
    gt = tol if precision == "fp32" else 0.35
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], gt, 2048, "dimg")
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        if "g_" + n in g:
            e = rel_err(p.grad.cpu(), g["g_" + n])
        else:
            e = _check_summary(p.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], gt, 1024, n)
        assert e < (0.6 if (precision == "bf16" and "sampling" in n) else gt), (n, e)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_tap_only_finetune_variant_vs_reference(golden, precision, tol):
    """fixture f10 = the reference's mmpretrain `RVSA_MTP`: block outputs at out_indices (two here) as NCHW maps,
    no fpn ops; fpn* / norm.* / blocks after the last tap get no gradient.  Here: mtp_amd.RVSA_MTP_taps."""
    g = golden("f10_taps.npz")
    net = mtp_amd.RVSA_MTP_taps(img_size=224, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True,
                                out_indices=[1, 3], precision=precision, feature_dtype=torch.float32, frozen_stages=-1)
    assert [k for k, v in net.state_dict().items() if v.dtype.is_floating_point] == [str(k) for k in g["keys"]]
    net.load_state_dict(recipe.make_params(recipe.state_shapes(128, 4, 2, 2)), strict=False)
    net = net.cuda().train()
    img = recipe.make_input(2, 224, 224, seed=55).cuda().requires_grad_(True)
    feats = net(img)
    assert isinstance(feats, tuple) and len(feats) == 2
    assert rel_err(feats[0].cpu(), g["f0"]) < tol and rel_err(feats[1].cpu(), g["f1"]) < tol
    loss = sum((f * recipe.loss_weights(f.shape, 400 + i).cuda()).sum() for i, f in enumerate(feats))
    loss.backward()
    gt = tol if precision == "fp32" else 0.35
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], gt, 2048, "dimg")
    for n, p in net.named_parameters():
        if "nograd_" + n in g:
            assert p.grad is None, n
        elif "g_" + n in g:
            assert rel_err(p.grad.cpu(), g["g_" + n]) < (0.6 if (precision == "bf16" and "sampling" in n) else gt), n
        else:
            e = _check_summary(p.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], gt, 1024, n)
            assert e < (0.6 if (precision == "bf16" and "sampling" in n) else gt), (n, e)
    # open-cd's frozen stages: patch embed, pos_embed and the first block stop requiring gradients; the rest still trains
    net.frozen_stages = 1
    net._freeze_stages()
    net.zero_grad()
    sum(f.sum() for f in net(img.detach())).backward()
    assert net.patch_embed.proj.weight.grad is None and net.pos_embed.grad is None and net.blocks[0].attn.qkv.weight.grad is None
    assert net.blocks[1].attn.qkv.weight.grad is not None and not net.blocks[0].training and net.blocks[1].training


@pytest.mark.parametrize("seed,tol,case", [(8, 1e-3, "kink_free"), (6, 5e-3, "hard")])
def test_448_pretraining_resolution_forward_and_gradients_vs_oracle(seed, tol, case):
    """448x448 (the resolution MTP really pretrains at): 28x28 = 784 tokens per image -> RVSA with 16 windows and
    full attention beyond one workgroup (generic forward + three-pass backward kernels); fp32 mode vs the oracle's autograd.
    kink_free (input seed 8: no RVSA sample within 1e-4 px of a cell edge of the bilinear interpolation, tests/golden/kinks.py -- asserted): the input
    gradient and every parameter gradient at north_star's 1e-3, values into the parity table (VERDICT r04 #4a).  hard (input seed 6: one sample 2.4e-6 px
    from an edge, i.e. within a few f32 ulps -- its one-sided derivative may flip): gradients at 5e-3."""
    import kinks
    kw = dict(img_size=448, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 3])
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(precision="fp32", feature_dtype=torch.float32, **kw)
    sd = recipe.make_params({k: v.shape for k, v in net.state_dict().items() if v.dtype.is_floating_point}, seed=21)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().train()
    img = recipe.make_input(1, 448, 448, seed=seed)
    x = img.cuda().requires_grad_(True)
    feats = net(x)
    assert [tuple(f.shape) for f in feats] == [(1, 128, 112, 112), (1, 128, 56, 56), (1, 128, 28, 28), (1, 128, 14, 14)]
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    d, _ = kinks.min_edge_distance(img, {k: v.detach() for k, v in p.items()}, 4, 2, 2, [0, 1, 2, 3])
    assert (d >= 1e-4) if case == "kink_free" else (d < 1e-5), d
    group = "vit_448_fp32" if case == "kink_free" else "vit_448_fp32_hard"
    xr = img.clone().requires_grad_(True)
    ref = O.backbone_forward(xr, p, 4, 2, 2, [0, 1, 2, 3])
    ws = [recipe.loss_weights(f.shape, 500 + i) for i, f in enumerate(ref)]
    for i, (a, b) in enumerate(zip(feats, ref)):
        v = rel_err(a.detach().cpu(), b.detach())
        record_parity(group, "f%d" % i, v)
        assert v < 1e-3
    sum((f * w.cuda()).sum() for f, w in zip(feats, ws)).backward()
    sum((f * w).sum() for f, w in zip(ref, ws)).backward()
    v = rel_err(x.grad.cpu(), xr.grad)
    record_parity(group, "dimg", v)
    assert v < tol
    for n, q in net.named_parameters():
        if p[n].grad is None:
            assert q.grad is None, n
        else:
            v = rel_err(q.grad.cpu(), p[n].grad)
            record_parity(group, n, v)
            assert v < tol, (n, v)


def test_non_square_input_taller_rel_pos_table_through_autograd_and_trainer():
    """224 x 160 input -> 14 x 10 token grid: the reference sizes BOTH full-attention tables from patch_shape[0] (VIT:81-84), so rel_pos_w has 27
    rows of which the 10-column grid uses 19.  Both backward paths -- the autograd Function with separately allocated gradients and
    the trainer's flat buffer -- must write rows [0, 19) only and agree with the oracle; the unused rows stay exactly zero."""
    from mtp_amd.parallel import DataParallelTrainer
    kw = dict(img_size=(224, 160), embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 3])
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(precision="fp32", feature_dtype=torch.float32, drop_path_rate=0.0, **kw)
    sd = recipe.make_params({k: v.shape for k, v in net.state_dict().items() if v.dtype.is_floating_point}, seed=31)
    net.load_state_dict(sd, strict=False)
    assert tuple(net.state_dict()["blocks.1.attn.full_attn_rel_pos_w"].shape) == (27, 64)
    net = net.cuda().train()
    img = recipe.make_input(1, 224, 160, seed=9)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    ref = O.backbone_forward(img, p, 4, 2, 2, [0, 1, 2, 3])
    ws = [recipe.loss_weights(f.shape, 700 + i) for i, f in enumerate(ref)]
    sum((f * w).sum() for f, w in zip(ref, ws)).backward()
    # ---- autograd path
    feats = net(img.cuda())
    for a, b in zip(feats, ref):
        assert rel_err(a.cpu(), b) < 1e-3
    sum((f * w.cuda()).sum() for f, w in zip(feats, ws)).backward()
    for n, q in net.named_parameters():
        if p[n].grad is not None:
            assert rel_err(q.grad.cpu(), p[n].grad) < 5e-3, n
    for blk in (1, 3):
        g = dict(net.named_parameters())["blocks.%d.attn.full_attn_rel_pos_w" % blk].grad
        assert float(g[19:].abs().max()) == 0.0 and float(g[:19].abs().max()) > 0
    # ---- trainer path (flat gradient buffer: a write past row 18 would land in the next parameter's gradient)
    tr = DataParallelTrainer(net, total_steps=10)

    def loss_and_grads(fs):
        return sum((f * w.cuda()).sum() for f, w in zip(fs, ws)), [w.cuda().to(f.dtype) for f, w in zip(fs, ws)]
    tr.flat.grad.zero_()
    tr.step(img.cuda(), loss_and_grads)
    for n in ("blocks.1.attn.full_attn_rel_pos_w", "blocks.1.attn.full_attn_rel_pos_h", "blocks.3.attn.full_attn_rel_pos_w", "blocks.1.attn.qkv.weight", "fpn1.0.weight"):
        assert rel_err(tr.flat.G[n].cpu(), p[n].grad) < 5e-3, n
    assert float(tr.flat.G["blocks.1.attn.full_attn_rel_pos_w"][19:].abs().max()) == 0.0


@pytest.mark.parametrize("size", [448, 1024])
def test_large_inputs_bf16_flash_attention_path_vs_oracle(size):
    """bf16 mode at 448^2 (784 tokens per image: MTP's pretraining resolution) and 1024^2 (4096 tokens: the detection fine-tunes'
    RVSA_MTP_branches(img_size=1024)): the full-attention blocks run the flash forward and the flash MFMA backward
    (attn_full_flash_bwd.hip).  Relative L2 against the fp32 oracle; bounds as for the ViT-L bf16 run (the bf16 rounding of one
    forward / backward: forward 2e-2, gradients 0.15, RVSA sampling heads 0.6), measured values go to the parity table."""
    kw = dict(img_size=size, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 3])
    depth = 4
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(precision="bf16", feature_dtype=torch.float32, **kw)
    sd = recipe.make_params({k: v.shape for k, v in net.state_dict().items() if v.dtype.is_floating_point}, seed=31)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().train()
    img = recipe.make_input(1, size, size, seed=9)
    x = img.cuda().requires_grad_(True)
    feats = net(x)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    xr = img.clone().requires_grad_(True)
    ref = O.backbone_forward(xr, p, depth, 2, 2, kw["out_indices"])
    ws = [recipe.loss_weights(f.shape, 700 + i) for i, f in enumerate(ref)]
    group = "large_input_%d_bf16_vs_fp32_l2" % size
    for i, (a, b) in enumerate(zip(feats, ref)):
        v = _l2(a.detach().float().cpu().numpy(), b.detach().numpy())
        record_parity(group, "f%d" % i, v)
        assert v < 2e-2, (i, v)
    sum((f * w.cuda()).sum() for f, w in zip(feats, ws)).backward()
    sum((f * w).sum() for f, w in zip(ref, ws)).backward()
    v = _l2(x.grad.cpu().numpy(), xr.grad.numpy())
    record_parity(group, "dimg", v)
    assert v < 0.15
    for n, q in net.named_parameters():
        if p[n].grad is None:
            assert q.grad is None, n
            continue
        v = _l2(q.grad.cpu().numpy(), p[n].grad.numpy())
        record_parity(group, n, v)
        assert v < (0.6 if "sampling" in n else 0.15), (n, v)


def test_vit_b_config2_whole_at_batch_32():
    """BASELINE configs[1] as a whole: ViT-B + RVSA, batch 32, 224^2, bf16 mode, forward + backward through the trainer's path of
    bench.py -- every map finite, the GEMM kernels the bench line runs on (6272 token rows: strip kernel for N = 768, 8-phase kernel for N = 2304 / 3072),
    and -- the images of a batch being independent -- the features and the input gradient of images 3 and 20 against the fp32 oracle run on those two images."""
    from mtp_amd import ops

    class A:
        image_size = 224
        use_ckpt = "False"
        precision = "bf16"
    net = mtp_amd.vit_b_rvsa(A)
    net.feature_dtype = torch.float32
    params = recipe.make_params(recipe.state_shapes(768, 12, 12, 3))
    net.load_state_dict(params, strict=False)
    net = net.cuda().eval()         # (eval: no drop path, so that the oracle sees the same function)
    img = recipe.make_input(32, 224, 224, seed=5)
    x = img.cuda().requires_grad_(True)
    feats = net.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(32, 768, 56, 56), (32, 768, 28, 28), (32, 768, 14, 14), (32, 768, 7, 7)]
    assert all(bool(torch.isfinite(f).all()) for f in feats)
    ws = [recipe.loss_weights((2,) + tuple(f.shape[1:]), 900 + i) for i, f in enumerate(feats)]        # cotangents of the two checked images; zero elsewhere
    pick = [3, 20]
    cot = []
    for f, w in zip(feats, ws):
        c = torch.zeros_like(f)
        c[pick] = w.cuda()
        cot.append(c)
    torch.autograd.backward(feats, cot)
    assert bool(torch.isfinite(x.grad).all())
    P = dict(net.named_parameters())
    assert all(bool(torch.isfinite(q.grad).all()) for n, q in P.items() if q.grad is not None) and P["norm.weight"].grad is None
    assert float(x.grad[[0, 1, 2, 4, 31]].abs().max()) == 0.0       # images that received no cotangent get no gradient: nothing leaks across the batch
    # the kernels of the bench line (tools: profiles/r05_vitb_b32_gemm_shapes.txt): tile width by problem
    T = 32 * 196
    a = torch.empty(T, 768, device="cuda", dtype=torch.bfloat16)
    for N, K, want in ((768, 768, 64), (2304, 768, 256), (3072, 768, 256), (768, 3072, 64)):
        ak = a if K == 768 else torch.empty(T, K, device="cuda", dtype=torch.bfloat16)
        got = ops.gemm_nt_tile(ak, torch.empty(N, K, device="cuda", dtype=torch.bfloat16), torch.empty(T, N, device="cuda", dtype=torch.bfloat16))
        assert got == want, (N, K, got, want)
    # the two images through the oracle (fp32, CPU)
    p = {k: v.clone().requires_grad_(False) for k, v in params.items()}
    xr = img[pick].clone().requires_grad_(True)
    ref = O.backbone_forward(xr, p, 12, 12, 3, [3, 5, 7, 11])
    sum((f * w).sum() for f, w in zip(ref, ws)).backward()
    for i, (f, r) in enumerate(zip(feats, ref)):
        v = _l2(f[pick].detach().float().cpu().numpy(), r.detach().numpy())
        record_parity("vit_b_batch32_bf16_vs_fp32_oracle_l2", "f%d" % i, v)
        assert v < 2e-2, (i, v)
    v = _l2(x.grad[pick].cpu().numpy(), xr.grad.numpy())
    record_parity("vit_b_batch32_bf16_vs_fp32_oracle_l2", "dimg", v)
    assert v < 0.15, v


def test_window_partition_reverse_on_device_bit_exact(golden):
    """VIT:113-140 (dead code in the reference's forward, kept in the module surface): the same view / permute on device tensors,
    bit for bit against fixture f1, and partition -> reverse is the identity on a bf16 activation-sized tensor"""
    from mtp_amd.backbone import window_partition, window_reverse
    g = golden("f1_index.npz")
    x = torch.from_numpy(g["wp_in"]).cuda()
    w = window_partition(x, 7)
    assert w.is_cuda and np.array_equal(w.cpu().numpy(), g["wp_out"])
    assert np.array_equal(window_reverse(w, 7, 14, 21).cpu().numpy(), g["wr_out"])
    a = torch.randn(4, 28, 28, 256, device="cuda").to(torch.bfloat16)
    assert torch.equal(window_reverse(window_partition(a, 7), 7, 28, 28), a)


def test_weight_gradients_on_a_side_stream_give_the_same_gradients(monkeypatch):
    """BackboneEngine.wgrad_side_stream (A/B option): the grouped weight-gradient launches go to a side stream, ordered by events;
    the operands stay referenced until the main stream has waited.  Same gradients as the single-stream schedule (f32 atomics of
    the bias-gradient by-product reorder sums: 1e-6)."""
    net = build(256, 8, 4, 4, [1, 3, 5, 7], "bf16").train()
    img = recipe.make_input(32, 224, 224, seed=5).cuda()      # 32 x 196 tokens: a multiple of 128, so the gradients really go through the group

    def grads():
        for p in net.parameters():
            p.grad = None
        sum(f.float().mean() for f in net(img)).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    from mtp_amd.engine import BackboneEngine
    monkeypatch.setattr(BackboneEngine, "wgrad_side_stream", False)
    a = grads()
    monkeypatch.setattr(BackboneEngine, "wgrad_side_stream", True)
    b = grads()
    assert a.keys() == b.keys()
    for n in a:
        assert rel_err(b[n], a[n]) < 1e-5, n


@pytest.mark.parametrize("heads", ["standin3", "standin_seg", "mean"])
def test_bench_stand_in_heads_hand_the_backward_the_cotangents_of_their_loss(heads):
    """bench.py's stand-in task heads (`--heads standin3` = BASELINE configs[1]'s three heads, `--heads standin_seg` = configs[4]'s segmentation decoder,
    default sum of means): the cotangents they hand to the backbone's backward are the gradients of the loss they report -- checked against torch autograd on
    the same four maps (a small backbone's real outputs), and one trainer step through each runs (VERDICT r04 missing #5)."""
    import importlib.util
    import os
    from conftest import ROOT
    from mtp_amd.parallel import DataParallelTrainer
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    net = build(128, 4, 2, 3, [0, 1, 2, 3], "bf16").train()
    img = recipe.make_input(2, 224, 224, seed=3).cuda()
    with torch.no_grad():
        feats = [f.float() for f in net(img)]
    fn = bench.make_loss_and_grads(heads, [128] * 4)
    loss, grads = fn(feats)
    leaves = [f.clone().requires_grad_(True) for f in feats]
    if heads == "standin3":
        ref = 0.0
        for i, f in enumerate(leaves):
            for t_ in range(3):                                   # three heads: each scores every pixel with its own 1x1 projection and averages
                ref = ref + (f * fn.head_w[t_][i].view(1, -1, 1, 1)).sum(1).mean()
    elif heads == "standin_seg":
        ref = 0.0
        for i, f in enumerate(leaves):
            w, y = fn.seg[i]
            ref = ref + torch.nn.functional.cross_entropy(torch.einsum("bchw,kc->bkhw", f, w), y)
    else:
        ref = sum(f.mean() for f in leaves)
    rg = torch.autograd.grad(ref, leaves)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-4 * max(1.0, abs(float(ref.detach())))
    for g, r, f in zip(grads, rg, feats):
        assert g.shape == f.shape and g.is_contiguous() and rel_err(g.float().cpu(), r.cpu()) < 1e-5
    if heads != "mean":
        assert float(grads[0].std()) > 0        # not the constant cotangent of the default loss
    tr = DataParallelTrainer(net, total_steps=10, feature_dtype=torch.float32)
    l0 = tr.step(img, fn)
    assert torch.isfinite(torch.as_tensor(float(l0)))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_patch_size_8_and_layer_scale_vs_reference(golden, precision, tol):
    """fixture f14 = the reference class with patch_size = 8 (FPN tail ConvT | identity | MaxPool 2 | MaxPool 4, VIT:656-670) and init_values (layer scale,
    VIT:500-512) -- folded into the proj / fc2 weight images and un-folded in their weight gradients (engine.py): four maps, the input gradient and EVERY
    parameter gradient, gamma_1 / gamma_2 included"""
    g = golden("f14_patch8_layerscale.npz")
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=112, patch_size=8, drop_path_rate=0.0, out_indices=[0, 1, 2, 3], embed_dim=128, depth=4, num_heads=2, mlp_ratio=4,
                                       qkv_bias=True, use_abs_pos_emb=True, interval=2, use_rel_pos_bias=True, init_values=0.1, precision=precision,
                                       feature_dtype=torch.float32)
    net.load_state_dict(recipe.make_params(recipe.state_shapes(128, 4, 2, 2, 112, patch_size=8, layer_scale=True)), strict=False)
    net = net.cuda().train()
    x = recipe.make_input(2, 112, 112, seed=41).cuda().requires_grad_(True)
    feats = net(x)
    loss = 0
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["f%d" % i].shape)
        v = rel_err(f.detach().cpu(), t(g["f%d" % i]))
        record_parity("f14_patch8_layerscale_%s" % precision, "f%d" % i, v)
        assert v < tol, (i, v)
        loss = loss + (f * recipe.loss_weights(f.shape, 800 + i).cuda()).sum()
    loss.backward()
    if precision == "fp32":
        _check_summary(x.grad, g["dimg_sum"], g["dimg_samples"], tol, 2048, "dimg")
    for n, q in net.named_parameters():
        if "nograd_" + n in g:
            assert q.grad is None, n
            continue
        if "g_" + n in g:
            a, b = q.grad.cpu().numpy().ravel(), g["g_" + n].ravel()
        else:
            a, b = recipe.summarize(q.grad.float().cpu(), 1024)[1], g["gs_%s_samples" % n]
        if precision == "fp32":
            v = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
            assert v < tol, (n, v)
        else:
            # bf16 mode: relative L2 (the three max pools of this tail route a gradient to ANOTHER element when bf16 rounding flips a near-tie: isolated
            # large differences, e.g. 0.4 of the maximum on one pos_embed entry with 0.07 in L2 -- the same piecewise-differentiability as the bilinear kinks)
            v = _l2(a, b)
            assert v < (0.6 if "sampling" in n else 0.2), (n, v)
        record_parity("f14_patch8_layerscale_%s" % precision, n, v)
