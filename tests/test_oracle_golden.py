"""CPU: pin the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  fp32 tolerance 2e-5 relative-to-max (the reference's own fp32-vs-fp64
noise is 1e-6); integer ops bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

import recipe
from conftest import GOLDEN, rel_err
from oracle import index_ops as I
from oracle import vit_rvsa_oracle as O

TOL = 2e-5
t = torch.from_numpy


def test_f0_state_keys_and_factories():
    rec = json.load(open(os.path.join(GOLDEN, "f0_state_keys.json")))
    for tag, cfg in {"vit_b": (768, 12, 12, 3), "small": (128, 6, 2, 3)}.items():
        shapes = recipe.state_shapes(*cfg)
        ref_float = [(k, tuple(s)) for k, s, d in rec[tag] if d.startswith("float")]
        assert ref_float == [(k, tuple(v)) for k, v in shapes.items()]
        ref_int = [k for k, s, d in rec[tag] if not d.startswith("float")]
        assert all(k.endswith("relative_position_index") for k in ref_int)
    assert rec["factory_b"]["window_blocks"] == I.block_schedule(12, 3)
    assert rec["factory_l"]["window_blocks"] == I.block_schedule(24, 6)
    assert rec["factory_b"]["n_params"] == 93291328 and rec["factory_l"]["n_params"] == 317628800


def test_f1_integer_ops_bit_exact(golden):
    g = golden("f1_index.npz")
    assert np.array_equal(I.relative_position_index(7), g["relative_position_index"])
    for k in (7, 14):
        assert np.array_equal(I.rel_pos_dist(k, k), g["dist_h_%d" % k])
        assert np.array_equal(I.rel_pos_dist(k, k), g["dist_w_%d" % k])
    assert np.array_equal(I.window_partition(g["wp_in"], 7), g["wp_out"])
    assert np.array_equal(I.window_reverse(g["wp_out"], 7, 14, 21), g["wr_out"])
    assert np.array_equal(g["wr_out"], g["wp_in"])
    assert I.rvsa_geometry(14, 14)["He"] == 14 and I.rvsa_geometry(32, 32) == dict(
        pad_top=1, pad_down=2, pad_left=1, pad_right=2, He=35, We=35, nh=5, nw=5, div_x=4, div_y=4)


def test_patch_index_matches_unfold():
    Hp, Wp, idx = I.patch_token_index(64, 48, 16)
    img = torch.arange(64 * 48, dtype=torch.float32).reshape(1, 1, 64, 48)
    cols = torch.nn.functional.unfold(img, 16, stride=16)[0].t().reshape(Hp * Wp, 16, 16)
    assert np.array_equal(cols.long().numpy(), idx)
    c2, _ = O.patchify(img.expand(1, 3, 64, 48), 16)
    assert np.array_equal(c2[:, :256].long().numpy().reshape(-1, 16, 16), idx)


def test_fpn_index_matches_tokens_to_nchw():
    for L in (0, 1, 2):
        idx = I.fpn_nchw_index(3, 5, L)
        rows = torch.arange(3 * 5 * 4 ** L, dtype=torch.float32)[:, None]
        f = O.tokens_to_nchw(rows, 1, 3, 5, L).reshape(-1)
        assert np.array_equal(f.long().numpy()[idx], np.arange(idx.size))


def test_f2_rel_pos_spatial(golden):
    g = golden("f2_relpos.npz")
    for k in (7, 14):
        q, attn, rh, rw = (t(g[n + str(k)]) for n in ("q", "attn", "rh", "rw"))
        ih = t(I.rel_pos_dist(k, k))
        q5 = q.reshape(2, 3, k, k, 16)
        out = attn.reshape(2, 3, k, k, k, k) + torch.einsum("byhwc,hkc->byhwk", q5, rh[ih])[..., :, None] \
            + torch.einsum("byhwc,wkc->byhwk", q5, rw[ih])[..., None, :]
        assert rel_err(out.reshape(2, 3, k * k, k * k), g["out%d" % k]) < TOL


def _grads(outs, ins, cot):
    return torch.autograd.grad(outs, ins, cot, allow_unused=True, retain_graph=True)


def test_f3_full_attention(golden):
    g = golden("f3_full_attn.npz")
    p = {k[2:]: t(v).requires_grad_(True) for k, v in g.items() if k.startswith("p_")}
    x = t(g["x"]).reshape(-1, 128).requires_grad_(True)
    qkv = x @ p["qkv.weight"].t() + p["qkv.bias"]
    o, lse = O.full_attn_fwd(qkv, 2, 14, 14, 2, p["full_attn_rel_pos_h"], p["full_attn_rel_pos_w"])
    y = o @ p["proj.weight"].t() + p["proj.bias"]
    assert rel_err(y.reshape(2, 196, 128), g["y"]) < TOL
    names = list(p)
    gr = _grads(y, [x] + [p[n] for n in names], t(g["dy"]).reshape(-1, 128))
    assert rel_err(gr[0].reshape(2, 196, 128), g["dx"]) < TOL
    for n, gg in zip(names, gr[1:]):
        assert rel_err(gg, g["g_" + n]) < TOL, n
    # hand-derived backward == autograd
    do, = _grads(y, [o], t(g["dy"]).reshape(-1, 128))
    m = O.full_attn_bwd(do, qkv.detach(), o.detach(), lse.detach(), 2, 14, 14, 2, p["full_attn_rel_pos_h"].detach(), p["full_attn_rel_pos_w"].detach())
    a = _grads(o, [qkv, p["full_attn_rel_pos_h"], p["full_attn_rel_pos_w"]], do)
    for u, v in zip(m, a):
        assert rel_err(u, v) < TOL


def _samp_from(p, x, B, Hp, Wp):
    _, pooled = O.rvsa_pool_fwd(x, B, Hp, Wp)
    Ws, bs = O.sampling_weight(p["sampling_offsets.2.weight"], p["sampling_offsets.2.bias"], p["sampling_scales.2.weight"],
                               p["sampling_scales.2.bias"], p["sampling_angles.2.weight"], p["sampling_angles.2.bias"])
    return pooled @ Ws.t() + bs


@pytest.mark.parametrize("tag,Hp,Wp", [("a", 14, 14), ("b", 32, 32), ("c", 16, 12), ("z", 14, 14)])
def test_f4_rvsa_sampling_grid(golden, tag, Hp, Wp):
    g = golden("f4_rvsa_grid.npz")
    p = {k[len("p_%s_" % tag):]: t(v) for k, v in g.items() if k.startswith("p_%s_" % tag)}
    x = t(g["x_" + tag]).reshape(-1, 128)
    B, heads = 2, 2
    ix, iy = O.rvsa_sample_coords(_samp_from(p, x, B, Hp, Wp), B, Hp, Wp, heads)
    geo = I.rvsa_geometry(Hp, Wp)
    He, We = geo["He"], geo["We"]
    grid = t(g["grid_" + tag])                                 # (B*heads, He, We, 2) normalised (x, y)
    assert grid.shape == (B * heads, He, We, 2)
    gx = (grid[..., 0] + 1) * 0.5 * (We - 1)
    gy = (grid[..., 1] + 1) * 0.5 * (He - 1)
    ixm = ix.permute(0, 1, 2, 4, 3, 5).reshape(B * heads, He, We)
    iym = iy.permute(0, 1, 2, 4, 3, 5).reshape(B * heads, He, We)
    assert (ixm - gx).abs().max() < 2e-5 * We and (iym - gy).abs().max() < 2e-5 * He
    if tag == "z":   # zeroed heads: samples hit exact pixel centres
        assert (ixm - torch.arange(We).float()[None, None, :]).abs().max() < 1e-4


def test_f4_rvsa_padded_forward(golden):
    g = golden("f4_rvsa_grid.npz")
    p = {k[len("p_b_"):]: t(v) for k, v in g.items() if k.startswith("p_b_")}
    p.update({k[len("pall_b_"):]: t(v) for k, v in g.items() if k.startswith("pall_b_")})
    x = t(g["x_b"]).reshape(-1, 128)
    qkv = x @ p["qkv.weight"].t() + p["qkv.bias"]
    o, _ = O.rvsa_attn_fwd(qkv, _samp_from(p, x, 2, 32, 32), 2, 32, 32, 2, p["rel_pos_h"], p["rel_pos_w"], p["relative_position_bias_table"])
    y = (o @ p["proj.weight"].t() + p["proj.bias"]).reshape(2, 1024, 128)
    s, v = recipe.summarize(y)
    assert np.abs(v - g["y_b_samples"]).max() < TOL * np.abs(g["y_b_samples"]).max()
    assert abs(s[1] - g["y_b_sum"][1]) < 1e-5 * g["y_b_sum"][1]


@pytest.mark.parametrize("tag,Hp,Wp", [("a", 14, 14), ("c", 16, 12), ("z", 14, 14)])
def test_f5_rvsa_attention(golden, tag, Hp, Wp):
    g = golden("f5_rvsa.npz")
    pre = "p_%s_" % tag
    p = {k[len(pre):]: t(v).requires_grad_(True) for k, v in g.items() if k.startswith(pre)}
    B, heads, C = 2, 2, 128
    x = t(g["x_" + tag]).reshape(-1, C).requires_grad_(True)
    qkv = x @ p["qkv.weight"].t() + p["qkv.bias"]
    samp = _samp_from(p, x, B, Hp, Wp)
    o, lse = O.rvsa_attn_fwd(qkv, samp, B, Hp, Wp, heads, p["rel_pos_h"], p["rel_pos_w"], p["relative_position_bias_table"])
    y = o @ p["proj.weight"].t() + p["proj.bias"]
    assert rel_err(y.reshape(B, -1, C), g["y_" + tag]) < TOL
    names = list(p)
    dy = t(g["dy_" + tag]).reshape(-1, C)
    gr = _grads(y, [x] + [p[n] for n in names], dy)
    assert rel_err(gr[0].reshape(B, -1, C), g["dx_" + tag]) < 5 * TOL
    for n, gg in zip(names, gr[1:]):
        if tag == "z" and "sampling" in n:
            # zeroed heads put every sample exactly on a pixel centre = a kink of bilinear interpolation:
            # the value is continuous but d/d(coord) is one-sided and flips with 1e-7 of coordinate noise.
            continue
        assert rel_err(gg, g["g_%s_%s" % (tag, n)]) < 5 * TOL, n
    do, = _grads(y, [o], dy)
    m = O.rvsa_attn_bwd(do, qkv.detach(), samp.detach(), o.detach(), lse.detach(), B, Hp, Wp, heads,
                        p["rel_pos_h"].detach(), p["rel_pos_w"].detach(), p["relative_position_bias_table"].detach())
    a = _grads(o, [qkv, samp, p["rel_pos_h"], p["rel_pos_w"], p["relative_position_bias_table"]], do)
    for k, (u, v) in enumerate(zip(m, a)):
        if not (tag == "z" and k == 1):
            assert rel_err(u, v) < 5 * TOL


def test_f6_ops(golden):
    g = golden("f6_ops.npz")
    C, heads, B = 128, 2, 2
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(recipe.state_shapes(C, 3, heads, 3)).items()}
    x = t(g["x"]).reshape(-1, C)
    assert rel_err(O.mlp(x, p["blocks.0.mlp.fc1.weight"], p["blocks.0.mlp.fc1.bias"], p["blocks.0.mlp.fc2.weight"], p["blocks.0.mlp.fc2.bias"]),
                   g["mlp"].reshape(-1, C)) < TOL
    for i, tag, window in ((0, "win", True), (2, "full", False)):
        xi = x.clone().requires_grad_(True)
        pre = "blocks.%d." % i
        y = O.block_forward(xi, p, pre, window, B, 14, 14, heads)
        assert rel_err(y, g["block_" + tag].reshape(-1, C)) < TOL
        names = [k for k in p if k.startswith(pre)]
        gr = _grads(y, [xi] + [p[k] for k in names], recipe.loss_weights((B, 196, C), 100 + i).reshape(-1, C))
        assert rel_err(gr[0], g["block_%s_dx" % tag].reshape(-1, C)) < 5 * TOL
        for k, gg in zip(names, gr[1:]):
            assert rel_err(gg, g["block_%s_g_%s" % (tag, k[len(pre):])]) < 5 * TOL, k
    img = recipe.make_input(B, 224, 224, seed=5)
    tok, hw = O.patch_embed(img, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"])
    assert hw == (14, 14) and rel_err(tok, g["patch_embed"].reshape(-1, C)) < TOL
    fm = t(g["fm"])
    tokens = O.nchw_to_tokens(fm, B, 14, 14, 0)
    n2 = O.layernorm_fwd(tokens, p["fpn1.1.ln.weight"], p["fpn1.1.ln.bias"])[0]
    assert rel_err(O.tokens_to_nchw(n2, B, 14, 14, 0), g["norm2d"]) < TOL
    f = O.fpn([tokens] * 4, B, 14, 14, p)
    assert rel_err(f[0], g["fpn1"]) < TOL and rel_err(f[1], g["fpn2"]) < TOL and rel_err(f[3], g["fpn4"]) == 0.0
    assert rel_err(f[2], fm) == 0.0


def _check_summary(tensor, gsum, gsamples, tol, n=512):
    s, v = recipe.summarize(tensor, n)
    scale = np.abs(gsamples).max() + 1e-30
    assert np.abs(v - gsamples).max() < tol * scale
    assert abs(s[0] - gsum[0]) < 50 * tol * gsum[1] and abs(s[1] - gsum[1]) < tol * gsum[1]


def test_f8_small_whole_model_fwd_and_grads(golden):
    g = golden("f8_small.npz")
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(recipe.state_shapes(128, 6, 2, 3)).items()}
    img = recipe.make_input(2, 224, 224, seed=99).requires_grad_(True)
    feats = O.backbone_forward(img, p, 6, 2, 3, [1, 2, 3, 5])
    loss = 0
    for i, f in enumerate(feats):
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], TOL, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 200 + i)).sum()
    assert rel_err(feats[2], g["f2"]) < TOL and rel_err(feats[3], g["f3"]) < TOL
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL, 2048)
    for n, v in p.items():
        if "nograd_" + n in g:
            assert v.grad is None        # encoder.norm is never used (VIT:638)
        elif "g_" + n in g:
            assert rel_err(v.grad, g["g_" + n]) < 5 * TOL, n
        else:
            _check_summary(v.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], 5 * TOL, 1024)


def test_f14_patch_size_8_and_layer_scale_fwd_and_grads(golden):
    """fixture f14 = the reference class with patch_size = 8 (FPN tail ConvT | identity | MaxPool 2 | MaxPool 4, VIT:656-670) and init_values (layer scale
    gamma_1 / gamma_2 on the two residual branches, VIT:500-512): options MTP's factories do not use but the reference constructor accepts"""
    g = golden("f14_patch8_layerscale.npz")
    shapes = recipe.state_shapes(128, 4, 2, 2, 112, patch_size=8, layer_scale=True)
    assert list(shapes) == [str(k) for k in g["keys"]]
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(shapes).items()}
    img = recipe.make_input(2, 112, 112, seed=41).requires_grad_(True)
    feats = O.backbone_forward(img, p, 4, 2, 2, [0, 1, 2, 3])
    loss = 0
    for i, f in enumerate(feats):
        assert f.shape == g["f%d" % i].shape and rel_err(f, g["f%d" % i]) < TOL, i
        loss = loss + (f * recipe.loss_weights(f.shape, 800 + i)).sum()
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL, 2048)
    for n, v in p.items():
        if "nograd_" + n in g:
            assert v.grad is None
        elif "g_" + n in g:
            assert rel_err(v.grad, g["g_" + n]) < 5 * TOL, n
        else:
            _check_summary(v.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], 5 * TOL, 1024)


def test_f9_vitdet_style_finetune_copy_fwd_and_grads(golden):
    """fixture f9 = the reference's mmdet `RVSA_MTP` (RS_Tasks_Finetune/Horizontal_Detection/mmdet/models/backbones/
    vit_rvsa_mtp.py): full attention without rel-pos, last block -> final norm -> fpn1-4 on that one map.  Every parameter
    (norm.* included) has a gradient there."""
    g = golden("f9_vitdet.npz")
    shapes = {k: v for k, v in recipe.state_shapes(128, 4, 2, 2).items() if "full_attn_rel_pos" not in k}
    assert list(shapes) == [str(k) for k in g["keys"]]
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(shapes).items()}
    img = recipe.make_input(2, 224, 224, seed=77).requires_grad_(True)
    feats = O.backbone_forward(img, p, 4, 2, 2, [], vitdet=True)
    loss = 0
    for i, f in enumerate(feats):
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], TOL, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 300 + i)).sum()
    assert rel_err(feats[2], g["f2"]) < TOL and rel_err(feats[3], g["f3"]) < TOL
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL, 2048)
    for n, v in p.items():
        if "g_" + n in g:
            assert rel_err(v.grad, g["g_" + n]) < 5 * TOL, n
        else:
            _check_summary(v.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], 5 * TOL, 1024)


def test_f10_tap_only_finetune_copy_fwd_and_grads(golden):
    """fixture f10 = the reference's mmpretrain `RVSA_MTP`: the taps (here blocks 1 and 3) as NCHW maps, no fpn ops; fpn* and
    norm.* exist in the state dict but receive no gradient."""
    g = golden("f10_taps.npz")
    shapes = recipe.state_shapes(128, 4, 2, 2)
    assert list(shapes) == [str(k) for k in g["keys"]]
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(shapes).items()}
    img = recipe.make_input(2, 224, 224, seed=55).requires_grad_(True)
    feats = O.backbone_forward(img, p, 4, 2, 2, [1, 3], taps_only=True)
    assert len(feats) == 2 and rel_err(feats[0], g["f0"]) < TOL and rel_err(feats[1], g["f1"]) < TOL
    loss = sum((f * recipe.loss_weights(f.shape, 400 + i)).sum() for i, f in enumerate(feats))
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL, 2048)
    for n, v in p.items():
        if "nograd_" + n in g:
            assert v.grad is None and (n.startswith("fpn") or n.startswith("norm."))
        elif "g_" + n in g:
            assert rel_err(v.grad, g["g_" + n]) < 5 * TOL, n
        else:
            _check_summary(v.grad, g["gs_%s_sum" % n], g["gs_%s_samples" % n], 5 * TOL, 1024)


def test_f7_vitb_whole_forward_config1(golden):
    """BASELINE config 1: ViT-B/16 forward, batch 2, 224x224 (+ input/param gradient samples)."""
    g = golden("f7_vitb.npz")
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(recipe.state_shapes(768, 12, 12, 3)).items()}
    img = recipe.make_input(2, 224, 224).requires_grad_(True)
    feats = O.backbone_forward(img, p, 12, 12, 3, [3, 5, 7, 11])
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["f%d_shape" % i])
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], TOL)
    loss = sum(f.mean() for f in feats)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL)
    for k in g:
        if k.startswith("g_") and k.endswith("_samples"):
            n = k[2:-len("_samples")]
            _check_summary(p[n].grad, g["g_%s_sum" % n], g[k], 5 * TOL)
    assert p["norm.weight"].grad is None and not bool(g["norm_has_grad"][0])


def test_f13_vitl_whole_forward_and_gradients(golden):
    """the headline model (BASELINE configs 3 / 4): ViT-L + RVSA from the reference's own factory, batch 2 -- the oracle is pinned at
    the size the benchmark runs (C = 1024, 16 heads, 24 blocks), forward, input gradient and twelve parameter gradients.

    The input is one without a bilinear sample near a cell edge (recipe.F13_INPUT_SEED: the closest of the 501,760 sample
    coordinates is 1.24e-5 px from an edge in the reference's float64 run; make_golden.f13 asserts it).  Round 2's fixture (input
    seed 2023) had a sample 3.3e-6 px from an edge: the oracle's f32 grid put it on the other side than the reference's, picked
    the other one-sided derivative, and everything upstream of that block moved by up to 1.3e-3 -- a property of that input, not
    of an implementation.  With this input every gradient agrees to 5 x TOL."""
    g = golden("f13_vitl.npz")
    p = {k: v.requires_grad_(True) for k, v in recipe.make_params(recipe.state_shapes(1024, 24, 16, 6)).items()}
    assert sum(v.numel() for v in p.values()) == int(g["n_params"][0])
    assert int(g["input_seed"][0]) == recipe.F13_INPUT_SEED and float(g["min_edge_distance_px"][0]) >= recipe.F13_MIN_EDGE_DISTANCE
    img = recipe.make_input(2, 224, 224, seed=recipe.F13_INPUT_SEED).requires_grad_(True)
    feats = O.backbone_forward(img, p, 24, 16, 6, [int(i) for i in g["out_indices"]])
    loss = 0
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["f%d_shape" % i])
        _check_summary(f, g["f%d_sum" % i], g["f%d_samples" % i], TOL, 4096)
        loss = loss + (f * recipe.loss_weights(f.shape, 600 + i)).sum()
    loss.backward()
    _check_summary(img.grad, g["dimg_sum"], g["dimg_samples"], 5 * TOL, 4096)
    for k in g:
        if k.startswith("g_") and k.endswith("_samples"):
            n = k[2:-len("_samples")]
            _check_summary(p[n].grad, g["g_%s_sum" % n], g[k], 5 * TOL, 2048)
    assert p["norm.weight"].grad is None and not bool(g["norm_has_grad"][0])
