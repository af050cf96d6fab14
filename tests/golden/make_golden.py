"""Generate the golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py        # writes tests/golden/*.npz, *.json

The reference (/root/reference) is imported through ref_loader.py; only inputs/outputs (plain arrays)
are stored.  Tests re-create inputs/params from the seeds in recipe.py.  Fixture map:
  f0 state-dict keys/shapes         f1 integer/index ops           f2 calc_rel_pos_spatial
  f3 Attention (full)               f4 RVSA sampling grid          f5 RVSA attention fwd+grads
  f6 Mlp/Block/PatchEmbed/Norm2d/fpn  f7 ViT-B whole forward (cfg 1)  f8 small whole model fwd+grads (fp32 and bf16 autocast)
  f9 / f10 fine-tune variants  f11 DCNv3 core  f12 InternImage (f15: its other layer branches)  f13 ViT-L (the headline model), fwd + grads, fp32 and bf16 autocast
"""
import contextlib
import io
import json
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from ref_loader import load_reference  # noqa: E402
import recipe  # noqa: E402

ref = load_reference()
torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build(embed_dim, depth, heads, interval, out_indices, img_size=224, drop_path_rate=0.0, seed=2023):
    net = quiet(ref.ViT_Win_RVSA_V3_WSZ7, img_size=img_size, patch_size=16, drop_path_rate=drop_path_rate,
                out_indices=out_indices, embed_dim=embed_dim, depth=depth, num_heads=heads, mlp_ratio=4,
                qkv_bias=True, use_abs_pos_emb=True, interval=interval, use_rel_pos_bias=True)
    shapes = recipe.state_shapes(embed_dim, depth, heads, interval, img_size)
    sd = net.state_dict()
    float_keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    assert float_keys == list(shapes.keys()), "state_shapes() order/names differ from the reference"
    for k in float_keys:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    msg = net.load_state_dict(recipe.make_params(shapes, seed), strict=False)
    assert not msg.unexpected_keys and all("relative_position_index" in k for k in msg.missing_keys), msg
    return net, shapes


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------ f0: state dict keys
def f0():
    rec = {}
    for tag, cfg in {"vit_b": (768, 12, 12, 3), "small": (128, 6, 2, 3)}.items():
        net = quiet(ref.ViT_Win_RVSA_V3_WSZ7, img_size=224, embed_dim=cfg[0], depth=cfg[1], num_heads=cfg[2], interval=cfg[3],
                    qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
        rec[tag] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]

    class A:
        image_size = 224
        use_ckpt = "False"
    for tag, fac in (("factory_b", ref.vit_b_rvsa), ("factory_l", ref.vit_l_rvsa)):
        net = quiet(fac, A)
        rec[tag] = dict(n_params=sum(p.numel() for p in net.parameters()), out_indices=list(net.out_indices), interval=net.interval,
                        depth=len(net.blocks), embed_dim=net.embed_dim, out_channels=list(net.out_channels),
                        heads=net.blocks[0].attn.num_heads,
                        window_blocks=[isinstance(b.attn, ref.RotatedVariedSizeWindowAttention) for b in net.blocks],
                        drop_path=[(b.drop_path.drop_prob if isinstance(b.drop_path, ref.DropPath) else 0.0) for b in net.blocks],
                        num_layers=net.get_num_layers(), no_weight_decay=sorted(net.no_weight_decay()),
                        patch_shape=list(net.patch_embed.patch_shape), n_keys=len(net.state_dict()))
        del net
    with open(os.path.join(HERE, "f0_state_keys.json"), "w") as f:
        json.dump(rec, f)
    print("f0_state_keys.json")


# ------------------------------------------------------------------ f1: integer ops
def f1():
    att = quiet(ref.RotatedVariedSizeWindowAttention, 128, 2, window_size=(7, 7))
    out = {"relative_position_index": att.relative_position_index}
    # dist tables via behaviour: table row r carries value r in channel 0, q = e0  => attn == dist index
    for k in (7, 14):
        tab = torch.zeros(2 * k - 1, 4)
        tab[:, 0] = torch.arange(2 * k - 1)
        q = torch.zeros(1, 1, k * k, 4)
        q[..., 0] = 1
        a = ref.calc_rel_pos_spatial(torch.zeros(1, 1, k * k, k * k), q, (k, k), (k, k), tab, torch.zeros_like(tab))
        out["dist_h_%d" % k] = a.reshape(k, k, k, k)[:, 0, :, 0].long()
        a = ref.calc_rel_pos_spatial(torch.zeros(1, 1, k * k, k * k), q, (k, k), (k, k), torch.zeros_like(tab), tab)
        out["dist_w_%d" % k] = a.reshape(k, k, k, k)[0, :, 0, :].long()
    x = torch.arange(2 * 14 * 21 * 3).reshape(2, 14, 21, 3)
    w = ref.window_partition(x, 7)
    out["wp_in"], out["wp_out"] = x, w
    out["wr_out"] = ref.window_reverse(w, 7, 14, 21)
    save("f1_index.npz", **out)


# ------------------------------------------------------------------ f2: calc_rel_pos_spatial
def f2():
    g = torch.Generator().manual_seed(11)
    out = {}
    for k in (7, 14):
        q = torch.randn(2, 3, k * k, 16, generator=g)
        attn = torch.randn(2, 3, k * k, k * k, generator=g)
        rh, rw = torch.randn(2 * k - 1, 16, generator=g), torch.randn(2 * k - 1, 16, generator=g)
        out.update({"q%d" % k: q, "attn%d" % k: attn.clone(), "rh%d" % k: rh, "rw%d" % k: rw,
                    "out%d" % k: ref.calc_rel_pos_spatial(attn.clone(), q, (k, k), (k, k), rh, rw)})
    save("f2_relpos.npz", **out)


# ------------------------------------------------------------------ f3: full attention module
def f3():
    g = torch.Generator().manual_seed(12)
    C, heads, B = 128, 2, 2
    att = ref.Attention(C, num_heads=heads, qkv_bias=True, window_size=(14, 14))
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.ndim == 2 and p.shape[1] == 64 else 0.08))
    x = torch.randn(B, 196, C, generator=g, requires_grad=True)
    y = att(x, 14, 14)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    save("f3_full_attn.npz", x=x, dy=dy, y=y, dx=x.grad,
         **{"p_" + n: p for n, p in att.named_parameters()}, **{"g_" + n: p.grad for n, p in att.named_parameters()})


# ------------------------------------------------------------------ f4/f5: RVSA
def _rvsa_module(C, heads, g, zero_heads=False):
    att = quiet(ref.RotatedVariedSizeWindowAttention, C, heads, window_size=(7, 7), qkv_bias=True)
    with torch.no_grad():
        for n, p in att.named_parameters():
            std = 0.3 if "sampling" in n else (0.2 if ("rel_pos" in n or "table" in n) else 0.08)
            p.copy_(torch.randn(p.shape, generator=g) * std)
            if zero_heads and "sampling" in n:
                p.zero_()
    return att


def f45():
    g = torch.Generator().manual_seed(13)
    C, heads, B = 128, 2, 2
    grids = {}
    real = F.grid_sample

    def spy(inp, grid, **kw):
        grids.setdefault("g", grid.detach().clone())
        return real(inp, grid, **kw)

    out4, out5 = {}, {}
    for tag, (Hp, Wp), zero in (("a", (14, 14), False), ("b", (32, 32), False), ("c", (16, 12), False), ("z", (14, 14), True)):
        att = _rvsa_module(C, heads, g, zero)
        x = torch.randn(B, Hp * Wp, C, generator=g, requires_grad=True)
        grids.clear()
        F.grid_sample = spy
        ref.F.grid_sample = spy
        try:
            y = att(x, Hp, Wp)
        finally:
            F.grid_sample = real
            ref.F.grid_sample = real
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out4["grid_" + tag] = grids["g"]
        out4["x_" + tag] = x
        for n, p in att.named_parameters():
            if "sampling" in n:
                out4["p_%s_%s" % (tag, n)] = p
        if tag != "b":
            out5.update({"x_" + tag: x, "dy_" + tag: dy, "y_" + tag: y, "dx_" + tag: x.grad})
            out5.update({"p_%s_%s" % (tag, n): p for n, p in att.named_parameters()})
            out5.update({"g_%s_%s" % (tag, n): p.grad for n, p in att.named_parameters()})
        else:
            # padded 32->35 case: keep compact (outputs + input grad summary)
            s, v = recipe.summarize(y)
            out4.update({"y_b_sum": s, "y_b_samples": v})
            out4.update({"pall_b_" + n: p for n, p in att.named_parameters() if "sampling" not in n})
    save("f4_rvsa_grid.npz", **out4)
    save("f5_rvsa.npz", **out5)


# ------------------------------------------------------------------ f6: Mlp / Block / PatchEmbed / Norm2d / fpn
def f6():
    C, heads, B = 128, 2, 2
    net, shapes = build(C, 3, heads, 3, [0, 1, 2, 2])   # blocks 0,1 window; block 2 full
    net.eval()
    g = torch.Generator().manual_seed(14)
    out = {}
    x = torch.randn(B, 196, C, generator=g)
    out["x"] = x
    out["mlp"] = net.blocks[0].mlp(x)
    for i, tag in ((0, "win"), (2, "full")):
        xi = x.clone().requires_grad_(True)
        y = net.blocks[i](xi, 14, 14)
        dy = recipe.loss_weights(y.shape, 100 + i)
        net.zero_grad()
        y.backward(dy)
        out["block_%s" % tag] = y
        out["block_%s_dx" % tag] = xi.grad
        for n, p in net.blocks[i].named_parameters():
            out["block_%s_g_%s" % (tag, n)] = p.grad.clone()
    img = recipe.make_input(B, 224, 224, seed=5)
    tok, hw = net.patch_embed(img)
    out["patch_embed"] = tok
    assert hw == (14, 14)
    fm = torch.randn(B, C, 14, 14, generator=g)
    out["fm"] = fm
    out["norm2d"] = net.fpn1[1](fm)
    out["fpn1"], out["fpn2"], out["fpn4"] = net.fpn1(fm), net.fpn2(fm), net.fpn4(fm)
    save("f6_ops.npz", **out)


# ------------------------------------------------------------------ f7: ViT-B whole forward (BASELINE config 1)
def f7():
    net, shapes = build(768, 12, 12, 3, [3, 5, 7, 11])
    net.eval()
    img = recipe.make_input(2, 224, 224).requires_grad_(True)
    feats = net(img)
    out = {}
    for i, f in enumerate(feats):
        out["f%d_sum" % i], out["f%d_samples" % i] = recipe.summarize(f)
        out["f%d_shape" % i] = np.array(f.shape)
    loss = sum(f.mean() for f in feats)
    loss.backward()
    out["loss"] = loss.detach()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad)
    for n in ("pos_embed", "blocks.0.attn.sampling_offsets.2.weight", "blocks.2.attn.full_attn_rel_pos_h", "blocks.5.mlp.fc1.weight",
              "blocks.7.attn.relative_position_bias_table", "blocks.11.attn.qkv.bias", "fpn1.0.weight", "patch_embed.proj.weight"):
        p = dict(net.named_parameters())[n]
        out["g_%s_sum" % n], out["g_%s_samples" % n] = recipe.summarize(p.grad)
    out["norm_has_grad"] = np.array([net.norm.weight.grad is not None])
    save("f7_vitb.npz", **out)


# ------------------------------------------------------------------ f8: small whole model, fwd + all grads, train mode, bf16 autocast
def f8():
    net, shapes = build(128, 6, 2, 3, [1, 2, 3, 5])
    net.train()   # drop_path_rate = 0 -> identical to eval (VIT:495)
    img = recipe.make_input(2, 224, 224, seed=99).requires_grad_(True)
    feats = net(img)
    out = {}
    loss = 0
    for i, f in enumerate(feats):
        out["f%d_sum" % i], out["f%d_samples" % i] = recipe.summarize(f, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 200 + i)).sum()
    out["f2"], out["f3"] = feats[2], feats[3]
    loss.backward()
    out["loss"] = loss.detach()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad, 2048)
    for n, p in net.named_parameters():
        if p.grad is None:
            out["nograd_" + n] = np.array([1])
            continue
        if p.numel() <= 4096:
            out["g_" + n] = p.grad
        else:
            out["gs_%s_sum" % n], out["gs_%s_samples" % n] = recipe.summarize(p.grad, 1024)
    # the reference under bf16 autocast: forward AND gradients (what the bf16 throughput mode is compared with, relative L2)
    net.zero_grad()
    imgb = img.detach().clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        fb = net(imgb)
        lossb = 0
        for i, f in enumerate(fb):
            out["bf16_f%d_sum" % i], out["bf16_f%d_samples" % i] = recipe.summarize(f.float(), 2048)
            lossb = lossb + (f.float() * recipe.loss_weights(f.shape, 200 + i)).sum()
    lossb.backward()
    out["bf16_dimg_sum"], out["bf16_dimg_samples"] = recipe.summarize(imgb.grad, 2048)
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        if p.numel() <= 4096:
            out["bf16_g_" + n] = p.grad.float()
        else:
            out["bf16_gs_%s_sum" % n], out["bf16_gs_%s_samples" % n] = recipe.summarize(p.grad.float(), 1024)
    save("f8_small.npz", **out)


# ------------------------------------------------------------------ f9: ViTDet-style fine-tune copy (mmdet RVSA_MTP), fwd + all grads
def f9():
    """RS_Tasks_Finetune/Horizontal_Detection/mmdet/models/backbones/vit_rvsa_mtp.py: full attention WITHOUT rel-pos, last block
    -> final `norm` -> fpn1-4 all applied to that one map.  Same parameter recipe minus the full_attn_rel_pos_* keys."""
    det = ref_loader.load_reference_det()
    net = quiet(det.RVSA_MTP, img_size=224, patch_size=16, drop_path_rate=0.0, out_indices=[1, 2, 3, 3], embed_dim=128, depth=4,
                num_heads=2, mlp_ratio=4, qkv_bias=True, use_abs_pos_emb=True, interval=2, use_rel_pos_bias=True)
    shapes = {k: v for k, v in recipe.state_shapes(128, 4, 2, 2, 224).items() if "full_attn_rel_pos" not in k}
    sd = net.state_dict()
    float_keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    assert float_keys == list(shapes.keys()), (set(float_keys) ^ set(shapes.keys()))
    msg = net.load_state_dict(recipe.make_params(shapes, 2023), strict=False)
    assert not msg.unexpected_keys and all("relative_position_index" in k for k in msg.missing_keys), msg
    net.train()
    img = recipe.make_input(2, 224, 224, seed=77).requires_grad_(True)
    feats = net(img)
    assert isinstance(feats, tuple) and len(feats) == 4
    out = {"keys": np.array(float_keys)}
    loss = 0
    for i, f in enumerate(feats):
        out["f%d_sum" % i], out["f%d_samples" % i] = recipe.summarize(f, 2048)
        loss = loss + (f * recipe.loss_weights(f.shape, 300 + i)).sum()
    out["f2"], out["f3"] = feats[2], feats[3]
    loss.backward()
    out["loss"] = loss.detach()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad, 2048)
    for n, p in net.named_parameters():
        assert p.grad is not None, n          # every parameter (norm.* included) is used in this variant
        if p.numel() <= 4096:
            out["g_" + n] = p.grad
        else:
            out["gs_%s_sum" % n], out["gs_%s_samples" % n] = recipe.summarize(p.grad, 1024)
    save("f9_vitdet.npz", **out)


# ------------------------------------------------------------------ f10: tap-only fine-tune copy (mmpretrain RVSA_MTP), fwd + grads
def f10():
    """RS_Tasks_Finetune/Scene_Classification/mmpretrain/models/backbones/vit_rvsa_mtp.py: the taps are returned as NCHW maps
    WITHOUT the fpn ops (:838-840 commented out); fpn*/norm parameters exist but are unused.  Two taps here (out_indices [1, 3])."""
    cls = ref_loader.load_reference_cls()
    net = quiet(cls.RVSA_MTP, img_size=224, patch_size=16, drop_path_rate=0.0, out_indices=[1, 3], embed_dim=128, depth=4,
                num_heads=2, mlp_ratio=4, qkv_bias=True, use_abs_pos_emb=True, interval=2, use_rel_pos_bias=True)
    shapes = recipe.state_shapes(128, 4, 2, 2, 224)
    float_keys = [k for k, v in net.state_dict().items() if v.dtype.is_floating_point]
    assert float_keys == list(shapes.keys())
    msg = net.load_state_dict(recipe.make_params(shapes, 2023), strict=False)
    assert not msg.unexpected_keys and all("relative_position_index" in k for k in msg.missing_keys), msg
    net.train()
    img = recipe.make_input(2, 224, 224, seed=55).requires_grad_(True)
    feats = net(img)
    assert isinstance(feats, tuple) and len(feats) == 2 and tuple(feats[0].shape) == (2, 128, 14, 14)
    out = {"keys": np.array(float_keys), "f0": feats[0], "f1": feats[1]}
    loss = 0
    for i, f in enumerate(feats):
        loss = loss + (f * recipe.loss_weights(f.shape, 400 + i)).sum()
    loss.backward()
    out["loss"] = loss.detach()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad, 2048)
    for n, p in net.named_parameters():
        if p.grad is None:
            out["nograd_" + n] = np.array([1])
        elif p.numel() <= 4096:
            out["g_" + n] = p.grad
        else:
            out["gs_%s_sum" % n], out["gs_%s_samples" % n] = recipe.summarize(p.grad, 1024)
    save("f10_taps.npz", **out)


# ------------------------------------------------------------------ f11: DCNv3 core (InternImage), reference = dcnv3_core_pytorch
DCN_CASES = [   # name, N, H, W, group, group_channels, kh, kw, stride, pad, dil, offset_scale, remove_center
    ("base", 2, 8, 8, 4, 16, 3, 3, 1, 1, 1, 2.0, 0),        # ops_dcnv3/test.py:19-30 (its forward / timing case)
    ("bwd4", 2, 8, 8, 2, 4, 3, 3, 1, 1, 1, 2.0, 0),         # test.py:94-100 (backward cases: M = 2, D = channels)
    ("odd", 2, 8, 8, 2, 5, 3, 3, 1, 1, 1, 2.0, 0),
    ("wide", 1, 6, 7, 3, 32, 3, 3, 1, 1, 1, 1.0, 0),
    ("stride2", 2, 9, 8, 4, 16, 3, 3, 2, 1, 1, 2.0, 0),
    ("dil2", 2, 8, 10, 4, 16, 3, 3, 1, 2, 2, 1.5, 0),
    ("nopad", 1, 8, 8, 4, 16, 3, 3, 1, 0, 1, 2.0, 0),
    ("rmc", 2, 8, 8, 4, 16, 3, 3, 1, 1, 1, 2.0, 1),
    ("k5rmc", 1, 9, 9, 2, 16, 5, 5, 1, 2, 1, 1.0, 1),
    ("k1x3", 1, 6, 8, 2, 16, 1, 3, 1, 1, 1, 2.0, 0),
]


def f11():
    """inputs by ops_dcnv3/test.py's recipe (rand*0.01 input, rand*10 offset, normalised positive mask), in float64;
    outputs and gradients of sum(out * G) from the reference's dcnv3_core_pytorch (dcnv3_func.py:168-236)"""
    core = ref_loader.load_reference_dcnv3().dcnv3_core_pytorch
    torch.manual_seed(3)
    out = {"cases": np.array([c[0] for c in DCN_CASES])}
    for name, N, H, W, M, D, kh, kw, st, pad, dil, osc, rmc in DCN_CASES:
        P = kh * kw - rmc
        Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // st + 1
        Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // st + 1
        inp = (torch.rand(N, H, W, M * D, dtype=torch.float64) * 0.01).requires_grad_(True)
        off = (torch.rand(N, Ho, Wo, M * P * 2, dtype=torch.float64) * 10 - (0 if name == "base" else 4)).requires_grad_(True)
        mask = torch.rand(N, Ho, Wo, M, P, dtype=torch.float64) + 1e-5
        mask = (mask / mask.sum(-1, keepdim=True)).reshape(N, Ho, Wo, M * P).requires_grad_(True)
        G = torch.randn(N, Ho, Wo, M * D, dtype=torch.float64)
        y = core(inp, off, mask, kh, kw, st, st, pad, pad, dil, dil, M, D, osc, rmc)
        assert y.shape == G.shape, (name, y.shape, G.shape)
        (y * G).sum().backward()
        out.update({name + ".cfg": np.array([N, H, W, M, D, kh, kw, st, pad, dil, rmc], dtype=np.int64), name + ".offset_scale": np.float64(osc),
                    name + ".input": inp, name + ".offset": off, name + ".mask": mask, name + ".grad_output": G, name + ".output": y,
                    name + ".grad_input": inp.grad, name + ".grad_offset": off.grad, name + ".grad_mask": mask.grad})
    save("f11_dcnv3.npz", **out)


# ------------------------------------------------------------------ f12: InternImage backbone (reference, core_op='DCNv3_pytorch')
def f12():
    from oracle import internimage_oracle as IO
    II = ref_loader.load_reference_internimage()
    cfg = recipe.II_CFG
    net = quiet(II.InternImage, core_op="DCNv3_pytorch", channels=cfg["channels"], depths=cfg["depths"], groups=cfg["groups"], mlp_ratio=4.0,
                drop_path_rate=0.0, norm_layer="LN", layer_scale=cfg["layer_scale"], offset_scale=cfg["offset_scale"], post_norm=True, with_cp=False,
                out_indices=(0, 1, 2, 3))
    sd = net.state_dict()
    shapes = IO.state_shapes(cfg["channels"], cfg["depths"], cfg["groups"])
    assert list(sd.keys()) == list(shapes.keys()), "state_shapes() order/names differ from the reference"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    params = recipe.internimage_params(shapes)
    net.load_state_dict(params, strict=True)
    net = net.double().eval()     # float64 run: what is left against an exact evaluation is dcnv3_core_pytorch's float32 sampling grid
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(12)).double().requires_grad_(True)
    feats = net(img)
    gs = [torch.randn(f.shape, generator=torch.Generator().manual_seed(100 + i)).double() for i, f in enumerate(feats)]
    sum((f * g).sum() for f, g in zip(feats, gs)).backward()
    grads = dict(net.named_parameters())
    keep = ["patch_embed.conv1.weight", "levels.0.blocks.0.dcn.offset.weight", "levels.0.blocks.0.dcn.mask.bias", "levels.0.blocks.0.gamma1",
            "levels.1.blocks.0.dcn.dw_conv.0.weight", "levels.2.blocks.1.dcn.input_proj.weight", "levels.2.blocks.0.norm1.0.weight",
            "levels.0.downsample.conv.weight", "levels.3.blocks.0.mlp.fc2.bias", "levels.3.blocks.0.dcn.output_proj.weight"]
    out = {"keys": np.array(list(sd.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]), "grad_img": img.grad}
    for i, f in enumerate(feats):
        out["feat%d" % i] = f
    for k in keep:
        out["grad." + k] = grads[k].grad
    save("f12_internimage.npz", **out)


# ------------------------------------------------------------------ f15: the other InternImageLayer branches (II:407-427) and block norms (II:497-517)
F15_VARIANTS = recipe.II_VARIANTS


def f15():
    from oracle import internimage_oracle as IO
    II = ref_loader.load_reference_internimage()
    cfg = recipe.II_CFG
    out = {}
    for name, kw in F15_VARIANTS.items():
        net = quiet(II.InternImage, core_op="DCNv3_pytorch", channels=cfg["channels"], depths=cfg["depths"], groups=cfg["groups"], mlp_ratio=4.0,
                    drop_path_rate=0.0, norm_layer="LN", offset_scale=cfg["offset_scale"], with_cp=False, out_indices=(0, 1, 2, 3), **kw)
        sd = net.state_dict()
        shapes = IO.state_shapes(cfg["channels"], cfg["depths"], cfg["groups"], post_norm=kw["post_norm"], layer_scale=kw["layer_scale"] is not None,
                                 res_post_norm=kw.get("res_post_norm", False), level2_post_norm_block_ids=kw.get("level2_post_norm_block_ids"),
                                 dw_kernel_size=kw.get("dw_kernel_size"), center_feature_scale=kw.get("center_feature_scale", False))
        assert list(sd.keys()) == list(shapes.keys()), "state_shapes() order/names differ from the reference (%s)" % name
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k
        net.load_state_dict(recipe.internimage_variant_params(shapes), strict=True)
        net = net.double().eval()
        img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(recipe.II_VARIANT_SEEDS[name])).double().requires_grad_(True)
        feats = net(img)
        gs = [torch.randn(f.shape, generator=torch.Generator().manual_seed(200 + i)).double() for i, f in enumerate(feats)]
        sum((f * g).sum() for f, g in zip(feats, gs)).backward()
        grads = dict(net.named_parameters())
        keep = ["patch_embed.conv1.weight", "levels.0.blocks.0.dcn.offset.weight", "levels.0.blocks.0.norm1.0.weight", "levels.1.blocks.0.dcn.dw_conv.0.weight",
                "levels.2.blocks.0.norm2.0.bias", "levels.2.blocks.1.dcn.input_proj.weight", "levels.0.downsample.conv.weight", "levels.3.blocks.0.mlp.fc2.bias",
                "levels.3.blocks.0.dcn.output_proj.weight"]
        keep += [k for k in ("levels.0.blocks.0.gamma1", "levels.2.blocks.1.gamma2", "levels.1.norm.0.weight", "levels.3.norm.0.bias", "levels.2.post_norms.0.0.weight",
                             "levels.2.blocks.0.res_post_norm1.0.weight", "levels.3.blocks.0.res_post_norm2.0.bias",
                             "levels.0.blocks.0.dcn.center_feature_scale_proj_weight", "levels.2.blocks.1.dcn.center_feature_scale_proj_bias") if k in grads]
        out[name + ".keys"] = np.array(list(sd.keys()))
        out[name + ".shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[name + ".grad_img"] = img.grad.float()           # (stored as float32: the float64 run's own error against an exact evaluation is 1e-5, see f12)
        for i, f in enumerate(feats):
            out[name + ".feat%d" % i] = f.float()
        for k in keep:
            out[name + ".grad." + k] = grads[k].grad.float()
    save("f15_internimage_variants.npz", **out)


# ------------------------------------------------------------------ f13: ViT-L (BASELINE configs 3/4: the headline model), fp32 + bf16 autocast
F13_GRADS = ("pos_embed", "patch_embed.proj.weight", "blocks.0.attn.sampling_offsets.2.weight", "blocks.3.attn.relative_position_bias_table",
             "blocks.5.attn.full_attn_rel_pos_h", "blocks.11.mlp.fc1.weight", "blocks.17.attn.qkv.weight", "blocks.23.attn.proj.bias",
             "blocks.22.attn.sampling_angles.2.weight", "blocks.20.attn.rel_pos_w", "fpn1.3.weight", "fpn2.0.bias")


def f13():
    class A:
        image_size = 224
        use_ckpt = "False"
    net = quiet(ref.vit_l_rvsa, A)                     # the factory itself (VIT:843-865): 1024 / 24 / 16 heads, interval 6, taps 7 11 15 23
    shapes = recipe.state_shapes(1024, 24, 16, 6)
    sd = net.state_dict()
    assert [k for k, v in sd.items() if v.dtype.is_floating_point] == list(shapes.keys())
    net.load_state_dict(recipe.make_params(shapes), strict=False)
    net.eval()                                         # drop_path 0.1 is inactive in eval
    # no bilinear sample of this input within 1e-5 px of a cell edge (recipe.F13_INPUT_SEED; float64 run of the reference)
    import copy
    import find_f13_seed
    dist, ncoord = find_f13_seed.edge_distance(copy.deepcopy(net).double(), recipe.make_input(2, 224, 224, seed=recipe.F13_INPUT_SEED).double())
    assert dist >= recipe.F13_MIN_EDGE_DISTANCE, "input seed %d has a sample %.3e px from a cell edge" % (recipe.F13_INPUT_SEED, dist)
    print("f13: closest of %d sample coordinates is %.3e px from a cell edge" % (ncoord, dist))
    out = {"out_indices": np.array(net.out_indices), "n_params": np.array([sum(p.numel() for p in net.parameters())]),
           "input_seed": np.array([recipe.F13_INPUT_SEED]), "min_edge_distance_px": np.array([dist])}
    P = dict(net.named_parameters())
    for tag, ctx in (("", contextlib.nullcontext()), ("bf16_", torch.autocast("cpu", dtype=torch.bfloat16))):
        net.zero_grad()
        img = recipe.make_input(2, 224, 224, seed=recipe.F13_INPUT_SEED).requires_grad_(True)
        with ctx:
            feats = net(img)
            loss = 0
            for i, f in enumerate(feats):
                out[tag + "f%d_sum" % i], out[tag + "f%d_samples" % i] = recipe.summarize(f.float(), 4096)
                out[tag + "f%d_shape" % i] = np.array(f.shape)
                loss = loss + (f.float() * recipe.loss_weights(f.shape, 600 + i)).sum()
        loss.backward()
        out[tag + "loss"] = loss.detach()
        out[tag + "dimg_sum"], out[tag + "dimg_samples"] = recipe.summarize(img.grad, 4096)
        for n in F13_GRADS:
            out[tag + "g_%s_sum" % n], out[tag + "g_%s_samples" % n] = recipe.summarize(P[n].grad.float(), 2048)
        print("f13", tag or "fp32", "loss", float(loss))
    out["norm_has_grad"] = np.array([net.norm.weight.grad is not None])
    save("f13_vitl.npz", **out)


def f13_hard():
    """The same model on round 2's input (seed 2023), fp32 only: ONE of its ~6 M bilinear samples sits 3.3e-6 px from a cell edge, where the
    interpolation's derivative jumps -- any f32 implementation picks one of the two one-sided derivatives by the last bit of the sample
    position, and every gradient upstream of that RVSA block moves by up to ~1e-3 with it.  Kept as a second case with the looser gradient
    bound that goes with it (the near-edge behaviour of the kernels stays covered); f13 proper is the kink-free seed."""
    class A:
        image_size = 224
        use_ckpt = "False"
    net = quiet(ref.vit_l_rvsa, A)
    shapes = recipe.state_shapes(1024, 24, 16, 6)
    net.load_state_dict(recipe.make_params(shapes), strict=False)
    net.eval()
    import copy
    import find_f13_seed
    dist, ncoord = find_f13_seed.edge_distance(copy.deepcopy(net).double(), recipe.make_input(2, 224, 224, seed=recipe.F13_HARD_INPUT_SEED).double())
    print("f13_hard: closest of %d sample coordinates is %.3e px from a cell edge" % (ncoord, dist))
    out = {"input_seed": np.array([recipe.F13_HARD_INPUT_SEED]), "min_edge_distance_px": np.array([dist])}
    P = dict(net.named_parameters())
    img = recipe.make_input(2, 224, 224, seed=recipe.F13_HARD_INPUT_SEED).requires_grad_(True)
    feats = net(img)
    loss = 0
    for i, f in enumerate(feats):
        out["f%d_sum" % i], out["f%d_samples" % i] = recipe.summarize(f.float(), 4096)
        loss = loss + (f.float() * recipe.loss_weights(f.shape, 600 + i)).sum()
    loss.backward()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad, 4096)
    for n in F13_GRADS:
        out["g_%s_sum" % n], out["g_%s_samples" % n] = recipe.summarize(P[n].grad.float(), 2048)
    save("f13_vitl_hard.npz", **out)


# ------------------------------------------------------------------ f14: the two remaining constructor options in one small model
def f14():
    """patch_size = 8 (VIT:656-670: FPN tail = ConvT | identity | MaxPool 2 | MaxPool 4) and init_values (layer scale gamma_1 / gamma_2, VIT:500-512) -- options
    MTP's two factories do not use but the reference class accepts.  112 x 112 input -> 14 x 14 tokens; forward + every gradient, and the state-dict key order."""
    net = quiet(ref.ViT_Win_RVSA_V3_WSZ7, img_size=112, patch_size=8, drop_path_rate=0.0, out_indices=[0, 1, 2, 3], embed_dim=128, depth=4, num_heads=2,
                mlp_ratio=4, qkv_bias=True, use_abs_pos_emb=True, interval=2, use_rel_pos_bias=True, init_values=0.1)
    shapes = recipe.state_shapes(128, 4, 2, 2, 112, patch_size=8, layer_scale=True)
    sd = net.state_dict()
    float_keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    assert float_keys == list(shapes.keys()), (float_keys[:8], list(shapes.keys())[:8])
    for k in float_keys:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    assert abs(float(sd["blocks.0.gamma_1"][0]) - 0.1) < 1e-7          # init_values * ones
    msg = net.load_state_dict(recipe.make_params(shapes, 2023), strict=False)
    assert not msg.unexpected_keys and all("relative_position_index" in k for k in msg.missing_keys), msg
    net.train()
    img = recipe.make_input(2, 112, 112, seed=41).requires_grad_(True)
    feats = net(img)
    assert [tuple(f.shape) for f in feats] == [(2, 128, 28, 28), (2, 128, 14, 14), (2, 128, 7, 7), (2, 128, 3, 3)]
    out = {"keys": np.array(float_keys)}
    loss = 0
    for i, f in enumerate(feats):
        out["f%d" % i] = f
        loss = loss + (f * recipe.loss_weights(f.shape, 800 + i)).sum()
    loss.backward()
    out["dimg_sum"], out["dimg_samples"] = recipe.summarize(img.grad, 2048)
    for n, p in net.named_parameters():
        if p.grad is None:
            out["nograd_" + n] = np.array([1])
        elif p.numel() <= 4096:
            out["g_" + n] = p.grad
        else:
            out["gs_%s_sum" % n], out["gs_%s_samples" % n] = recipe.summarize(p.grad, 1024)
    save("f14_patch8_layerscale.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f0", "f1", "f2", "f3", "f45", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f13_hard", "f14", "f15"]
    for w in which:
        globals()[w]()
