"""CPU: checkpoint / resume in the reference's formats (main_pretrain.py:483-499 resume, :823-829 save) and the parameter
groups of LayerDecayOptimizerConstructor_ViT (mmcv_custom/layer_decay_optimizer_constructor_vit.py:33-67)."""
import math

import numpy as np
import pytest
import torch

import mtp_amd
from mtp_amd.parallel import DataParallelTrainer, FlatAdamW, FlatParams, reference_param_groups


def small():
    torch.manual_seed(0)
    return mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])


def torch_adamw_like_reference(net, lr=6e-5, wd=0.05):
    """what build_optim_wrapper(model, LayerDecayOptimizerConstructor_ViT) hands to torch.optim.AdamW for the backbone"""
    P = dict(net.named_parameters())
    groups = [{"params": [P[n] for n in names], "weight_decay": gwd, "lr": lr * scale, "param_names": names, "lr_scale": scale, "group_name": g}
              for g, scale, gwd, names in reference_param_groups(net.named_parameters(), wd)]
    return torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), weight_decay=wd)


def test_param_groups_follow_the_reference_rule():
    net = small()
    groups = reference_param_groups(net.named_parameters(), 0.05)
    # "encoder.*" never matches get_num_layer_for_vit's "backbone.*" tests: every parameter gets layer id num_layers - 1 = depth + 1,
    # lr_scale 0.9 ** 0 = 1, so exactly two groups, in first-seen order (pos_embed comes first -> no_decay first)
    assert [g[0] for g in groups] == ["layer_7_no_decay", "layer_7_decay"] and all(g[1] == 1.0 for g in groups)
    nd, d = set(groups[0][3]), set(groups[1][3])
    assert groups[0][2] == 0.0 and groups[1][2] == 0.05
    assert "pos_embed" in nd and "blocks.0.norm1.weight" in nd and "blocks.0.attn.qkv.bias" in nd and "blocks.0.attn.sampling_offsets.2.bias" in nd
    assert "blocks.0.attn.qkv.weight" in d and "blocks.0.attn.rel_pos_h" in d and "blocks.0.attn.relative_position_bias_table" in d
    assert "patch_embed.proj.weight" in d and "fpn1.0.weight" in d
    assert nd | d == {n for n, _ in net.named_parameters()} and not (nd & d)
    # the flat optimizer's per-segment weight decay is the same rule
    flat = FlatParams(net, unused=net._unused_params)
    starts, wds = flat.weight_decay_segments(0.05)
    for n, wd in zip(flat.names, wds.tolist()):
        assert (wd == 0.0) == (n in nd), n


def test_optimizer_state_round_trips_with_torch_adamw():
    net = small()
    flat = FlatParams(net, unused=net._unused_params)
    opt = FlatAdamW(flat, total_steps=100)
    g = torch.Generator().manual_seed(1)
    opt.m.copy_(torch.randn(opt.m.shape, generator=g))
    opt.v.copy_(torch.rand(opt.v.shape, generator=g))
    opt.t = opt.last_epoch = 7
    sd = opt.state_dict(net)
    ref = torch_adamw_like_reference(net)
    ref.load_state_dict(sd)                                   # torch validates group sizes / ids
    P = dict(net.named_parameters())
    for n in ("blocks.2.attn.qkv.weight", "pos_embed", "fpn1.0.bias", "blocks.5.mlp.fc2.bias"):
        st = ref.state[P[n]]
        assert torch.equal(st["exp_avg"], flat.view(opt.m, n)) and torch.equal(st["exp_avg_sq"], flat.view(opt.v, n)) and float(st["step"]) == 7
    assert P["norm.weight"] not in ref.state                  # never gets a gradient (VIT:638): no state, as in torch
    assert sd["param_groups"][0]["lr"] == pytest.approx(opt.lr_at(7)) and sd["param_groups"][1]["weight_decay"] == 0.05
    # and back: a state dict produced by torch's AdamW loads into the flat optimizer
    opt2 = FlatAdamW(FlatParams(small(), unused=net._unused_params), total_steps=100)
    opt2.load_state_dict(ref.state_dict(), net)
    used = torch.zeros_like(opt.m, dtype=torch.bool)
    for n in flat.names:
        if flat.groups[n] is not None:
            flat.view(used, n).fill_(True)
    assert opt2.t == 7 and torch.equal(opt2.m[used], opt.m[used]) and torch.equal(opt2.v[used], opt.v[used])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref, 100, eta_min=0, last_epoch=-1)
    ssd = opt.scheduler_state_dict()
    assert set(ssd) >= {"T_max", "eta_min", "base_lrs", "last_epoch"}
    sched.load_state_dict(ssd)
    assert sched.last_epoch == 7 and sched.get_last_lr()[0] == pytest.approx(opt.lr_at(7))


def test_trainer_checkpoint_in_reference_format(tmp_path):
    net = small()
    tr = DataParallelTrainer(net, total_steps=50)
    tr.opt.t = tr.opt.last_epoch = 3
    tr.opt.m.fill_(0.25)
    tr.opt.v.fill_(0.5)
    path = tmp_path / "Iter_3_vit_l_rvsa_pretrn_model_encoder.pth"
    tr.save_checkpoint(str(path), epoch=1, losses=[1.0, 0.5])
    ck = torch.load(str(path), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "iteration", "state_dict", "optimizer", "scheduler", "loss_pretrain"}      # MAIN:826
    assert list(ck["state_dict"]) == list(small().state_dict()) and isinstance(ck["loss_pretrain"], np.ndarray) and ck["iteration"] == 3
    # resume into a fresh trainer (MAIN:483-499): weights, moments, step counter, schedule
    net2 = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
    with torch.no_grad():
        for p in net2.parameters():
            p.add_(1.0)
    tr2 = DataParallelTrainer(net2, total_steps=10)
    ck["state_dict"]["decoder.not_ours.weight"] = torch.zeros(3)       # foreign keys are skipped like MAIN:490-492
    epoch, it, losses = tr2.load_checkpoint(ck)
    assert (epoch, it, losses) == (1, 3, [1.0, 0.5]) and tr2.opt.t == 3 and tr2.opt.total_steps == 50
    for (n, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), n
    assert net2.pos_embed.data_ptr() == tr2.flat.view(tr2.flat.data, "pos_embed").data_ptr()     # still views of the flat buffer
    n = "blocks.1.mlp.fc1.weight"
    assert float(tr2.flat.view(tr2.opt.m, n).mean()) == 0.25 and float(tr2.flat.view(tr2.opt.v, n).mean()) == 0.5
    # ckpt['state_dict'] is the encoder's own state dict: loads strictly into a fresh backbone.  (The pretrain-side
    # init_weights(), VIT:744-768, always strips one cls token from pos_embed -- it is written for MAE checkpoints -- so, as in
    # the reference, it is not the way to resume from these files.)
    net3 = small()
    with torch.no_grad():
        net3.pos_embed.zero_()
    assert not any(net3.load_state_dict(ck["state_dict"] if "decoder.not_ours.weight" not in ck["state_dict"] else
                                        {k: v for k, v in ck["state_dict"].items() if k != "decoder.not_ours.weight"}, strict=True))
    assert torch.equal(net3.pos_embed, net.pos_embed)


def test_resume_from_a_whole_model_reference_checkpoint():
    """the reference's *_encoder.pth holds optimizer.state_dict() of the WHOLE pretrain model -- encoder + three decoders, names
    prefixed `encoder.` / `semsegdecoder.` ... (MAIN:826-829) -- written after optimizer.step() but BEFORE scheduler.step()
    (MAIN:788, 823, 832): Adam step = iteration, scheduler last_epoch = iteration - 1.  The backbone's entries are found by
    name, the decoders' skipped, and the two counters stay apart."""
    net = small()

    class Whole(torch.nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.encoder = enc
            self.semsegdecoder = torch.nn.Conv2d(8, 4, 3)       # stands for the mm* decoders: parameters this build does not own
            self.rotdetdecoder = torch.nn.Linear(16, 5)
    whole = Whole(net)
    # the reference's constructor: groups in first-seen order of named_parameters(), `param_names` stored per group
    groups = {}
    for n, p in whole.named_parameters():
        nd = p.dim() == 1 or n.endswith(".bias") or "pos_embed" in n
        g = groups.setdefault("no_decay" if nd else "decay", {"params": [], "param_names": [], "weight_decay": 0.0 if nd else 0.05, "lr_scale": 1.0})
        g["params"].append(p)
        g["param_names"].append(n)
    ref = torch.optim.AdamW(list(groups.values()), lr=6e-5, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref, 100, eta_min=0)
    gen = torch.Generator().manual_seed(3)
    for it in range(1, 4):                                        # three iterations of MAIN's loop
        for p in whole.parameters():
            p.grad = torch.randn(p.shape, generator=gen) * 1e-2
        ref.step()
        if it == 3:                                               # checkpoint of iteration 3: saved before the scheduler steps
            ck = {"epoch": 0, "iteration": it, "state_dict": {k: v.clone() for k, v in net.state_dict().items()},
                  "optimizer": ref.state_dict(), "scheduler": sched.state_dict(), "loss_pretrain": np.array([1.0])}
        sched.step()
    assert len(ck["optimizer"]["state"]) > len(list(net.parameters()))           # really the whole model's optimizer
    tr = DataParallelTrainer(small(), total_steps=100)
    tr.load_checkpoint(ck)
    assert tr.opt.t == 3 and tr.opt.last_epoch == 2                               # Adam step 3, scheduler epoch 2
    P = dict(whole.encoder.named_parameters())
    for n in ("blocks.2.attn.qkv.weight", "pos_embed", "fpn1.0.bias", "blocks.4.attn.sampling_angles.2.weight"):
        st = ref.state[P[n]]
        assert torch.equal(tr.flat.view(tr.opt.m, n), st["exp_avg"]) and torch.equal(tr.flat.view(tr.opt.v, n), st["exp_avg_sq"])
    # the next step uses the learning rate the reference's resumed loop would use (its scheduler is still at epoch 2)
    assert tr.opt.hyper_values()[0] == pytest.approx(0.5 * 6e-5 * (1 + math.cos(math.pi * 2 / 100)))


def test_checkpoint_scheduler_fields_match_the_reference_save_point():
    """MAIN:788, 823-832: optimizer.step() -> save -> scheduler.step().  After N trainer steps the file must hold Adam step N and the
    scheduler of epoch N - 1: FlatAdamW.step() leaves its scheduler step pending until the next step."""
    net = small()
    tr = DataParallelTrainer(net, total_steps=50)
    opt = tr.opt
    # what step() does around the kernels, three times (the kernels themselves need the GPU)
    for _ in range(3):
        opt.scheduler_step()
        opt.t += 1
        lr_used = opt.hyper_values()[0]
        opt.sched_pending = True
    assert opt.t == 3 and opt.last_epoch == 2 and lr_used == pytest.approx(opt.lr_at(2))
    ck = tr.checkpoint()
    ref = torch_adamw_like_reference(net)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref, 50, eta_min=0)
    for _ in range(2):          # the reference's scheduler at the save point of iteration 3 has stepped twice
        ref.step()
        sched.step()
    want = sched.state_dict()
    assert ck["iteration"] == 3 and ck["scheduler"]["last_epoch"] == want["last_epoch"] == 2
    assert ck["scheduler"]["_step_count"] == want["_step_count"] and ck["scheduler"]["_last_lr"][0] == pytest.approx(want["_last_lr"][0])
    assert ck["optimizer"]["param_groups"][0]["lr"] == pytest.approx(ref.param_groups[0]["lr"])
    # a resumed trainer continues like the reference's resumed loop: no pending scheduler step
    tr2 = DataParallelTrainer(small(), total_steps=50)
    tr2.opt.sched_pending = True
    tr2.load_checkpoint(ck)
    assert tr2.opt.last_epoch == 2 and tr2.opt.sched_pending is False


def test_resume_with_foreign_optimizer_names_fails_loudly():
    """optimizer state whose names match nothing must not silently restart Adam at a late-schedule learning rate"""
    net = small()
    tr = DataParallelTrainer(net, total_steps=50)
    tr.opt.t = tr.opt.last_epoch = 5
    tr.opt.m.fill_(0.1)
    ck = tr.checkpoint()
    for pg in ck["optimizer"]["param_groups"]:
        pg["param_names"] = ["wrapper.model." + n for n in pg["param_names"]]
    tr2 = DataParallelTrainer(small(), total_steps=50)
    with pytest.raises(ValueError, match="none of them matches"):
        tr2.load_checkpoint(ck)
    # a partial match warns and reports the count
    ck = tr.checkpoint()
    first = ck["optimizer"]["param_groups"][0]
    first["param_names"] = ["wrapper." + n if i % 2 else n for i, n in enumerate(first["param_names"])]
    with pytest.warns(UserWarning, match="optimizer state restored for"):
        tr2.load_checkpoint(ck)
    assert 0 < tr2.restored_optimizer_entries < len([n for n in tr2.flat.names if tr2.flat.groups[n] is not None])
