"""CPU, world_size 2 / 3 / 8, gloo: the N > 1 path of mtp_amd.parallel -- flat reverse-execution-order layout, bucket partition,
bucketed all-reduce == one big all-reduce, unused parameters excluded statically, optimizer schedule/segments."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mtp_amd
from mtp_amd.parallel import ALIGN, FlatAdamW, FlatParams, GradReducer, execution_order


def small():
    torch.manual_seed(0)
    return mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])


def test_execution_order_and_layout():
    net = small()
    ref = {n: p.detach().clone() for n, p in net.named_parameters()}
    flat = FlatParams(net, unused=net._unused_params)
    order, groups = execution_order([n for n, _ in net.named_parameters()], 6)
    assert groups["norm.weight"] == groups["fpn1.0.weight"] == 6      # tail group (the final norm is used by RVSA_MTP_det only)
    assert flat.names[0].startswith("fpn") and flat.names[-1].startswith("norm.")   # ... and dropped here through `unused`
    det = mtp_amd.RVSA_MTP_det(embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True, out_indices=[3])
    fdet = FlatParams(det, unused=det._unused_params)
    assert fdet.groups["norm.weight"] == 4 and "norm.weight" in fdet.G and fdet.offsets["norm.bias"] < fdet.reduced
    assert not any("full_attn_rel_pos" in n for n in fdet.names)
    gids = [flat.groups[n] for n in flat.names if flat.groups[n] is not None]
    assert gids == sorted(gids, reverse=True)                      # FPN (6), blocks 5..0, embed (-1)
    assert all(flat.groups[n] is None for n in ("norm.weight", "norm.bias"))
    for n, p in net.named_parameters():                            # parameters are now views of the flat buffer, values kept
        assert torch.equal(p.data, ref[n]) and flat.offsets[n] % ALIGN == 0
        assert p.data.data_ptr() == flat.data.data_ptr() + 4 * flat.offsets[n]
    assert flat.offsets["norm.weight"] >= flat.reduced              # never all-reduced / never stepped
    assert set(flat.G) == set(ref) - {"norm.weight", "norm.bias"}
    for bb in (1, 1 << 20, 1 << 40):
        bk = flat.buckets(bb)
        assert bk[0][1] == 0 and bk[-1][2] == flat.reduced and bk[-1][0] == -1
        assert all(a[2] == b[1] for a, b in zip(bk, bk[1:]))        # disjoint cover of [0, reduced)
    assert len(flat.buckets(1)) == 8 and len(flat.buckets(1 << 40)) == 1
    st, wd = flat.weight_decay_segments(0.05)
    by = dict(zip(flat.names, wd.tolist()))
    assert by["pos_embed"] == 0 and by["blocks.0.attn.qkv.bias"] == 0 and by["blocks.0.norm1.weight"] == 0
    assert by["blocks.0.attn.qkv.weight"] == pytest.approx(0.05) and by["blocks.0.attn.rel_pos_h"] == pytest.approx(0.05)
    assert st.tolist() == sorted(st.tolist()) and all(s % 4 == 0 for s in st.tolist())


def test_rest_table_of_the_gradient_norm_by_product_is_the_complement_of_the_covered_tensors():
    """FlatAdamW._rest_table (round 6): the weight-gradient launches account for the norm of the tensors they write (`covered`); the table lists exactly the other
    runs of [0, reduced) in pieces of at most 8192 floats; tensors that are not slices of the flat gradient buffer, or overlap, make it decline (None)"""
    net = small()
    flat = FlatParams(net, unused=net._unused_params)
    opt = FlatAdamW(flat, total_steps=10)
    names = [n for n in flat.names if flat.groups[n] is not None and n.endswith(".weight") and len(flat.shapes[n]) == 2]
    covered = [flat.G[n] for n in names]
    tab = opt._rest_table(covered)
    start, count = tab
    assert int(count.max()) <= 8192 and int(count.min()) > 0
    mask = torch.zeros(flat.reduced, dtype=torch.int32)
    for a, c in zip(start.tolist(), count.tolist()):
        mask[a:a + c] += 1
    for t in covered:
        off = (t.data_ptr() - flat.grad.data_ptr()) // 4
        mask[off:off + t.numel()] += 1
    assert bool((mask == 1).all())                                   # a partition of the reduced range: nothing twice, nothing missing
    assert opt._rest_table(covered) is tab                           # cached by the covered ranges
    assert opt._rest_table(covered + [torch.zeros(8)]) is None       # a tensor outside the flat buffer
    assert opt._rest_table(covered + [covered[0]]) is None           # overlapping tensors
    assert opt._rest_table([flat.grad[:flat.reduced]]) == ()         # everything covered: nothing left to sum


def test_cosine_schedule_and_bias_correction():
    class F:
        pass
    net = small()
    flat = FlatParams(net, unused=net._unused_params)
    opt = FlatAdamW.__new__(FlatAdamW)
    opt.lr0, opt.betas, opt.eps, opt.total_steps, opt.t, opt.last_epoch = 6e-5, (0.9, 0.999), 1e-8, 100, 0, 0
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=6e-5), T_max=100)
    for t in range(1, 6):
        opt.t, opt.last_epoch = t, t - 1          # the t-th Adam step runs with the learning rate of scheduler epoch t - 1
        hv = opt.hyper_values()
        assert hv[0] == pytest.approx(sched.get_last_lr()[0], rel=1e-9)
        assert hv[4] == pytest.approx(1 - 0.9 ** t) and hv[5] == pytest.approx(1 - 0.999 ** t)
        sched.optimizer.step()
        sched.step()
    assert flat.total >= flat.reduced > 0 and math.isfinite(opt.lr_at(1000))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = small()
        flat = FlatParams(net, unused=net._unused_params)
        g = torch.Generator().manual_seed(100 + rank)
        flat.grad.copy_(torch.randn(flat.total, generator=g))
        local = flat.grad.clone()
        ref = local.clone()
        dist.all_reduce(ref)                                   # the single-tensor reference
        # world 2: one addition per element, commutative -> the bucketed sums are bit-identical to the single all-reduce's; beyond that gloo's
        # ring adds each chunk in an order that depends on its position in the buffer: equal to f32 rounding only (measured 2.9e-6 at world 8)
        def same(x, y):
            return torch.equal(x, y) if world == 2 else torch.allclose(x, y, rtol=1e-5, atol=1e-5)
        red = GradReducer(flat, bucket_bytes=1 << 20)
        assert red.world == world and len(red.buckets) > 2
        order = [6] + list(range(5, -1, -1)) + [-1]             # the engine's completion order, one group at a time
        for gid in order:
            red.on_block_done(gid)
        red.finish()
        ok = same(flat.grad[:flat.reduced], ref[:flat.reduced]) and torch.equal(flat.grad[flat.reduced:], local[flat.reduced:])
        ok = ok and red.collectives == len(red.buckets)
        # ... and in bursts, as the engine reports them when the weight gradients of several blocks are launched together
        flat.grad.copy_(local)
        red.begin_step()
        for gid in (6, 5, 4, 3):
            red.on_block_done(gid)          # (all at once after the burst's launch)
        n_after_burst = red.collectives
        for gid in (2, 1, 0, -1):
            red.on_block_done(gid)
        red.finish()
        ok = ok and same(flat.grad[:flat.reduced], ref[:flat.reduced]) and torch.equal(flat.grad[flat.reduced:], local[flat.reduced:])
        ok = ok and n_after_burst >= 1 and red.start == flat.reduced
        # ... and as reduce-scatter + all-gather of the same buckets (the direct form for xGMI): bit for bit the all-reduce's sums
        flat.grad.copy_(local)
        rs = GradReducer(flat, bucket_bytes=1 << 20, mode="rs_ag")
        for gid in order:
            rs.on_block_done(gid)
        rs.finish()
        ok = ok and same(flat.grad[:flat.reduced], ref[:flat.reduced]) and torch.equal(flat.grad[flat.reduced:], local[flat.reduced:])
        ok = ok and rs.wire_bytes == flat.reduced * 4
        # ... and with bf16 on the wire: the sum of the bf16-rounded local gradients, rounded to bf16
        flat.grad.copy_(local)
        others = [torch.randn(flat.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        hb = GradReducer(flat, bucket_bytes=1 << 20, mode="rs_ag", bf16=True)
        for gid in order:
            hb.on_block_done(gid)
        hb.finish()
        if world == 2:      # one addition: bf16 + bf16 -> bf16 exactly, as gloo / RCCL reduce in the wire dtype
            want = (others[0].bfloat16() + others[1].bfloat16()).float()
            ok = ok and torch.equal(flat.grad[:flat.reduced], want[:flat.reduced])
        else:               # world - 1 bf16 additions in the backend's order: every partial sum is rounded to 8 bits of mantissa
            want = sum(o.bfloat16().float() for o in others)
            err = (flat.grad[:flat.reduced] - want[:flat.reduced]).abs()
            bound = 2.0 ** -8 * (world - 1) * sum(o.abs() for o in others)[:flat.reduced]
            ok = ok and bool((err <= bound + 1e-6).all()) and float(err.mean()) < 0.05
            ok = ok and torch.equal(flat.grad[:flat.reduced], flat.grad[:flat.reduced].bfloat16().float())      # values came back through bf16
        ok = ok and torch.equal(flat.grad[flat.reduced:], local[flat.reduced:])
        ok = ok and hb.wire_bytes == flat.reduced * 2 and hb.bytes_reduced == flat.reduced * 4
        # every rank holds the same averaged-gradient bits after each mode (the optimizer step must not drift the replicas apart)
        chk = torch.tensor([float(flat.grad[:flat.reduced].double().sum())], dtype=torch.float64)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        ok = ok and all(torch.equal(c, allc[0]) for c in allc)
        # a world size that does not divide the bucket lengths: rs_ag must fall back to the all-reduce bucket by bucket, same sums
        if flat.reduced % world or any((b[2] - b[1]) % world for b in rs.buckets):
            ok = ok and rs.collectives < 2 * len(rs.buckets)
        else:
            ok = ok and rs.collectives == 2 * len(rs.buckets)
        q.put((rank, bool(ok), red.bytes_reduced == flat.reduced * 4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])      # 8 = the node BASELINE's headline names; 3 does not divide the bucket lengths (rs_ag falls back per bucket)
def test_bucketed_allreduce_equals_single_allreduce(world):
    from conftest import spawn_ranks
    res = spawn_ranks(_worker, world, deadline=300)
    assert sorted(res) == [(r, True, True) for r in range(world)]


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mtp_amd.parallel import DataParallelTrainer
        torch.manual_seed(1000 + rank)                          # every rank initialises DIFFERENTLY ...
        net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
        tr = DataParallelTrainer(net, total_steps=20)           # ... the trainer broadcasts rank 0's replica (DDP does, MAIN:508-518)
        a = tr.flat.data.clone()
        dist.broadcast(a, src=0)
        same_params = torch.equal(a, tr.flat.data)
        # resume: only rank 0 "read the checkpoint"; optimizer state and both counters must follow
        if rank == 0:
            tr.opt.m.fill_(0.5); tr.opt.v.fill_(0.25); tr.opt.t, tr.opt.last_epoch = 9, 8
        tr.sync_replicas(optimizer_state=True)
        ok = same_params and float(tr.opt.m.mean()) == 0.5 and float(tr.opt.v.mean()) == 0.25 and (tr.opt.t, tr.opt.last_epoch) == (9, 8)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_trainer_synchronises_replicas(world):
    from conftest import spawn_ranks
    res = spawn_ranks(_sync_worker, world, deadline=300)
    assert sorted(res) == [(r, True) for r in range(world)]
