"""GPU: the data-parallel step on the real RCCL path (torch.distributed backend "nccl" = RCCL): side-stream all-reduce gated by
events, replica broadcast, dynamic buckets driven by the engine's bursts.  One rank with MTP_FORCE_COMM=1 runs on any MI355X box
(the collectives are really issued, on the side stream); the two-rank variant needs two GPUs and is skipped otherwise."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(seed):
    import mtp_amd
    torch.manual_seed(seed)
    return mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=256, depth=8, num_heads=4, interval=4, qkv_bias=True, use_abs_pos_emb=True,
                                       out_indices=[1, 3, 5, 7], drop_path_rate=0.0, precision="bf16")


def _loss(feats):
    loss = sum(f.float().mean() for f in feats)
    return loss, [torch.full_like(f, 1.0 / f.numel()) for f in feats]


def _worker(rank, world, port, q, native=False, mode="allreduce", bf16=False):
    import torch.distributed as dist
    from mtp_amd.parallel import DataParallelTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        dev = torch.device("cuda", rank)
        g = torch.Generator().manual_seed(7)
        imgs = torch.randn(2 * world, 3, 224, 224, generator=g)
        # ---- reference: ONE process computes the whole batch, no communication
        os.environ["MTP_FORCE_COMM"] = "0"
        ref = DataParallelTrainer(_net(0).to(dev), total_steps=10, bucket_bytes=1 << 20)
        ref.reducer.active = False
        ref.opt.world = 1
        ref.step(imgs.to(dev), _loss)
        gref = ref.flat.grad.clone()
        # ---- the data-parallel step: this rank's shard, replicas seeded differently on purpose (the trainer broadcasts rank 0's)
        os.environ["MTP_FORCE_COMM"] = "1"
        if native is None:       # the default: the C-ABI communicator (round 6)
            os.environ.pop("MTP_NATIVE_COMM", None)
            native = True
        elif native == "fail":   # the communicator cannot be created: every rank agrees on torch.distributed's collectives, and says so
            os.environ["MTP_NATIVE_COMM"] = "fail"
            native = False
        else:
            os.environ["MTP_NATIVE_COMM"] = "1" if native else "0"
        import warnings
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            tr = DataParallelTrainer(_net(0 if rank == 0 else 123).to(dev), total_steps=10, bucket_bytes=1 << 20, comm_mode=mode, comm_bf16=bf16)
        if os.environ.get("MTP_NATIVE_COMM") == "fail":
            d = tr.reducer.describe()
            assert d["native_c_abi"] is False and "injected failure" in d["native_error"] and any("mtp_comm_init failed" in str(w.message) for w in wlist)
        assert tr.reducer.active and tr.reducer.stream is not None and (tr.reducer.native is not None) == native
        if native:      # what RCCL itself says about the communicator behind mtp_comm_* (mtp_comm_info)
            info = tr.reducer.native.info()
            assert info["nranks"] == world and info["rank"] == rank and info["device"] == rank and info["version_code"] > 0, info
            assert tr.reducer.describe()["communicator"]["nranks"] == world
        assert tr.reducer.mode == mode and tr.reducer.bf16 == bf16
        shard = imgs[2 * rank:2 * rank + 2].to(dev)
        tr.step(shard, _loss)
        torch.cuda.synchronize()
        red = tr.reducer
        # SUM over ranks of the shard-mean gradients = world x the whole-batch-mean gradient (the optimizer folds 1/world in)
        got, want = tr.flat.grad[:tr.flat.reduced] / world, gref[:tr.flat.reduced]
        err = float((got - want).abs().max() / want.abs().max())
        same_params = float((tr.flat.data - ref.flat.data).abs().max())
        q.put((rank, err, same_params, red.collectives, red.bytes_reduced == tr.flat.reduced * 4))
    finally:
        dist.destroy_process_group()


def _run(world, native=False, mode="allreduce", bf16=False):
    from conftest import spawn_ranks
    return sorted(spawn_ranks(_worker, world, (native, mode, bf16)))


def test_forced_comm_single_rank_rccl_side_stream():
    """world size 1 with MTP_FORCE_COMM=1: every bucket really goes through ncclAllReduce on the side stream, gated by events; the
    gradients must equal the run without communication (up to the f32-atomic summation order of the RVSA scatter, ~1e-7: two
    runs of the SAME configuration differ by as much) and the updated parameters likewise"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1)
    assert err < 1e-5 and dparam < 1e-6 and ncoll >= 2 and all_bytes


def test_c_abi_communicator_is_the_default_exchange():
    """no MTP_NATIVE_COMM in the environment: GradReducer exchanges through mtp_comm_* (the entry points the header advertises), torch.distributed only boots"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1, native=None)
    assert err < 1e-5 and dparam < 1e-6 and ncoll >= 2 and all_bytes


def test_forced_comm_single_rank_through_the_c_abi_communicator():
    """the same with MTP_NATIVE_COMM=1: the buckets go through mtp_comm_allreduce_bucket (ncclAllReduce via the C ABI, communicator from
    mtp_comm_unique_id / mtp_comm_init) on the side stream"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1, native=True)
    assert err < 1e-5 and dparam < 1e-6 and ncoll >= 2 and all_bytes


def test_failed_c_abi_communicator_falls_back_to_torch_distributed_on_every_rank():
    """mtp_comm_init fails (injected): the ranks agree through one MIN all-reduce, the exchange runs on torch.distributed's collectives on the same side stream,
    describe() carries the reason and a warning is issued; MTP_NATIVE_COMM=strict would raise instead"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1, native="fail")
    assert err < 1e-5 and dparam < 1e-6 and ncoll >= 2 and all_bytes


@pytest.mark.parametrize("native", [False, True])
def test_forced_comm_single_rank_reduce_scatter_all_gather(native):
    """mode rs_ag on one rank: every bucket goes through ncclReduceScatter + ncclAllGather (torch.distributed, or mtp_comm_* via the C ABI)
    on the side stream; same gradients as without communication"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1, native=native, mode="rs_ag")
    assert err < 1e-5 and dparam < 1e-6 and ncoll >= 4 and all_bytes


def test_forced_comm_single_rank_bf16_buckets():
    """bf16 on the wire: cast (mtp_cast on the side stream) -> collective -> cast back; on one rank the gradients come back rounded to bf16"""
    (rank, err, dparam, ncoll, all_bytes), = _run(1, mode="rs_ag", bf16=True)
    assert err < 1e-2 and ncoll >= 4 and all_bytes


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("native,mode", [(False, "allreduce"), (True, "allreduce"), (False, "rs_ag"), (True, "rs_ag")])
def test_two_ranks_match_single_process_whole_batch(native, mode):
    """2 x MI355X: each rank computes half the batch; all-reduced gradients / 2 == the whole-batch gradients of one process (bf16
    rounding differs between a batch of 4 and two batches of 2 only through accumulation order: 2e-2), replicas end up identical"""
    res = _run(2, native=native, mode=mode)
    for rank, err, dparam, ncoll, all_bytes in res:
        assert err < 2e-2 and ncoll >= 2 and all_bytes


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_fused_adamw_writes_the_weight_images_it_would_otherwise_need_a_second_pass_for(precision, monkeypatch):
    """mtp_adamw_weight_images (round 6): ONE launch = AdamW + clipping over the whole flat buffer AND the GEMM-side weight images.  Against the two-pass form
    (mtp_adamw_flat, then mtp_weight_images at the next forward): parameters, both moments and every image bit-identical after three steps, and the images the
    fused launch left behind equal to what a refresh from the updated masters writes."""
    import mtp_amd
    from mtp_amd.parallel import DataParallelTrainer
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3)).cuda()

    def run(fused):
        monkeypatch.setenv("MTP_FUSED_ADAMW", "1" if fused else "0")
        torch.manual_seed(5)
        net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=128, depth=4, num_heads=2, interval=2, qkv_bias=True, use_abs_pos_emb=True,
                                           out_indices=[0, 1, 2, 3], drop_path_rate=0.0, precision=precision)
        with torch.no_grad():
            for n, q in net.named_parameters():
                if "rel_pos" in n:
                    q.normal_(0, 0.02)
        tr = DataParallelTrainer(net.cuda().train(), lr=1e-3, total_steps=10)
        for _ in range(3):
            tr.step(img, _loss)
        torch.cuda.synchronize()
        assert (tr.opt._fused is not None) == fused
        eng = tr.engine
        left = [(None if w is None else w.clone(), None if wt is None else wt.clone()) for _, w, wt, _ in eng._wimg.entries]      # as the last step left them
        if not fused:
            eng._wimg.refresh()
            left = [(None if w is None else w.clone(), None if wt is None else wt.clone()) for _, w, wt, _ in eng._wimg.entries]
        else:
            assert eng._images_fresh is not None
            eng._wimg.refresh()          # what the image pass makes of the updated masters
            for (w0, t0), (_, w, wt, _) in zip(left, eng._wimg.entries):
                assert (w0 is None or torch.equal(w0, w)) and (t0 is None or torch.equal(t0, wt))
            # a parameter edited through torch after the fused step: the next forward must NOT trust the optimizer's images
            wq = eng._blk[0].wqkv.clone()
            with torch.no_grad():
                tr.module.blocks[0].attn.qkv.weight.mul_(2.0)
            tr.module(img)
            assert torch.equal(eng._blk[0].wqkv.float(), wq.float() * 2.0)
            with torch.no_grad():
                tr.module.blocks[0].attn.qkv.weight.mul_(0.5)      # (exact: back to the same bits)
            eng.prepare_weights(force=True)
            # the same update by both kernels from the same state: bit-identical parameters and moments
            from mtp_amd import ops
            f, o = tr.flat, tr.opt
            state = (f.data.clone(), o.m.clone(), o.v.clone(), left)
            n = f.reduced
            pc, mc, vc = f.data.clone(), o.m.clone(), o.v.clone()
            ops.adamw_flat(pc[:n], f.grad[:n], mc[:n], vc[:n], o.seg_start, o.seg_wd, o.hyper, o.sqn, 5.0, 1.0)
            o._fused.step(o.m, o.v, o.hyper, o.sqn, 5.0, 1.0)
            torch.cuda.synchronize()
            assert torch.equal(pc[:n], f.data[:n]) and torch.equal(mc[:n], o.m[:n]) and torch.equal(vc[:n], o.v[:n])
            assert torch.equal(pc[n:], f.data[n:])          # parameters outside the reduced range (norm.*) are not touched
            return state
        return tr.flat.data.clone(), tr.opt.m.clone(), tr.opt.v.clone(), left
    a, b = run(False), run(True)
    # (two runs of the same configuration differ in the last bits: f32-atomic summation order in a few gradient by-products)
    for x, y in zip(a[:3], b[:3]):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())


def test_gradient_norm_as_a_by_product_of_the_weight_gradient_launches(monkeypatch):
    """round 6: on one rank the grouped weight-gradient launches add sum(dW^2) of what they write to the optimizer's accumulator (mtp_gemm_args.workspace) and the
    clipping step sums only the rest of the flat buffer (mtp_sqnorm_segments_f32) -- the same norm as one pass over the whole buffer, the same update."""
    import mtp_amd
    from mtp_amd.parallel import DataParallelTrainer
    img = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(3)).cuda()      # 6272 token rows: the grouped kernel takes the problems

    def run(fold):
        monkeypatch.setenv("MTP_FUSED_SQNORM", "1" if fold else "0")
        torch.manual_seed(5)
        net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=256, depth=4, num_heads=4, interval=2, qkv_bias=True, use_abs_pos_emb=True,
                                           out_indices=[0, 1, 2, 3], drop_path_rate=0.0, precision="bf16")
        tr = DataParallelTrainer(net.cuda().train(), lr=1e-3, total_steps=10, max_norm=0.05)      # (a norm bound that clips: the coefficient matters)
        for _ in range(2):
            tr.step(img, _loss)
        torch.cuda.synchronize()
        n = tr.flat.reduced
        want = float((tr.flat.grad[:n].double() ** 2).sum())
        got = float(tr.opt.sqn.item())
        cov = list(tr.engine.norm_covered)
        return want, got, cov, None, tr
    w0, g0, c0, p0, _ = run(False)
    w1, g1, c1, p1, tr = run(True)
    assert abs(g0 - w0) <= 1e-5 * w0 and abs(g1 - w1) <= 1e-5 * w1
    assert len(c1) >= 4 * 4 + 1 and len(c0) == 0                          # the blocks' four Linear weights + the patch embedding (+ unsplit FPN weights)
    covered = sum(t.numel() for t in c1)
    assert covered > 0.8 * tr.flat.reduced
    assert tr.opt._rest_tab is not None and int(tr.opt._rest_tab[1].sum()) == tr.flat.reduced - covered      # the rest table is the complement
    # (the parameters of two runs are not compared: Adam's first steps are +- lr per element, and the f32-atomic by-products flip the sign of near-zero gradients)


def test_gradient_norm_by_product_on_internimage():
    """the same by-product through InternEngine's weight-gradient queue (padded Linear layers and the 3x3 convolutions hand the norm of their re-laid images):
    the optimizer's accumulator equals the squared norm of the whole gradient buffer"""
    import mtp_amd
    import recipe
    from mtp_amd.parallel import DataParallelTrainer
    c = recipe.II_CFG
    torch.manual_seed(3)
    net = mtp_amd.InternImage(channels=c["channels"], depths=c["depths"], groups=c["groups"], layer_scale=c["layer_scale"], offset_scale=c["offset_scale"],
                              post_norm=True, drop_path_rate=0.0, precision="bf16")
    with torch.no_grad():
        for n, q in net.named_parameters():
            if ".dcn.offset.weight" in n or ".dcn.mask.weight" in n:
                q.normal_(0, 0.02)
    tr = DataParallelTrainer(net.cuda().train(), lr=1e-4, total_steps=10, max_norm=0.05)
    img = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(4)).cuda()

    def lg(feats):
        loss = sum(f.float().mean() for f in feats)
        return loss, [torch.full_like(f, 1.0 / f.numel()) for f in feats]
    tr.step(img, lg)
    tr.step(img, lg)
    torch.cuda.synchronize()
    n = tr.flat.reduced
    want = float((tr.flat.grad[:n].double() ** 2).sum())
    got = float(tr.opt.sqn.item())
    assert len(tr.engine.norm_covered) > 0 and abs(got - want) <= 1e-5 * want, (len(tr.engine.norm_covered), got, want)
