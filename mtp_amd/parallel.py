"""Data-parallel training of the backbone hot path: one process per GPU, RCCL over xGMI through torch.distributed.

Replaces the reference's DistributedDataParallel wrap (main_pretrain.py:508-518: 25 MB buckets, find_unused_parameters=True)
with an explicit schedule built for 8 x MI355X (7 xGMI links per GPU, 288 GB HBM):

  * FlatParams: every parameter lives in ONE f32 buffer, laid out in REVERSE execution order (FPN tail, block depth-1 .. 0,
    patch/pos embed), so the gradients completed by each stretch of the backward are one contiguous slice -> few, large
    collectives instead of ~50 small buckets; parameters that never get a gradient (`norm.*`, VIT:638) sit after the
    reduced range, which replaces find_unused_parameters' per-step graph walk;
  * GradReducer: as soon as the engine reports a bucket's last block done (BackboneEngine.backward(on_block_done=...)),
    an event is recorded on the compute stream and the bucket's all-reduce (SUM, then 1/world folded into the optimizer)
    is issued on a side HIP stream, overlapping the rest of the backward;
  * FlatAdamW: clip_grad_norm_(5) + AdamW (main_pretrain.py:424-457, 783-788) as two HBM-bound kernels over the flat buffers,
    with the reference's no-decay rule (mmcv_custom/layer_decay_optimizer_constructor_vit.py:43-48).

All of it also runs on CPU tensors with the gloo backend (no kernels involved) -- that is how tests/test_parallel_gloo.py
covers the world_size > 1 path in the build container.
"""
import math

import torch
import torch.distributed as dist

ALIGN = 64   # elements (256 B): every parameter starts on its own cache lines; AdamW segments stay 4-element aligned


def execution_order(names, depth):
    """Reverse execution order of the parameter names; returns (ordered names, group id per name) with groups
    depth+0 = FPN tail, block i = i, -1 = patch/pos embed, None = never receives a gradient."""
    def group(n):
        if n.startswith("fpn"):
            return depth
        if n.startswith("blocks."):
            return int(n.split(".")[1])
        if n.startswith("patch_embed") or n == "pos_embed":
            return -1
        return None
    used = [n for n in names if group(n) is not None]
    unused = [n for n in names if group(n) is None]
    used.sort(key=lambda n: -group(n))          # stable: keeps the module's order inside a group
    return used + unused, {n: group(n) for n in names}


class FlatParams:
    def __init__(self, module, unused=()):
        params = dict(module.named_parameters())
        depth = len(module.blocks)
        order, groups = execution_order(list(params), depth)
        for n in unused:
            groups[n] = None
        order = [n for n in order if groups[n] is not None] + [n for n in order if groups[n] is None]
        self.names, self.groups, self.offsets, self.shapes = order, groups, {}, {}
        off = 0
        for n in order:
            self.offsets[n] = off
            self.shapes[n] = tuple(params[n].shape)
            off += (params[n].numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        used = [n for n in order if groups[n] is not None]
        last = used[-1]
        self.reduced = self.offsets[last] + (params[last].numel() + ALIGN - 1) // ALIGN * ALIGN   # [0, reduced) is all-reduced
        dev = next(iter(params.values())).device
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for n in order:
            p = params[n]
            v = self.view(self.data, n)
            v.copy_(p.data)
            p.data = v
            p.grad = None
        self.G = {n: self.view(self.grad, n) for n in order if groups[n] is not None}
        self.depth = depth

    def view(self, flat, n):
        o = self.offsets[n]
        numel = 1
        for s in self.shapes[n]:
            numel *= s
        return flat[o:o + numel].view(self.shapes[n])

    def group_end(self, gid):
        """end offset (exclusive, aligned) of the last parameter of group gid."""
        ns = [n for n in self.names if self.groups[n] == gid]
        n = ns[-1]
        numel = self.view(self.data, n).numel()
        return self.offsets[n] + (numel + ALIGN - 1) // ALIGN * ALIGN

    def buckets(self, bucket_bytes=256 << 20):
        """[(last_group_id, start, end)] in completion order; a bucket closes when it reaches bucket_bytes."""
        out, start = [], 0
        gids = [self.depth] + list(range(self.depth - 1, -1, -1)) + [-1]
        gids = [g for g in gids if any(self.groups[n] == g for n in self.names)]
        for g in gids:
            end = self.group_end(g)
            if (end - start) * 4 >= bucket_bytes or g == gids[-1]:
                out.append((g, start, end))
                start = end
        return out

    def weight_decay_segments(self, weight_decay, no_decay=("pos_embed", "cls_token")):
        """per-parameter segments (start offsets, wd) following the reference's rule: 1-D params, biases, pos_embed -> 0."""
        starts, wds = [], []
        for n in self.names:
            nd = len(self.shapes[n]) == 1 or n.endswith(".bias") or n in no_decay
            starts.append(self.offsets[n])
            wds.append(0.0 if nd else weight_decay)
        return torch.tensor(starts, dtype=torch.int64), torch.tensor(wds, dtype=torch.float32)


class GradReducer:
    """Bucketed gradient all-reduce overlapped with the backward (side stream on GPU; synchronous on CPU/gloo)."""

    def __init__(self, flat, bucket_bytes=256 << 20, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = flat.buckets(bucket_bytes)
        self.by_gid = {g: (s, e) for g, s, e in self.buckets}
        self.cuda = flat.grad.is_cuda
        # MTP_FORCE_COMM=1: issue the collectives even at world size 1 (exercises the RCCL + side-stream path on a 1-GPU box)
        import os
        self.active = self.world > 1 or (os.environ.get("MTP_FORCE_COMM") == "1" and dist.is_available() and dist.is_initialized())
        self.stream = torch.cuda.Stream() if self.cuda and self.active else None
        self.works = []
        self.bytes_reduced = 0

    def on_block_done(self, gid):
        """engine hook: gradients of group `gid` (and everything before it in completion order) are on the compute stream."""
        if not self.active or gid not in self.by_gid:
            return
        s, e = self.by_gid[gid]
        buf = self.flat.grad[s:e]
        self.bytes_reduced += (e - s) * 4
        if self.stream is None:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """make the compute stream (or the host, on CPU) wait for every outstanding bucket."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for w in self.works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self.works:
                w.wait()
        self.works = []


class FlatAdamW:
    """torch.optim.AdamW semantics over FlatParams (lr 6e-5, betas (0.9, 0.999), wd 0.05 in the reference) with the
    cosine schedule of main_pretrain.py:832 and clip_grad_norm_(max_norm=5) of main_pretrain.py:786."""

    def __init__(self, flat, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=5.0, total_steps=None, world_size=1):
        self.flat, self.lr0, self.betas, self.eps, self.max_norm = flat, lr, betas, eps, max_norm
        self.total_steps, self.world = total_steps, world_size
        dev = flat.data.device
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        st, wd = flat.weight_decay_segments(weight_decay)
        self.seg_start, self.seg_wd = st.to(dev), wd.to(dev)
        self.hyper = torch.zeros(6, device=dev, dtype=torch.float32)
        self.sqn = torch.zeros(1, device=dev, dtype=torch.float32)
        self.t = 0

    def lr_at(self, t):
        if not self.total_steps:
            return self.lr0
        return 0.5 * self.lr0 * (1.0 + math.cos(math.pi * min(t, self.total_steps) / self.total_steps))   # CosineAnnealingLR, eta_min 0

    def hyper_values(self):
        b1, b2 = self.betas
        return [self.lr_at(self.t - 1), b1, b2, self.eps, 1.0 - b1 ** self.t, 1.0 - b2 ** self.t]

    def step(self):
        from . import ops
        self.t += 1
        self.hyper.copy_(torch.tensor(self.hyper_values(), dtype=torch.float32), non_blocking=True)
        f = self.flat
        n = f.reduced
        gs = 1.0 / self.world
        sq = None
        if self.max_norm and self.max_norm > 0:
            self.sqn.zero_()
            ops.sqnorm(f.grad[:n], self.sqn)
            sq = self.sqn
        ops.adamw_flat(f.data[:n], f.grad[:n], self.m[:n], self.v[:n], self.seg_start, self.seg_wd, self.hyper, sq, float(self.max_norm or 0.0), gs)


class DataParallelTrainer:
    """fwd -> loss -> bwd (+ overlapped bucketed all-reduce) -> clip + AdamW, on the HIP engine."""

    def __init__(self, module, lr=6e-5, weight_decay=0.05, max_norm=5.0, total_steps=None, bucket_bytes=256 << 20, feature_dtype=None):
        self.module = module
        self.engine = module._engine()
        self.flat = FlatParams(module, unused=module._unused_params)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.reducer = GradReducer(self.flat, bucket_bytes)
        self.opt = FlatAdamW(self.flat, lr=lr, weight_decay=weight_decay, max_norm=max_norm, total_steps=total_steps, world_size=self.world)
        self.feature_dtype = feature_dtype

    def step(self, img, loss_and_grads):
        """loss_and_grads(feats) -> (loss, [dfeat or None] * 4).  Returns the (local) loss tensor."""
        self.flat.grad.zero_()     # one memset per step: the engine accumulates bias / LayerNorm gradients (engine._colsum)
        feats, ctx = self.engine.forward(img, training=True, need_grad=True, feature_dtype=self.feature_dtype)
        loss, dfeats = loss_and_grads(feats)
        self.engine.backward(ctx, dfeats, self.flat.G, on_block_done=self.reducer.on_block_done)
        self.reducer.finish()
        self.opt.step()
        self.engine._key = None   # parameters changed under torch's version counters: rebuild the GEMM weight images next forward
        return loss
