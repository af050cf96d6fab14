"""Data-parallel training of the backbone hot path: one process per GPU, RCCL over xGMI through torch.distributed.

Replaces the reference's DistributedDataParallel wrap (main_pretrain.py:508-518: 25 MB buckets, find_unused_parameters=True)
with an explicit schedule built for 8 x MI355X (7 xGMI links per GPU, 288 GB HBM):

  * FlatParams: every parameter lives in ONE f32 buffer, laid out in REVERSE execution order (FPN tail, block depth-1 .. 0,
    patch/pos embed), so the gradients completed by each stretch of the backward are one contiguous slice -> few, large
    collectives instead of ~50 small buckets; parameters that never get a gradient (`norm.*`, VIT:638) sit after the
    reduced range, which replaces find_unused_parameters' per-step graph walk;
  * GradReducer: as soon as the engine reports a stretch of the backward done (BackboneEngine.backward(on_block_done=...): the
    FPN tail, then bursts of four blocks whose weight gradients were launched together), an event is recorded on the compute
    stream and the all-reduce of that contiguous slice (SUM, 1/world folded into the optimizer) is issued on a side HIP stream,
    overlapping the rest of the backward;
  * FlatAdamW: clip_grad_norm_(5) + AdamW (main_pretrain.py:424-457, 783-788) as two HBM-bound kernels over the flat buffers,
    with the reference's no-decay rule (mmcv_custom/layer_decay_optimizer_constructor_vit.py:43-48).

All of it also runs on CPU tensors with the gloo backend (no kernels involved) -- that is how tests/test_parallel_gloo.py
covers the world_size > 1 path in the build container.
"""
import math
import os

import torch
import torch.distributed as dist

ALIGN = 64   # elements (256 B): every parameter starts on its own cache lines; AdamW segments stay 4-element aligned


def execution_order(names, depth):
    """Reverse execution order of the parameter names; returns (ordered names, group id per name) with groups
    depth+0 = FPN tail, block i = i, -1 = patch/pos embed, None = never receives a gradient."""
    def group(n):
        if n.startswith("fpn") or n.startswith("norm."):   # norm.*: used only by the ViTDet-style copies (final norm before the
            return depth                                    # fpn ops); FlatParams drops it through `unused` for the others
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
        if hasattr(module, "_flat_param_order"):        # backbones with another layer structure (InternImage) supply their own grouping
            order, groups, depth = module._flat_param_order()
        else:
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
        # Gradients the engine OVERWRITES in every backward (the blocks' Linear weights and the patch embedding: TN GEMM outputs, ~97 % of
        # the buffer) need no clearing; everything else accumulates (bias by-products, LayerNorm / table partial sums, the stacked
        # sampling heads; the FPN weights, which a loss that ignores a map leaves unwritten) and is cleared by ONE launch over a table of
        # runs of at most 64 K floats (zero_accumulating()).
        over = getattr(module, "_overwritten_grads", None)
        self._zero_tab = None
        if over is not None and self.grad.is_cuda:
            runs = []
            for n in order:
                if groups[n] is None or over(n):
                    continue
                a, c = self.offsets[n], params[n].numel()
                if runs and runs[-1][0] + runs[-1][1] == a:
                    runs[-1][1] += c
                else:
                    runs.append([a, c])
            ent = [(a + o, min(65536, c - o)) for a, c in runs for o in range(0, c, 65536)]
            if ent:
                self._zero_tab = (torch.tensor([e[0] for e in ent], dtype=torch.int64, device=dev), torch.tensor([e[1] for e in ent], dtype=torch.int64, device=dev))

    def zero_accumulating(self):
        """clear what the backward accumulates into (everything, when the module does not say which gradients it overwrites)"""
        if self._zero_tab is None:
            self.grad.zero_()
        else:
            from . import ops
            ops.zero_segments(self.grad, *self._zero_tab)

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


def reference_param_groups(named_params, weight_decay, prefix="encoder."):
    """Parameter groups as LayerDecayOptimizerConstructor_ViT.add_params builds them
    (mmcv_custom/layer_decay_optimizer_constructor_vit.py:33-67): group = (layer id, decay / no_decay), created in
    first-seen order of named_parameters(); no_decay = 1-D, `.bias`, or 'pos_embed' in the name (:43-48).
    Faithful quirk: get_num_layer_for_vit (:7-16) tests for names starting with "backbone.", but in the pretrain model the
    backbone's parameters are called "encoder.*" (MODELS:85-89), so EVERY parameter falls through to the last layer id
    (num_layers - 1) and lr_scale = 0.9 ** 0 = 1: the layer decay is a no-op there, and there are exactly two groups.
    Returns [(group_name, lr_scale, weight_decay, [names])], group_name = "layer_<depth+1>_<decay|no_decay>" as in :50."""
    named_params = list(named_params)
    depth = 1 + max([int(n.split(".")[1]) for n, _ in named_params if n.startswith("blocks.")] or [-1])
    groups, order = {}, []
    for n, p in named_params:
        if not p.requires_grad:
            continue
        full = prefix + n
        nd = p.dim() == 1 or full.endswith(".bias") or "pos_embed" in full
        key = "layer_%d_%s" % (depth + 1, "no_decay" if nd else "decay")
        if key not in groups:
            groups[key] = (key, 1.0, 0.0 if nd else weight_decay, [])
            order.append(key)
        groups[key][3].append(n)
    return [groups[k] for k in order]


class GradReducer:
    """Gradient exchange overlapped with the backward (side stream on GPU; synchronous on CPU / gloo).

    The engine reports completed groups in completion order (BackboneEngine.backward(on_block_done=...): FPN tail, then the
    blocks -- in bursts, because the weight gradients of several blocks are launched together (ops.WgradQueue: 4 ViT-L blocks =
    ~200 MB of gradients per burst) -- then the embeddings).  Everything between the previous cut and the end of the reported
    group is contiguous in the flat buffer: it becomes ONE collective as soon as it is at least `bucket_bytes` long (or the
    last group arrived).  With the default 64 MB that is one ~200 MB exchange per burst of blocks, issued while the next four
    blocks' backward (~4.5 ms) runs; the engine cuts the last burst short (BackboneEngine.backward(split_last=True)) so that only
    block 0 + the embeddings (~50 MB) are exposed.

    mode   "allreduce": one SUM all-reduce per bucket (what DistributedDataParallel does, MAIN:508-518);
           "rs_ag": reduce-scatter + all-gather of the same bucket, in place -- the direct form for xGMI's point-to-point links
           (every GPU owns 1/world of the bucket; 2.1 ms vs 14.5 ms for a ring over 1.27 GB at 8 GPUs).  Same sums.
    bf16   exchange the bucket as bf16: cast on the side stream into a scratch bucket, collective, cast back into the f32 gradient
           buffer (half the xGMI bytes; the sum over ranks is then rounded to bf16 -- off by default).
    A bucket whose length is not a multiple of the world size falls back to the all-reduce (buckets end on 64-element boundaries,
    so this only happens for world sizes that do not divide 64)."""

    def __init__(self, flat, bucket_bytes=64 << 20, group=None, mode=None, bf16=None):
        self.flat, self.group, self.bucket_bytes = flat, group, bucket_bytes
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.buckets = flat.buckets(bucket_bytes)       # the partition one-group-at-a-time reporting produces (tests, DESIGN)
        self.last_gid = self.buckets[-1][0]
        self.cuda = flat.grad.is_cuda
        self.mode = mode or os.environ.get("MTP_COMM_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError("GradReducer mode must be 'allreduce' or 'rs_ag', got %r" % (self.mode,))
        self.bf16 = (os.environ.get("MTP_COMM_BF16") == "1") if bf16 is None else bool(bf16)
        # MTP_FORCE_COMM=1: issue the collectives even at world size 1 (exercises the RCCL + side-stream path on a 1-GPU box)
        self.active = self.world > 1 or (os.environ.get("MTP_FORCE_COMM") == "1" and dist.is_available() and dist.is_initialized())
        self.stream = torch.cuda.Stream() if self.cuda and self.active else None
        # The collectives go through the C ABI (mtp_comm_* -> ncclAllReduce / ncclReduceScatter / ncclAllGather on the side stream): the exchange entry
        # points include/mtp_hip.h advertises are the ones every GPU run exercises (round 6; torch.distributed is the bootstrap: rendezvous, the
        # 128-byte id, barriers).  MTP_NATIVE_COMM=0: torch.distributed's collectives instead (same RCCL underneath).
        self.native, self.native_error = None, None
        if self.stream is not None and os.environ.get("MTP_NATIVE_COMM", "1") != "0":
            from .comm import RcclComm
            # (the exchange stream takes its hardware queue at first use: use it BEFORE RCCL creates the communicator's own streams -- BackboneEngine.warm_streams)
            with torch.cuda.stream(self.stream):
                torch.zeros(1, device=flat.grad.device).add_(1.0)
            self.stream.synchronize()
            try:
                if os.environ.get("MTP_NATIVE_COMM") == "fail":      # (test switch: behave as if mtp_comm_init had failed on this rank)
                    raise RuntimeError("mtp_comm_init: injected failure (MTP_NATIVE_COMM=fail)")
                self.native = RcclComm(group)
            except Exception as e:       # (MTP_NATIVE_COMM=strict: no second choice)
                if os.environ.get("MTP_NATIVE_COMM") == "strict":
                    raise
                self.native_error = "%s: %s" % (type(e).__name__, e)
            # every rank must exchange through the same library entry: if ANY rank could not create its communicator, all of them use torch.distributed's
            # collectives (the same RCCL; said on stderr and in describe(), never silently) -- a mixed job would hang in its first collective
            ok = torch.tensor([0.0 if self.native is None else 1.0], device=flat.grad.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if float(ok.item()) < 1.0:
                if self.native is not None:
                    self.native.close()
                    self.native = None
                    self.native_error = "another rank could not create its RCCL communicator through the C ABI"
                import warnings
                warnings.warn("mtp_comm_init failed (%s): the gradient exchange uses torch.distributed's collectives on the side stream" % self.native_error)
        self.works = []
        self.pending_casts = []   # bf16 mode: (scratch bucket, f32 slice) pairs whose cast back waits for the collective
        self.start = 0
        self.bytes_reduced = 0    # gradient bytes covered (f32), independent of the wire format
        self.wire_bytes = 0       # bytes handed to the collectives (half of bytes_reduced in bf16 mode)
        self.collectives = 0
        self.timing = False       # bench.py: HIP events around every bucket's exchange on the side stream
        self.timed = []

    def begin_step(self):
        self.start, self.bytes_reduced, self.wire_bytes, self.collectives = 0, 0, 0, 0

    # ---- one bucket -----------------------------------------------------------------------------------------------------------
    def _cast(self, src, dst):
        if src.is_cuda:
            from . import ops
            ops.cast(src, dst)       # HIP kernel on the current (side) stream
        else:
            dst.copy_(src)
        return dst

    def _exchange(self, wire):
        """issue the collective(s) for one bucket on the current stream / asynchronously; returns the torch Work handles"""
        rs = self.mode == "rs_ag" and wire.numel() % self.world == 0
        if self.native is not None:
            if rs:
                self.native.reduce_scatter_(wire)
                self.native.all_gather_(wire)
                self.collectives += 2
            else:
                self.native.all_reduce_(wire)
                self.collectives += 1
            return []
        if rs:
            n = wire.numel() // self.world
            shard = wire[self.rank * n:(self.rank + 1) * n]
            if self.stream is not None:
                w1 = dist.reduce_scatter_tensor(shard, wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                w1.wait()       # stream-level: orders the all-gather behind the reduce-scatter on the side stream
                w2 = dist.all_gather_into_tensor(wire, shard, group=self.group, async_op=True)
                self.collectives += 2
                return [w2]
            # CPU / gloo: synchronous, and through a private copy of the shard (gloo does not promise the in-place forms)
            mine = torch.empty_like(shard)
            dist.reduce_scatter_tensor(mine, wire, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(wire, mine, group=self.group)
            self.collectives += 2
            return []
        self.collectives += 1
        return [dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]

    def _bucket(self, buf):
        if self.bf16:
            wire = self._cast(buf, torch.empty(buf.numel(), device=buf.device, dtype=torch.bfloat16))
        else:
            wire = buf
        self.wire_bytes += wire.numel() * wire.element_size()
        works = self._exchange(wire)
        if self.bf16:
            if self.stream is not None:
                for w in works:
                    w.wait()                       # stream-level wait on the side stream, then the cast back on the same stream
                self._cast(wire, buf)
                works = []
            else:
                self.pending_casts.append((wire, buf))
        return works

    def on_block_done(self, gid):
        """engine hook: gradients of group `gid` (and everything before it in completion order) are on the compute stream."""
        if not self.active:
            return
        if not any(self.flat.groups[n] == gid for n in self.flat.names):
            return
        end = self.flat.group_end(gid)
        if end <= self.start or ((end - self.start) * 4 < self.bucket_bytes and gid != self.last_gid):
            return
        buf = self.flat.grad[self.start:end]
        self.bytes_reduced += (end - self.start) * 4
        self.start = end
        if self.stream is None:
            self.works += self._bucket(buf)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if self.timing:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.stream)
            works = self._bucket(buf)
            if self.timing:
                for w in works:
                    w.wait()                  # (stream-level wait: orders t1 behind the collective on the side stream)
                t1.record(self.stream)
                self.timed.append((buf.numel() * 4, t0, t1))
                works = []
            self.works += works

    def finish(self):
        """make the compute stream (or the host, on CPU) wait for every outstanding bucket."""
        if self.active and self.start < self.flat.reduced:      # (a caller that never reported the last group)
            self.on_block_done(self.last_gid)
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for w in self.works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self.works:
                w.wait()
            for wire, buf in self.pending_casts:
                buf.copy_(wire)
            self.pending_casts = []
        self.works = []

    def describe(self):
        """what bench.py prints about the exchange (library, algorithm knobs in effect)"""
        d = dict(mode=self.mode, wire_dtype="bf16" if self.bf16 else "f32", native_c_abi=self.native is not None, bucket_bytes=self.bucket_bytes)
        if getattr(self, "native_error", None):
            d["native_error"] = self.native_error
        if self.native is not None:
            try:
                d["communicator"] = self.native.info()       # ranks / rank / device / version as RCCL reports them for THIS communicator
            except Exception as e:
                d["communicator"] = "unreadable: %s" % e
        for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE", "NCCL_DEBUG"):
            if k in os.environ:
                d[k] = os.environ[k]
        try:
            if self.cuda:
                d["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        return d


class FlatAdamW:
    """torch.optim.AdamW semantics over FlatParams (lr 6e-5, betas (0.9, 0.999), wd 0.05 in the reference) with the
    cosine schedule of main_pretrain.py:832 and clip_grad_norm_(max_norm=5) of main_pretrain.py:786."""

    def __init__(self, flat, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=5.0, total_steps=None, world_size=1):
        self.flat, self.lr0, self.betas, self.eps, self.max_norm = flat, lr, betas, eps, max_norm
        self.weight_decay = weight_decay
        self.total_steps, self.world = total_steps, world_size
        dev = flat.data.device
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        st, wd = flat.weight_decay_segments(weight_decay)
        self.seg_start, self.seg_wd = st.to(dev), wd.to(dev)
        self.hyper = torch.zeros(6, device=dev, dtype=torch.float32)
        self.sqn = torch.zeros(1, device=dev, dtype=torch.float32)
        self.t = 0            # Adam step count (bias correction)
        self.last_epoch = 0   # CosineAnnealingLR.last_epoch: scheduler steps taken
        # The reference's loop is optimizer.step() -> [save the checkpoint] -> scheduler.step() (MAIN:788, 823-832).  step() here
        # leaves the scheduler step PENDING until the next step() (or scheduler_step()), so that a checkpoint written between two
        # steps holds what the reference's file holds at the same iteration: Adam step N, last_epoch N - 1, lr of epoch N - 1.
        self.sched_pending = False

    def lr_at(self, t):
        if not self.total_steps:
            return self.lr0
        return 0.5 * self.lr0 * (1.0 + math.cos(math.pi * min(t, self.total_steps) / self.total_steps))   # CosineAnnealingLR, eta_min 0

    def hyper_values(self):
        b1, b2 = self.betas
        return [self.lr_at(self.last_epoch), b1, b2, self.eps, 1.0 - b1 ** self.t, 1.0 - b2 ** self.t]

    # ---- checkpoint / resume in the reference's formats (MAIN:483-499, 823-829) --------------------------------------------
    def state_dict(self, module):
        """torch.optim.AdamW.state_dict() layout for the parameter groups the reference builds (reference_param_groups):
        loadable by a torch AdamW constructed the same way and vice versa.  Parameters that never receive a gradient
        (`norm.*`) carry no state, as in torch."""
        f = self.flat
        groups = reference_param_groups(module.named_parameters(), self.weight_decay)
        state, pgs, idx = {}, [], 0
        for name, scale, wd, names in groups:
            ids = []
            for n in names:
                if self.t > 0 and f.groups[n] is not None:
                    state[idx] = {"step": torch.tensor(float(self.t)), "exp_avg": f.view(self.m, n).detach().cpu().clone(),
                                  "exp_avg_sq": f.view(self.v, n).detach().cpu().clone()}
                ids.append(idx)
                idx += 1
            pgs.append({"lr": self.lr_at(self.last_epoch) * scale, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": wd, "amsgrad": False,
                        "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                        "initial_lr": self.lr0 * scale, "param_names": list(names), "lr_scale": scale, "group_name": name, "params": ids})
        return {"state": state, "param_groups": pgs}

    def load_state_dict(self, sd, module, prefixes=("encoder.", "module.encoder.", "backbone.", "")):
        """torch.optim.AdamW.state_dict() -> flat m / v / step.  Entries are matched BY NAME through the `param_names` the
        reference's constructor stores in every group (mmcv_custom/layer_decay_optimizer_constructor_vit.py:60-66), with the
        pretrain model's `encoder.` prefix stripped: the reference's checkpoints hold the optimizer of the WHOLE model
        (encoder + three decoders, MAIN:826-829); the decoders' entries are skipped.  State dicts without names (this class's
        own older files, plain torch) fall back to positional matching, which needs the same parameter list.
        Returns the number of parameters restored."""
        f = self.flat
        own = [n for g in reference_param_groups(module.named_parameters(), self.weight_decay) for n in g[3]]
        pairs = []      # (optimizer state index, our parameter name)
        named = all("param_names" in pg and len(pg["param_names"]) == len(pg["params"]) for pg in sd["param_groups"])
        if named:
            known = set(own)
            for pg in sd["param_groups"]:
                for i, full in zip(pg["params"], pg["param_names"]):
                    for pre in prefixes:
                        if full.startswith(pre) and full[len(pre):] in known:
                            pairs.append((i, full[len(pre):]))
                            break
        else:
            ids = [i for pg in sd["param_groups"] for i in pg["params"]]
            if len(ids) != len(own):
                raise ValueError("optimizer state without parameter names has %d entries, the backbone has %d parameters" % (len(ids), len(own)))
            pairs = list(zip(ids, own))
        t, restored = 0, 0
        self.m.zero_()
        self.v.zero_()
        for i, n in pairs:
            st = sd["state"].get(i)
            if st is None or f.groups[n] is None:
                continue
            if tuple(st["exp_avg"].shape) != f.shapes[n]:
                continue        # a different architecture under the same name: keep the fresh state
            f.view(self.m, n).copy_(st["exp_avg"])
            f.view(self.v, n).copy_(st["exp_avg_sq"])
            t = max(t, int(float(st["step"])))
            restored += 1
        self.t = t
        self.last_epoch = t      # (overwritten by load_scheduler_state_dict when the checkpoint carries the scheduler)
        self.sched_pending = False
        return restored

    def scheduler_state_dict(self):
        """the fields of torch.optim.lr_scheduler.CosineAnnealingLR.state_dict() (MAIN:457: T_max = end_iter, eta_min = 0)"""
        return {"T_max": self.total_steps, "eta_min": 0, "base_lrs": [self.lr0, self.lr0], "last_epoch": self.last_epoch, "verbose": False,
                "_step_count": self.last_epoch + 1, "_get_lr_called_within_step": False, "_last_lr": [self.lr_at(self.last_epoch)] * 2}

    def load_scheduler_state_dict(self, sd):
        self.total_steps = sd.get("T_max", self.total_steps)
        self.last_epoch = int(sd.get("last_epoch", self.last_epoch))
        self.sched_pending = False     # the file's scheduler is the reference's at its save point: the resumed loop continues from it

    def scheduler_step(self):
        """scheduler.step() of MAIN:832 for the iteration step() ran last (no-op when none is pending)"""
        if self.sched_pending:
            self.last_epoch += 1
            self.sched_pending = False

    def fuse_images(self, wimg):
        """from now on step() also writes the engine's GEMM-side weight images (ops.AdamWImages: one launch for the update and the images); returns whether the
        engine's images qualify (every image source is a whole parameter of the flat buffer).  MTP_FUSED_ADAMW=0 keeps the two passes (A/B)."""
        from . import ops
        self._fused, self._fused_for = None, wimg
        if wimg is None or not self.flat.data.is_cuda or os.environ.get("MTP_FUSED_ADAMW", "1") == "0":
            return False
        no_decay = ("pos_embed", "cls_token")
        wd = lambda n: 0.0 if (len(self.flat.shapes[n]) == 1 or n.endswith(".bias") or n in no_decay) else self.weight_decay      # (= FlatParams.weight_decay_segments)
        self._fused = ops.AdamWImages.build(self.flat, wimg, wd)
        return self._fused is not None

    def _rest_table(self, covered):
        """device tables (start, count) of the runs of [0, reduced) that `covered` (gradient tensors inside the flat buffer) leaves out, in pieces of at most 8 K
        floats (one workgroup each, 16-byte loads); cached by the covered ranges (the same every step for one model / schedule).  None when a tensor is not a slice of the flat gradient buffer."""
        f = self.flat
        base, item = f.grad.data_ptr(), 4
        rng = []
        for t in covered:
            off = t.data_ptr() - base
            if off < 0 or off % item or off // item + t.numel() > f.reduced or not t.is_contiguous():
                return None
            rng.append((off // item, t.numel()))
        key = tuple(sorted(rng))
        if getattr(self, "_rest_key", None) != key:
            ent, pos = [], 0
            for a, c in key:
                if a < pos:
                    return None          # overlapping tensors: not a partition
                if a > pos:
                    ent += [(pos + o, min(8192, a - pos - o)) for o in range(0, a - pos, 8192)]
                pos = a + c
            if pos < f.reduced:
                ent += [(pos + o, min(8192, f.reduced - pos - o)) for o in range(0, f.reduced - pos, 8192)]
            dev = f.grad.device
            self._rest_tab = (torch.tensor([e[0] for e in ent], dtype=torch.int64, device=dev), torch.tensor([e[1] for e in ent], dtype=torch.int64, device=dev)) if ent else None
            self._rest_key = key
        return self._rest_tab if self._rest_tab is not None else ()

    def step(self, norm_covered=None):
        """returns True when the step also refreshed the engine's weight images (fuse_images).
        norm_covered: gradient tensors whose squared norm the weight-gradient launches have ALREADY added to self.sqn (BackboneEngine.backward(sqn=...)): only the
        rest of the buffer is summed here.  None: one pass over the whole gradient buffer (every world size > 1: the norm is that of the REDUCED gradients)."""
        from . import ops
        self.scheduler_step()
        self.t += 1
        self.hyper.copy_(torch.tensor(self.hyper_values(), dtype=torch.float32), non_blocking=True)
        f = self.flat
        n = f.reduced
        gs = 1.0 / self.world
        sq = None
        if self.max_norm and self.max_norm > 0:
            rest = self._rest_table(norm_covered) if norm_covered else None
            if rest is None:
                self.sqn.zero_()
                ops.sqnorm(f.grad[:n], self.sqn)
            elif len(rest):
                ops.sqnorm_segments(f.grad, rest[0], rest[1], self.sqn)
            sq = self.sqn
        self.sched_pending = True     # scheduler.step() of MAIN:832 happens after the save point: see __init__
        if getattr(self, "_fused", None) is not None:
            self._fused.step(self.m, self.v, self.hyper, sq, float(self.max_norm or 0.0), gs)
            return True
        ops.adamw_flat(f.data[:n], f.grad[:n], self.m[:n], self.v[:n], self.seg_start, self.seg_wd, self.hyper, sq, float(self.max_norm or 0.0), gs)
        return False


class DataParallelTrainer:
    """fwd -> loss -> bwd (+ overlapped bucketed all-reduce) -> clip + AdamW, on the HIP engine."""

    def __init__(self, module, lr=6e-5, weight_decay=0.05, max_norm=5.0, total_steps=None, bucket_bytes=64 << 20, feature_dtype=None,
                 comm_mode=None, comm_bf16=None):
        self.module = module
        self.engine = module._engine()
        self.flat = FlatParams(module, unused=module._unused_params)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.flat.data.is_cuda and hasattr(self.engine, "warm_streams"):
            self.engine.warm_streams(self.flat.data.device)      # before the reducer creates its RCCL communicator
        self.reducer = GradReducer(self.flat, bucket_bytes, mode=comm_mode, bf16=comm_bf16)
        self.opt = FlatAdamW(self.flat, lr=lr, weight_decay=weight_decay, max_norm=max_norm, total_steps=total_steps, world_size=self.world)
        self.feature_dtype = feature_dtype
        self.sync_replicas()

    def sync_replicas(self, optimizer_state=False):
        """rank 0's parameters (and, after a resume, optimizer state and counters) to every rank -- what DistributedDataParallel
        does at construction (MAIN:508-518): replicas that were seeded or loaded differently would otherwise sum unrelated
        gradients silently."""
        if self.world <= 1:
            return
        dist.broadcast(self.flat.data, src=0)
        if optimizer_state:
            dist.broadcast(self.opt.m, src=0)
            dist.broadcast(self.opt.v, src=0)
            cnt = torch.tensor([self.opt.t, self.opt.last_epoch, self.opt.total_steps or 0, int(self.opt.sched_pending)], dtype=torch.int64,
                               device=self.flat.data.device)
            dist.broadcast(cnt, src=0)
            self.opt.t, self.opt.last_epoch = int(cnt[0]), int(cnt[1])
            self.opt.total_steps = int(cnt[2]) or None
            self.opt.sched_pending = bool(int(cnt[3]))
        self.engine._key = None
        self.engine._images_fresh = None

    # ---- encoder checkpoint in the reference's dict format (MAIN:823-829 save, MAIN:483-499 resume) -------------------------
    def checkpoint(self, epoch=0, iteration=None, losses=()):
        import numpy as np
        return {"epoch": epoch, "iteration": self.opt.t if iteration is None else iteration,
                "state_dict": {k: v.detach().cpu().clone() for k, v in self.module.state_dict().items()},
                "optimizer": self.opt.state_dict(self.module), "scheduler": self.opt.scheduler_state_dict(),
                "loss_pretrain": np.array(list(losses))}

    def save_checkpoint(self, path, **kw):
        """rank 0 writes (MAIN:823: main_process only); every rank holds identical state after the all-reduce"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
            torch.save(self.checkpoint(**kw), path)

    def load_checkpoint(self, ckpt):
        """ckpt: path or dict.  Same tolerance as MAIN:487-495: keys present in both are loaded, the rest keep their values.
        Returns (epoch, iteration, losses)."""
        if not isinstance(ckpt, dict):
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
        own = self.module.state_dict()
        with torch.no_grad():
            for k, v in ckpt["state_dict"].items():
                if k in own:
                    own[k].copy_(v)        # in place: the parameters stay views of the flat buffer
        self.restored_optimizer_entries = None
        if "optimizer" in ckpt:
            # A checkpoint that carries optimizer state none of which matches a backbone parameter (another wrapper prefix, another
            # architecture) would otherwise resume with zero moments and step 0 at the checkpoint's late-schedule learning rate -- silently.
            n = self.opt.load_state_dict(ckpt["optimizer"], self.module)
            self.restored_optimizer_entries = n
            have = len(ckpt["optimizer"].get("state", {}))
            want = sum(1 for k in self.flat.names if self.flat.groups[k] is not None)
            if have and n == 0:
                raise ValueError("the checkpoint holds optimizer state for %d parameters but none of them matches a backbone parameter (expected names "
                                 "like 'encoder.blocks.0...' in param_groups[*]['param_names'], or a positional match)" % have)
            if have and n < want:
                import warnings
                warnings.warn("optimizer state restored for %d of the backbone's %d trained parameters; the others restart with zero moments" % (n, want))
        if "scheduler" in ckpt:
            self.opt.load_scheduler_state_dict(ckpt["scheduler"])
        self.engine._key = None
        self.engine._images_fresh = None
        self.sync_replicas(optimizer_state=True)
        losses = ckpt.get("loss_pretrain", [])
        return ckpt.get("epoch", 0), ckpt.get("iteration", self.opt.t), (losses.tolist() if hasattr(losses, "tolist") else list(losses))

    def step(self, img, loss_and_grads):
        """loss_and_grads(feats) -> (loss, [dfeat or None] * 4).  Returns the (local) loss tensor."""
        self.flat.zero_accumulating()     # the engine accumulates bias / LayerNorm / table gradients; the big weight gradients are overwritten
        self.reducer.begin_step()
        feats, ctx = self.engine.forward(img, training=True, need_grad=True, feature_dtype=self.feature_dtype)
        loss, dfeats = loss_and_grads(feats)
        # the gradient norm of the clipping step as a by-product of the weight-gradient launches (one rank only: with an exchange the norm is that of the REDUCED
        # gradients; MTP_FUSED_SQNORM=0: always the separate pass)
        fold = (self.world == 1 and not self.reducer.active and self.flat.grad.is_cuda and bool(self.opt.max_norm) and os.environ.get("MTP_FUSED_SQNORM", "1") != "0"
                and "sqn" in self.engine.backward.__code__.co_varnames)
        kw = {}
        if fold:
            self.opt.sqn.zero_()
            kw["sqn"] = self.opt.sqn
        # (split_last: with collectives in flight, block 0's weight gradients go out on their own so that only ~50 MB stay exposed)
        self.engine.backward(ctx, dfeats, self.flat.G, on_block_done=self.reducer.on_block_done, split_last=self.reducer.active, **kw)
        self.reducer.finish()
        covered = list(getattr(self.engine, "norm_covered", None) or []) if fold else None
        wimg = getattr(self.engine, "_wimg", None)
        if getattr(self.opt, "_fused_for", 0) is not wimg:       # (first step, or the engine rebuilt its image buffers)
            self.opt.fuse_images(wimg if not getattr(self.engine, "_ls", None) else None)
        fresh = self.opt.step(norm_covered=covered)
        self.engine._key = None   # parameters changed under torch's version counters: rebuild the GEMM weight images next forward ...
        if fresh and hasattr(self.engine, "mark_images_fresh"):
            self.engine.mark_images_fresh()       # ... unless the optimizer kernel has just written them (the packed ConvT / convolution weights still follow)
        return loss
