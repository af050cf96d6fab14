"""Explicit forward/backward schedule of the ViT+RVSA backbone on the HIP kernels.

Host-side mirror of ViT_Win_RVSA_V3_WSZ7.forward_features (reference VIT:787-813) and of its autograd: instead of a
traced graph, the engine runs a fixed list of C-ABI kernel launches on the current HIP stream with caller-owned
buffers, keeps the saved activations of each block explicitly (or recomputes them: use_checkpoint, VIT:799-800),
and produces every parameter gradient into caller-provided f32 buffers (the flat, reverse-execution-ordered
gradient buffer of mtp_amd.parallel when training data-parallel).

Data layout: tokens (T, C) row-major, T = B*Hp*Wp; residual stream and its gradient f32;
everything consumed by a GEMM ("ACT": LN outputs, qkv, attention output, MLP hidden, and their gradients) in
`act_dtype` (bf16 = throughput mode, f32 = parity mode); parameters / parameter gradients / LN statistics f32.
"""
import os

import torch

from . import ops

F32 = torch.float32


class _Blk:
    """Per-block prepared weights (ACT-dtype copies, transposes) and parameter handles."""
    __slots__ = ("window", "pre", "wqkv", "wqkvT", "wproj", "wprojT", "w1", "w1T", "w2", "w2T", "wsamp", "bsamp", "bproj", "b2", "wproj_s", "w2_s")


class BackboneEngine:
    def __init__(self, module, act_dtype=torch.bfloat16):
        self.m = module
        self.act = act_dtype
        self.C = module.embed_dim
        self.depth = len(module.blocks)
        self.heads = module.num_heads
        self.hd = self.C // self.heads
        self.scale = module.qk_scale if module.qk_scale is not None else self.hd ** -0.5
        self.window = [bool(w) for w in module.window_blocks]
        self.out_indices = list(module.out_indices)
        self._key = None
        self._ln_parts = []
        self._sl_jobs = []
        self._blk = None
        self._fpn = None
        self._pe = None
        self._wimg = None
        self._wimg_ptrs = None
        self._ls = []
        self._zero_rel = {}

    # ------------------------------------------------------------------ parameters
    def params(self):
        """name -> nn.Parameter (reference state-dict names)."""
        return dict(self.m.named_parameters())

    def _weights_key(self, P):
        return (self.act,) + tuple((p.data_ptr(), p._version) for p in P.values())

    def prepare_weights(self, force=False):
        """Refresh the GEMM-side weight images when a parameter changed: ACT-dtype copies (W) and transposes (W^T) for the
        dgrad GEMMs, the three RVSA 1x1-conv heads stacked -- ONE launch over a descriptor table (ops.WeightImages) into
        persistent buffers; ConvTranspose2d weights as (4C, C) GEMM matrices (3 small launches)."""
        P = self.params()
        key = self._weights_key(P)
        if not force and key == self._key:
            return
        ptrs = (self.act,) + tuple(p.data_ptr() for p in P.values())
        if self._wimg is None or ptrs != self._wimg_ptrs:
            self._build_weight_images(P)
            self._wimg_ptrs = ptrs
        with torch.no_grad():
            for pre, b in self._ls:       # layer scale folded into the sources of the proj / fc2 images and their biases
                g1, g2 = P[pre + "gamma_1"].detach(), P[pre + "gamma_2"].detach()
                torch.mul(P[pre + "attn.proj.weight"].detach(), g1[:, None], out=b.wproj_s)
                torch.mul(P[pre + "mlp.fc2.weight"].detach(), g2[:, None], out=b.w2_s)
                torch.mul(P[pre + "attn.proj.bias"].detach(), g1, out=b.bproj)
                torch.mul(P[pre + "mlp.fc2.bias"].detach(), g2, out=b.b2)
        # (the optimizer launch of DataParallelTrainer writes the images itself and says so: mark_images_fresh -- honoured only while no parameter has been
        #  touched through torch since, i.e. the version counters still are what they were then)
        if force or getattr(self, "_images_fresh", None) != key:
            self._wimg.refresh()
        self._images_fresh = None
        for name, (wg, wgT) in self._fpn.items():
            ops.convt_pack(P[name + ".weight"].detach().contiguous(), wg, wgT)
        self._key = key

    def mark_images_fresh(self):
        """the GEMM-side weight images have just been written from the current parameters by somebody else (mtp_adamw_weight_images): the next
        prepare_weights() skips its own image launch -- unless a parameter is modified through torch in between"""
        self._images_fresh = self._weights_key(self.params())
        self._key = None

    def _build_weight_images(self, P):
        dev = P["patch_embed.proj.weight"].device
        act, H, C = self.act, self.heads, self.C
        entries = []

        def both(w2d):
            w2d = w2d.detach()
            assert w2d.is_contiguous()
            R, Cc = w2d.shape
            wt = torch.empty(Cc, R, device=dev, dtype=act)
            w = w2d if act == F32 else torch.empty(R, Cc, device=dev, dtype=act)   # f32 mode multiplies by the parameter itself
            entries.append((w2d, None if act == F32 else w, wt, False))
            return w, wt

        # Layer scale (init_values, VIT:500-512): x + gamma * (W y + b) = x + (diag(gamma) W) y + gamma * b -- the factor is folded into the images of the
        # two branch-closing projections and into their biases (f32 temporaries refreshed with the images), so the forward and the data gradients run the
        # unscaled schedule; the weight gradients are un-folded after their launch (_layer_scale_grads).  None of MTP's factories uses it.
        ls = getattr(self.m, "init_values", None) is not None
        self._ls = []

        blks = []
        for i in range(self.depth):
            b = _Blk()
            b.window = self.window[i]
            b.pre = pre = "blocks.%d." % i
            b.wqkv, b.wqkvT = both(P[pre + "attn.qkv.weight"])
            b.bproj, b.b2 = P[pre + "attn.proj.bias"].detach(), P[pre + "mlp.fc2.bias"].detach()
            if ls:
                b.wproj_s, b.w2_s = torch.empty_like(P[pre + "attn.proj.weight"]), torch.empty_like(P[pre + "mlp.fc2.weight"])
                b.bproj, b.b2 = torch.empty_like(P[pre + "attn.proj.bias"]), torch.empty_like(P[pre + "mlp.fc2.bias"])
                self._ls.append((pre, b))
                b.wproj, b.wprojT = both(b.wproj_s)
            else:
                b.wproj, b.wprojT = both(P[pre + "attn.proj.weight"])
            b.w1, b.w1T = both(P[pre + "mlp.fc1.weight"])
            b.w2, b.w2T = both(b.w2_s if ls else P[pre + "mlp.fc2.weight"])
            if b.window:
                b.wsamp = torch.empty(5 * H, C, device=dev, dtype=F32)
                b.bsamp = torch.empty(5 * H, device=dev, dtype=F32)
                r0 = 0
                for head, rows in (("sampling_offsets", 2 * H), ("sampling_scales", 2 * H), ("sampling_angles", H)):
                    entries.append((P[pre + "attn.%s.2.weight" % head].detach().view(rows, C), b.wsamp[r0:r0 + rows], None, True))
                    entries.append((P[pre + "attn.%s.2.bias" % head].detach().view(1, rows), b.bsamp[r0:r0 + rows], None, True))
                    r0 += rows
            blks.append(b)
        self._blk = blks
        wpe = P["patch_embed.proj.weight"].detach()
        self._pe = both(wpe.view(wpe.shape[0], -1))
        fpn = {}
        for name in ("fpn1.0", "fpn1.3", "fpn2.0"):
            if name + ".weight" in P:
                Cin, Cout = P[name + ".weight"].shape[:2]
                fpn[name] = (torch.empty(4 * Cout, Cin, device=dev, dtype=act), torch.empty(Cin, 4 * Cout, device=dev, dtype=act))
        self._fpn = fpn
        self._wimg = ops.WeightImages(entries, act)

    # ------------------------------------------------------------------ helpers
    # grouped weight-gradient launches on a side stream (ops.WgradQueue): the data-gradient GEMMs run 0.875 of whole
    # rounds and one workgroup per CU -- the weight-gradient tiles take the CUs a round leaves idle.  Round 4, same box, three runs each: 35.76 -> 35.38 ms / step;
    # rounds 2-3 had measured it as a loss (bursts of four blocks, the reductions not yet batched); next to the gradient exchange it needs
    # GPU_MAX_HW_QUEUES = 8 (mtp_amd/__init__.py: with 4 hardware queues the side stream shares the compute stream's).  DESIGN section 5.
    # False: everything on the current stream; 2: a stream of the device's lowest priority.
    wgrad_side_stream = True
    wgrad_max_jobs = 8            # side-stream mode: a burst goes out every two blocks (4 blocks = whole rounds matter only when nothing runs next to it):
                                  # 35.51 -> 35.25 ms on one box; one block per burst the same, three blocks worse
    wgrad_keep = 2                # bursts that may stay in flight on the side stream when the next one is launched

    def _wgrad_stream(self):
        if not self.wgrad_side_stream:
            return None
        import mtp_amd
        note = mtp_amd.hw_queue_note()        # once per process: the side stream needs a hardware queue of its own (mtp_amd/__init__.py)
        if note:
            import warnings
            warnings.warn(note, RuntimeWarning, stacklevel=2)
        if int(self.wgrad_side_stream) == 2:      # a stream of the device's lowest priority
            return ops.low_priority_stream(self.dev)
        st = getattr(self, "_wstream", None)
        if st is None or st.device != self.dev:
            st = self._wstream = torch.cuda.Stream(device=self.dev)
        return st

    def warm_streams(self, device):
        """create AND use the weight-gradient side stream now: the HIP runtime hands a stream its hardware queue at first use, in the order of first uses.  A library
        that creates streams of its own in between (RCCL at communicator creation) otherwise pushes the side stream onto the compute stream's queue -- measured in
        round 6: 34.6 -> 47 ms per step when the C-ABI communicator was created before the first backward (tools/probes/native_comm_probe.py)."""
        self.dev = torch.device(device)
        st = self._wgrad_stream()
        if st is not None:
            with torch.cuda.stream(st):
                torch.zeros(1, device=self.dev).add_(1.0)
            st.synchronize()

    def _e(self, *shape, dtype=None):
        return torch.empty(*shape, device=self.dev, dtype=dtype or self.act)

    # Bias / LayerNorm gradients ACCUMULATE into G (no per-call clearing pass: that was ~250 tiny memsets per step).
    # Contract of backward(): the gradient buffers in G are zero on entry (zeros_like in the autograd path, one memset of
    # the flat buffer per step in mtp_amd.parallel).
    @staticmethod
    def _colsum(dy, out):
        return ops.colsum(dy, out, accumulate=True)

    def _ln_bwd(self, *args, **kw):
        # the dgamma / dbeta partial rows wait in self._ln_parts and are reduced once per burst of blocks (_ln_flush)
        return ops.layernorm_bwd(*args, accumulate=True, defer=self._ln_parts, **kw)

    def _ln_flush(self):
        if self._ln_parts:
            ops.reduce_rows_deferred(self._ln_parts)
        if self._sl_jobs:
            ops.small_linear_dw_segments_flush(self._sl_jobs)

    def _full_rel(self, pre, Hp, Wp):
        """decomposed rel-pos tables of a full-attention block; zero tables for the ViTDet-style fine-tune copies, whose full
        attention has none (mmdet vit_rvsa_mtp.py:73-74, 93) -- adding q.0 is exact, so the same kernels serve both"""
        h = self.P.get(pre + "attn.full_attn_rel_pos_h")
        if h is not None:
            w = self.P[pre + "attn.full_attn_rel_pos_w"]
            # the kernels index rel_h[hq - hk + Hp - 1] / rel_w[wq - wk + Wp - 1] with the RUNTIME grid: a table sized for another
            # input size would be read (and, in the backward, written) out of bounds.  The reference fails here too
            # (calc_rel_pos_spatial's reshape / index error, VIT:142-193).
            # (A LARGER table is fine and means what it means in the reference: rows [0, 2 Hp - 2] are used -- the tables are sized from
            #  patch_shape[0] for both axes, VIT:81-84, so every non-square input has a taller rel_pos_w than it needs.)
            if h.shape[0] < 2 * Hp - 1 or w.shape[0] < 2 * Wp - 1:
                raise ValueError("%sattn.full_attn_rel_pos_h/w have %d / %d rows, a %d x %d token grid needs %d / %d (resize them like "
                                 "init_weights does, or build the model for this input size)" % (pre, h.shape[0], w.shape[0], Hp, Wp, 2 * Hp - 1, 2 * Wp - 1))
            return h[:2 * Hp - 1], w[:2 * Wp - 1]      # (row slices of contiguous tables are contiguous)
        key = (Hp, Wp)
        if self._zero_rel.get(key) is None:
            self._zero_rel[key] = (torch.zeros(2 * Hp - 1, self.hd, device=self.dev, dtype=F32),
                                   torch.zeros(2 * Wp - 1, self.hd, device=self.dev, dtype=F32))
        return self._zero_rel[key]

    def _drop_scales(self, B, training):
        """VIT:31-42, 619: per-sample factor floor(keep + U[0,1)) / keep for each residual branch; None when inactive.
        One RNG launch for the whole network (the per-block torch.rand / floor / div were ~100 tiny launches per step)."""
        rates = [float(self.m.drop_path_rates[i]) for i in range(self.depth)]
        live = [training and r > 0.0 and self.m.blocks[i].training for i, r in enumerate(rates)]   # (a frozen stage is in eval(): no drop-path)
        if not any(live):
            return [(None, None)] * self.depth
        keep = torch.tensor([1.0 - r for r in rates], dtype=F32).to(self.dev, non_blocking=True).view(self.depth, 1, 1)
        s = torch.floor(keep + torch.rand(self.depth, 2, B, device=self.dev, dtype=F32)) / keep
        return [(s[i, 0], s[i, 1]) if live[i] else (None, None) for i in range(self.depth)]

    # ------------------------------------------------------------------ block forward
    def _block_fwd(self, i, x, B, Hp, Wp, dps, save):
        P, b, C, T, N = self.P, self._blk[i], self.C, x.shape[0], Hp * Wp
        pre = b.pre
        s = {}
        mean1, rstd1 = self._e(T, dtype=F32), self._e(T, dtype=F32)
        ln1 = ops.layernorm_fwd(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], self._e(T, C), mean1, rstd1)
        if b.window:
            # (the sampling heads on a second stream next to the qkv GEMM, joined by events: +0.25 ms per forward pass, round 4 -- the two cross-queue
            #  waits per block cost more than the 22 us they hide)
            nh, nw = ops.rvsa_windows(Hp, Wp)
            R = B * nh * nw
            avg, pooled = self._e(R, C, dtype=F32), self._e(R, C, dtype=F32)
            samp = ops.rvsa_sampling_fwd(ln1, b.wsamp, b.bsamp, avg, pooled, self._e(R, 5 * self.heads, dtype=F32), B, Hp, Wp)
        qkv = ops.gemm_nt(ln1, b.wqkv, self._e(T, 3 * C), bias=P[pre + "attn.qkv.bias"])
        o = self._e(T, C)
        if b.window:
            lse = self._e(R * self.heads * 49, dtype=F32)
            ops.rvsa_attn_fwd(qkv, samp, o, lse, P[pre + "attn.rel_pos_h"], P[pre + "attn.rel_pos_w"],
                              P[pre + "attn.relative_position_bias_table"], B, Hp, Wp, self.heads, self.scale)
            s.update(avg=avg, pooled=pooled, samp=samp)
        else:
            lse = self._e(B * self.heads * N, dtype=F32)
            rel_h, rel_w = self._full_rel(pre, Hp, Wp)
            ops.full_attn_fwd(qkv, o, lse, rel_h, rel_w, B, Hp, Wp, self.heads, self.scale)
        x1 = ops.gemm_nt(o, b.wproj, self._e(T, C, dtype=F32), epi=ops.EPI_BIAS_RES, bias=b.bproj, res=x,
                         rowscale=dps[0], rows_per_sample=N)
        mean2, rstd2 = self._e(T, dtype=F32), self._e(T, dtype=F32)
        ln2 = ops.layernorm_fwd(x1, P[pre + "norm2.weight"], P[pre + "norm2.bias"], self._e(T, C), mean2, rstd2)
        u = self._e(T, 4 * C)     # gelu'(fc1 pre-activation): all the MLP backward needs of it (one multiplication in the dgrad epilogue)
        h = ops.gemm_nt(ln2, b.w1, self._e(T, 4 * C), epi=ops.EPI_BIAS_GELU_DG, bias=P[pre + "mlp.fc1.bias"], aux=u)
        x2 = ops.gemm_nt(h, b.w2, self._e(T, C, dtype=F32), epi=ops.EPI_BIAS_RES, bias=b.b2, res=x1,
                         rowscale=dps[1], rows_per_sample=N)
        if save:
            s.update(x=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, o=o, lse=lse, x1=x1, mean2=mean2, rstd2=rstd2, ln2=ln2, u=u, h=h)
        return x2, s

    def _layer_scale_grads(self, pre, lin, gname, G):
        """un-fold the layer scale: the launch just left d/d(diag(gamma) W) in G[W] and d/d(gamma * b) in G[b] (runs on the launch's stream, right behind it):
        d gamma = rowsum(G_W' * W) + G_b' * b;  d W = gamma (rows) * G_W';  d b = gamma * G_b'"""
        P = self.P
        gw, gb, gam = G[pre + lin + ".weight"], G[pre + lin + ".bias"], P[pre + gname]
        G[pre + gname].add_((gw * P[pre + lin + ".weight"]).sum(1) + gb * P[pre + lin + ".bias"])
        gw.mul_(gam[:, None])
        gb.mul_(gam)

    # ------------------------------------------------------------------ block backward
    def _block_bwd(self, i, s, dx2, dx2_act, B, Hp, Wp, dps, G, extra, prev_scale):
        """dx2 (f32) = gradient of the block output; dx2_act = dps[1]-scaled ACT copy.  Returns (dx0, dx0_act) for the
        block input where dx0_act is pre-scaled by `prev_scale` (the previous block's mlp drop-path factor)."""
        P, b, C, N = self.P, self._blk[i], self.C, Hp * Wp
        pre = b.pre
        T = dx2.shape[0]
        # ---- MLP branch
        # (bias gradients = column sums of dY: by-product of the dW GEMM that streams dY anyway)
        # (weight gradients: queued, launched together with those of the neighbouring blocks -- ops.WgradQueue)
        wq = self._wq
        ls = pre + "gamma_1" in G
        wq.add(dx2_act, s["h"], G[pre + "mlp.fc2.weight"], G[pre + "mlp.fc2.bias"], after=(lambda: self._layer_scale_grads(pre, "mlp.fc2", "gamma_2", G)) if ls else None,
               norm_ok=not ls)
        du = ops.gemm_nt(dx2_act, b.w2T, self._e(T, 4 * C), epi=ops.EPI_MUL, aux=s["u"])
        wq.add(du, s["ln2"], G[pre + "mlp.fc1.weight"], G[pre + "mlp.fc1.bias"])
        dln2 = ops.gemm_nt(du, b.w1T, self._e(T, C))
        del du
        dx1, dx1_act = self._e(T, C, dtype=F32), self._e(T, C)
        self._ln_bwd(dln2, s["x1"], s["mean2"], s["rstd2"], P[pre + "norm2.weight"], dx1, G[pre + "norm2.weight"], G[pre + "norm2.bias"],
                          dres=dx2, dx_copy=dx1_act, copy_scale=dps[0], rows_per_sample=N)
        # ---- attention branch
        wq.add(dx1_act, s["o"], G[pre + "attn.proj.weight"], G[pre + "attn.proj.bias"], after=(lambda: self._layer_scale_grads(pre, "attn.proj", "gamma_1", G)) if ls else None,
               norm_ok=not ls)
        do = ops.gemm_nt(dx1_act, b.wprojT, self._e(T, C))
        dqkv = self._e(T, 3 * C)
        if b.window:
            nh, nw = ops.rvsa_windows(Hp, Wp)
            R = B * nh * nw
            H = self.heads
            dsamp = self._e(R, 5 * H, dtype=F32)
            ops.rvsa_attn_bwd(s["qkv"], s["samp"], s["o"], do, s["lse"], dqkv, dsamp,
                              P[pre + "attn.rel_pos_h"], P[pre + "attn.rel_pos_w"], P[pre + "attn.relative_position_bias_table"],
                              G[pre + "attn.rel_pos_h"], G[pre + "attn.rel_pos_w"], G[pre + "attn.relative_position_bias_table"],
                              B, Hp, Wp, H, self.scale, accumulate=True, defer=self._ln_parts)
            names = ("sampling_offsets", "sampling_scales", "sampling_angles")
            # the stacked heads' weight / bias gradients accumulate straight into the three parameters' buffers; queued, one launch per burst of
            # blocks (round 4: 20 launches of 18 us that are pure latency -> one per weight-gradient group)
            self._sl_jobs.append((s["pooled"], dsamp, [G[pre + "attn.%s.2.weight" % n].view(-1, C) for n in names],
                                  [G[pre + "attn.%s.2.bias" % n] for n in names]))
        else:
            rel_h, rel_w = self._full_rel(pre, Hp, Wp)
            if pre + "attn.full_attn_rel_pos_h" in G:
                # (tables taller than this grid needs -- every non-square input, VIT:81-84 -- get gradients in their first rows only)
                drel_h, drel_w = G[pre + "attn.full_attn_rel_pos_h"][:2 * Hp - 1], G[pre + "attn.full_attn_rel_pos_w"][:2 * Wp - 1]
            else:   # ViTDet-style copies: no such parameters -- the table gradients go to a scratch buffer
                drel_h, drel_w = self._e(*rel_h.shape, dtype=F32), self._e(*rel_w.shape, dtype=F32)
            ops.full_attn_bwd(s["qkv"], s["o"], do, s["lse"], dqkv, rel_h, rel_w, drel_h, drel_w, B, Hp, Wp, self.heads, self.scale,
                              accumulate=True, defer=self._ln_parts)
        wq.add(dqkv, s["ln1"], G[pre + "attn.qkv.weight"], G[pre + "attn.qkv.bias"])
        win_add = None
        if b.window:
            # dln1 += pool'(linear'(dsamp)): the per-window factor (windows, C) is one small launch; norm1's backward adds it to every token
            # row of the window while it reads the row anyway (round 4: was a read-modify-write pass over (T, C), 17 us per block).
            win_add = ops.rvsa_sampling_bwd_win(dsamp, b.wsamp, s["avg"], self._e(*s["avg"].shape, dtype=F32))
        dln1 = ops.gemm_nt(dqkv, b.wqkvT, self._e(T, C))
        dx0, dx0_act = self._e(T, C, dtype=F32), self._e(T, C)
        self._ln_bwd(dln1, s["x"], s["mean1"], s["rstd1"], P[pre + "norm1.weight"], dx0, G[pre + "norm1.weight"], G[pre + "norm1.bias"],
                          dres=dx1, extra=extra, dx_copy=dx0_act, copy_scale=prev_scale, rows_per_sample=N,
                          **(dict(win_add=win_add, grid=(B, Hp, Wp)) if win_add is not None else {}))
        return dx0, dx0_act

    # ------------------------------------------------------------------ whole forward
    def forward(self, img, training=False, need_grad=False, feature_dtype=None):
        """img (B,3,H,W) f32 on the GPU -> ([f1..f4] NCHW, ctx).  ctx is None unless need_grad."""
        m = self.m
        self.dev = img.device
        self.prepare_weights()
        self.P = P = {k: v.detach() for k, v in self.params().items()}
        C, act = self.C, self.act
        ps = m.patch_size
        fdt = feature_dtype or act
        if img.dtype == torch.uint8:
            # raw (B,H,W,3) uint8 batch: MTP_DataPreprocessor's normalise / flip / pad (preprocessing.py:145-148, MODELS:37-41)
            # fused with the im2col -- the f32 NCHW image is never materialised
            pp = getattr(m, "data_preprocessor", None)
            if pp is None:
                raise ValueError("uint8 input needs ViT_Win_RVSA_V3_WSZ7.set_data_preprocessor(mean, std, ...)")
            B, H, W, Cin = img.shape
            Hp, Wp = ops.padded_grid(H, W, ps, pp["pad_size_divisor"])
            N, T = Hp * Wp, B * Hp * Wp
            cols = ops.preprocess_patchify(img.contiguous(), self._e(T, Cin * ps * ps), ps, pp["mean"], pp["std"], pp["bgr_to_rgb"],
                                           pp["pad_size_divisor"], pp["pad_value"])
            H, W = Hp * ps, Wp * ps
        else:
            B, Cin, H, W = img.shape
            Hp, Wp = H // ps, W // ps
            N, T = Hp * Wp, B * Hp * Wp
            img = img.contiguous()
            if img.dtype != F32:
                img = img.float()
            # ---- patch embed (+ abs pos embed fused as the "residual" of the GEMM epilogue; VIT:536-539, 793-794)
            cols = ops.patchify(img, self._e(T, Cin * ps * ps), ps)
        pos = P.get("pos_embed")
        if pos is not None:
            assert pos.shape[1] == N, "pos_embed does not match the input size"
            x = ops.gemm_nt(cols, self._pe[0], self._e(T, C, dtype=F32), epi=ops.EPI_BIAS_RES, bias=P["patch_embed.proj.bias"],
                            res=pos.reshape(N, C), res_mod=N)
        else:
            x = ops.gemm_nt(cols, self._pe[0], self._e(T, C, dtype=F32), bias=P["patch_embed.proj.bias"])
        dps = self._drop_scales(B, training)
        ckpt = bool(m.use_checkpoint) and need_grad
        saved, taps = [], {}
        vitdet = bool(getattr(m, "_vitdet", False))
        last = self.depth - 1 if vitdet else max(self.out_indices)
        for i in range(self.depth):
            if i > last:
                break      # blocks after the last tap do not influence the outputs
            xin = x
            x, s = self._block_fwd(i, x, B, Hp, Wp, dps[i], save=need_grad and not ckpt)
            saved.append(s if (need_grad and not ckpt) else ({"x": xin} if need_grad else None))
            if i in self.out_indices:
                taps[i] = x
        final = None
        if vitdet:
            # mmdet / mmrotate `RVSA_MTP` (vit_rvsa_mtp.py:835-841): last block -> final norm -> the four fpn ops on that ONE map
            fm, fr = self._e(T, dtype=F32), self._e(T, dtype=F32)
            xn = ops.layernorm_fwd(x, P["norm.weight"], P["norm.bias"], self._e(T, C, dtype=F32), fm, fr)
            final = (x, fm, fr)
            tap_list = [xn, xn, xn, xn]
        else:
            tap_list = [taps[i] for i in self.out_indices]
        if getattr(m, "_taps_only", False):
            # mmpretrain / opencd `RVSA_MTP` (vit_rvsa_mtp.py:836-842): the taps as NCHW maps, no fpn ops
            feats = [ops.tokens_to_nchw(t, self._e(B, C, Hp, Wp, dtype=fdt), B, Hp, Wp, 0) for t in tap_list]
            fctx = {"taps_only": True}
        else:
            feats, fctx = self._fpn_fwd(tap_list, B, Hp, Wp, fdt, need_grad)
        ctx = None
        if need_grad:
            ctx = dict(saved=saved, dps=dps, fctx=fctx, cols=cols, geom=(B, Cin, H, W, Hp, Wp), ckpt=ckpt, last=last, final=final)
        return feats, ctx

    # ------------------------------------------------------------------ FPN tail (VIT:640-654, 807-811)
    def _fpn_fwd(self, taps, B, Hp, Wp, fdt, need_grad):
        P, C, T = self.P, self.C, B * Hp * Wp
        m = self.m
        feats, fctx = [], {}
        if m.patch_size == 8:
            # VIT:656-670: fpn1 = ConvT, fpn2 = identity, fpn3 = MaxPool 2, fpn4 = MaxPool 4 (= two 2 x 2 pools: the maximum of maxima)
            t0 = taps[0] if self.act == F32 else ops.cast(taps[0], self._e(T, C))
            z = ops.gemm_nt(t0, self._fpn["fpn1.0"][0], self._e(T, 4 * C), bias=P["fpn1.0.bias"], bias_mod=C)
            feats.append(ops.tokens_to_nchw(z.view(4 * T, C), self._e(B, C, 2 * Hp, 2 * Wp, dtype=fdt), B, Hp, Wp, 1))
            feats.append(ops.tokens_to_nchw(taps[1], self._e(B, C, Hp, Wp, dtype=fdt), B, Hp, Wp, 0))
            H2, W2, H4, W4 = Hp // 2, Wp // 2, Hp // 4, Wp // 4
            p3 = ops.maxpool2_tokens_fwd(taps[2], self._e(B * H2 * W2, C), B, Hp, Wp)
            feats.append(ops.tokens_to_nchw(p3, self._e(B, C, H2, W2, dtype=fdt), B, H2, W2, 0))
            q1 = ops.maxpool2_tokens_fwd(taps[3], self._e(B * H2 * W2, C, dtype=F32), B, Hp, Wp)
            q2 = ops.maxpool2_tokens_fwd(q1, self._e(B * H4 * W4, C), B, H2, W2)
            feats.append(ops.tokens_to_nchw(q2, self._e(B, C, H4, W4, dtype=fdt), B, H4, W4, 0))
            if need_grad:
                fctx = dict(p8=True, t0=t0, tap2=taps[2], tap3=taps[3], q1=q1)
            return feats, fctx
        if m.patch_size != 16:
            raise NotImplementedError("the reference defines the FPN tail for patch_size 16 and 8 only (VIT:640-670)")
        # fpn1: ConvT -> Norm2d -> GELU -> ConvT
        t0 = taps[0] if self.act == F32 else ops.cast(taps[0], self._e(T, C))
        y1 = ops.gemm_nt(t0, self._fpn["fpn1.0"][0], self._e(T, 4 * C), bias=P["fpn1.0.bias"], bias_mod=C)
        mean, rstd = self._e(4 * T, dtype=F32), self._e(4 * T, dtype=F32)
        g1 = ops.layernorm_fwd(y1.view(4 * T, C), P["fpn1.1.ln.weight"], P["fpn1.1.ln.bias"], self._e(4 * T, C), mean, rstd, gelu=True)
        y2 = ops.gemm_nt(g1, self._fpn["fpn1.3"][0], self._e(4 * T, 4 * C), bias=P["fpn1.3.bias"], bias_mod=C)
        feats.append(ops.tokens_to_nchw(y2.view(16 * T, C), self._e(B, C, 4 * Hp, 4 * Wp, dtype=fdt), B, Hp, Wp, 2))
        # fpn2: ConvT
        t1 = taps[1] if self.act == F32 else ops.cast(taps[1], self._e(T, C))
        z = ops.gemm_nt(t1, self._fpn["fpn2.0"][0], self._e(T, 4 * C), bias=P["fpn2.0.bias"], bias_mod=C)
        feats.append(ops.tokens_to_nchw(z.view(4 * T, C), self._e(B, C, 2 * Hp, 2 * Wp, dtype=fdt), B, Hp, Wp, 1))
        # fpn3: identity, fpn4: MaxPool2d(2,2)
        feats.append(ops.tokens_to_nchw(taps[2], self._e(B, C, Hp, Wp, dtype=fdt), B, Hp, Wp, 0))
        Ho, Wo = Hp // 2, Wp // 2
        pooled = ops.maxpool2_tokens_fwd(taps[3], self._e(B * Ho * Wo, C), B, Hp, Wp)
        feats.append(ops.tokens_to_nchw(pooled, self._e(B, C, Ho, Wo, dtype=fdt), B, Ho, Wo, 0))
        if need_grad:
            fctx = dict(t0=t0, y1=y1, mean=mean, rstd=rstd, g1=g1, t1=t1, tap3=taps[3])
        return feats, fctx

    def _convt_wgrad(self, dy, x, gw):
        """weight gradient of a 2x2 / stride-2 ConvTranspose2d as the GEMM dy (rows, 4 C_out)^T x (rows, C_in), queued with the blocks'
        weight gradients; the (4 C_out, C_in) image is re-laid into the parameter's (C_in, C_out, 2, 2) right after the launch"""
        dwg = self._e(dy.shape[1], x.shape[1], dtype=F32)
        self._wq.add(dy, x, dwg, after=lambda: ops.convt_unpack_grad(dwg, gw), norm_of=gw)      # (the re-laid copy holds the same values: same norm)

    def _fpn_bwd(self, dfeats, fctx, B, Hp, Wp, G):
        """returns [dtap0..dtap3] (f32 (T,C) each, or None when the feature received no gradient)."""
        P, C, T, act = self.P, self.C, B * Hp * Wp, self.act
        out = [None] * 4

        def as_in(df):
            df = df.contiguous()
            return df if df.dtype in (F32, torch.bfloat16) else df.float()

        if fctx.get("p8"):       # patch_size == 8 tail: ConvT | identity | MaxPool 2 | MaxPool 4
            H2, W2, H4, W4 = Hp // 2, Wp // 2, Hp // 4, Wp // 4
            if dfeats[0] is not None:
                dz = ops.nchw_to_tokens(as_in(dfeats[0]), self._e(4 * T, C), B, Hp, Wp, 1)
                self._convt_wgrad(dz.view(T, 4 * C), fctx["t0"], G["fpn1.0.weight"])
                self._colsum(dz, G["fpn1.0.bias"])
                out[0] = ops.gemm_nt(dz.view(T, 4 * C), self._fpn["fpn1.0"][1], self._e(T, C, dtype=F32))
            if dfeats[1] is not None:
                out[1] = ops.nchw_to_tokens(as_in(dfeats[1]), self._e(T, C, dtype=F32), B, Hp, Wp, 0)
            if dfeats[2] is not None:
                dp = ops.nchw_to_tokens(as_in(dfeats[2]), self._e(B * H2 * W2, C, dtype=F32), B, H2, W2, 0)
                out[2] = ops.maxpool2_tokens_bwd(fctx["tap2"], dp, self._e(T, C, dtype=F32), B, Hp, Wp)
            if dfeats[3] is not None:
                dq2 = ops.nchw_to_tokens(as_in(dfeats[3]), self._e(B * H4 * W4, C, dtype=F32), B, H4, W4, 0)
                dq1 = ops.maxpool2_tokens_bwd(fctx["q1"], dq2, self._e(B * H2 * W2, C, dtype=F32), B, H2, W2)
                out[3] = ops.maxpool2_tokens_bwd(fctx["tap3"], dq1, self._e(T, C, dtype=F32), B, Hp, Wp)
            return out

        if dfeats[0] is not None:
            dy2 = ops.nchw_to_tokens(as_in(dfeats[0]), self._e(16 * T, C), B, Hp, Wp, 2)
            self._convt_wgrad(dy2.view(4 * T, 4 * C), fctx["g1"], G["fpn1.3.weight"])
            self._colsum(dy2, G["fpn1.3.bias"])
            dg1 = ops.gemm_nt(dy2.view(4 * T, 4 * C), self._fpn["fpn1.3"][1], self._e(4 * T, C))
            dy1 = self._e(4 * T, C)
            self._ln_bwd(dg1, fctx["y1"].view(4 * T, C), fctx["mean"], fctx["rstd"], P["fpn1.1.ln.weight"], dy1,
                              G["fpn1.1.ln.weight"], G["fpn1.1.ln.bias"], beta=P["fpn1.1.ln.bias"], gelu=True)
            self._convt_wgrad(dy1.view(T, 4 * C), fctx["t0"], G["fpn1.0.weight"])
            self._colsum(dy1, G["fpn1.0.bias"])
            out[0] = ops.gemm_nt(dy1.view(T, 4 * C), self._fpn["fpn1.0"][1], self._e(T, C, dtype=F32))
        if dfeats[1] is not None:
            dz = ops.nchw_to_tokens(as_in(dfeats[1]), self._e(4 * T, C), B, Hp, Wp, 1)
            self._convt_wgrad(dz.view(T, 4 * C), fctx["t1"], G["fpn2.0.weight"])
            self._colsum(dz, G["fpn2.0.bias"])
            out[1] = ops.gemm_nt(dz.view(T, 4 * C), self._fpn["fpn2.0"][1], self._e(T, C, dtype=F32))
        if dfeats[2] is not None:
            out[2] = ops.nchw_to_tokens(as_in(dfeats[2]), self._e(T, C, dtype=F32), B, Hp, Wp, 0)
        if dfeats[3] is not None:
            Ho, Wo = Hp // 2, Wp // 2
            dp = ops.nchw_to_tokens(as_in(dfeats[3]), self._e(B * Ho * Wo, C, dtype=F32), B, Ho, Wo, 0)
            out[3] = ops.maxpool2_tokens_bwd(fctx["tap3"], dp, self._e(T, C, dtype=F32), B, Hp, Wp)
        return out

    # ------------------------------------------------------------------ whole backward
    def backward(self, ctx, dfeats, G, need_input_grad=False, on_block_done=None, split_last=False, sqn=None):
        """dfeats: 4 NCHW cotangents (or None).  G: name -> f32 gradient buffer (overwritten; parameters that receive no
        gradient -- `norm.*`, blocks after the last tap -- are left untouched).  on_block_done(i) is called once the gradients
        of block i AND of every block after it (and, for i == -1, of patch-embed / pos-embed) are complete on the current stream:
        once per burst of blocks whose weight gradients were launched together (ops.WgradQueue), with the lowest block index of
        the burst -- or after every block when nothing is queued (f32 parity mode, shapes the grouped kernel does not take: the
        weight gradients were launched immediately) -- the hook mtp_amd.parallel uses to launch RCCL collectives of contiguous
        gradient slices on a side stream.  split_last: launch what is queued after block 1 as well, so that the last report before
        the embeddings covers block 0 only (data-parallel runs: that last slice is the part of the exchange nothing overlaps).
        sqn (1-element f32 device tensor, zeroed by the caller): the grouped weight-gradient launches add the squared norm of what they write to it
        (ops.WgradQueue); afterwards self.norm_covered lists those gradient tensors -- the clipping step then sums only the rest (FlatAdamW.step)."""
        B, Cin, H, W, Hp, Wp = ctx["geom"]
        C, N, T = self.C, Hp * Wp, B * Hp * Wp
        P = self.P
        self.dev = ctx["cols"].device
        self._ln_parts = []
        self._sl_jobs = []
        # weight gradients (FPN deconvolutions, the blocks' Linears, patch embed) are queued and launched in bursts (ops.WgradQueue)
        self._wq = wq = ops.WgradQueue(stream=self._wgrad_stream())
        wq.max_jobs = self.wgrad_max_jobs if wq.stream is not None else 0      # (on the current stream a burst should be whole rounds of the CUs)
        wq.sqn = sqn
        self.norm_covered = wq.covered
        if ctx["fctx"].get("taps_only"):
            dtaps = [None if d is None else ops.nchw_to_tokens((d.contiguous() if d.dtype in (F32, torch.bfloat16) else d.float().contiguous()),
                                                               self._e(T, C, dtype=F32), B, Hp, Wp, 0) for d in dfeats]
        else:
            dtaps = self._fpn_bwd(dfeats, ctx["fctx"], B, Hp, Wp, G)
        tapgrad = {}
        last, dps, saved = ctx["last"], ctx["dps"], ctx["saved"]
        if ctx.get("final") is not None:
            # ViTDet-style copies: the four fpn gradients meet at the final norm's output; LN backward gives the last block's
            # output gradient (and norm.weight / norm.bias, which DO get gradients in this variant)
            dsum = None
            for d in dtaps:
                if d is None:
                    continue
                if dsum is None:
                    dsum = d
                else:
                    ops.axpy(dsum, d)
            if dsum is not None:
                xl, fm, fr = ctx["final"]
                tapgrad[last] = self._ln_bwd(dsum, xl, fm, fr, P["norm.weight"], self._e(T, C, dtype=F32), G["norm.weight"], G["norm.bias"])
        else:
            for idx, d in zip(self.out_indices, dtaps):
                if d is None:
                    continue
                if idx in tapgrad:
                    ops.axpy(tapgrad[idx], d)
                else:
                    tapgrad[idx] = d
        self._ln_flush()
        if on_block_done is not None and not wq.jobs:
            on_block_done(self.depth)          # FPN (and final-norm) parameter gradients are complete on the stream (queued FPN weight
                                               # gradients go out with the first burst of blocks, whose report covers them)
        if last not in tapgrad:
            tapgrad[last] = torch.zeros(T, C, device=self.dev, dtype=F32)
        dx = tapgrad[last]
        # ACT copy of the output gradient of the last block, scaled by its mlp drop-path factor
        dx_act = self._scaled_copy(dx, dps[last][1], N)
        waiting = []     # blocks whose weight gradients are still queued: on_block_done fires once they have been launched
        pending = []     # side-stream mode: (lowest block, launch mark) of the bursts in flight, reported once the current stream has waited for them
        for i in range(last, -1, -1):
            s = saved[i]
            if ctx["ckpt"]:
                _, s = self._block_fwd(i, s["x"], B, Hp, Wp, dps[i], save=True)
            extra = tapgrad.get(i - 1) if i > 0 else None
            prev_scale = dps[i - 1][1] if i > 0 else None
            dx, dx_act = self._block_bwd(i, s, dx, dx_act, B, Hp, Wp, dps[i], G, extra, prev_scale)
            saved[i] = None
            waiting.append(i)
            if i == 0:      # the patch-embed weight gradient rides in the last burst (its input gradient is block 0's dx)
                wq.add(dx_act, ctx["cols"], G["patch_embed.proj.weight"].view(C, -1), G["patch_embed.proj.bias"])
            # nothing queued (every weight gradient of the burst is already on the stream): report block by block -- the reducer
            # cuts its buckets by size, and a single report at the end would leave no backward to overlap the exchange with
            if i == 0 or wq.should_flush() or not wq.jobs or (split_last and i == 1):
                wq.flush()
                self._ln_flush()      # (on the current stream: behind the burst on the side stream they cost the whole gain, 35.4 -> 35.7 ms)
                if wq.stream is None:
                    if on_block_done is not None:
                        on_block_done(waiting[-1])     # the lowest block of the burst: its group end covers the whole burst
                else:
                    pending.append((waiting[-1], wq.launched))
                    wq.wait(keep=self.wgrad_keep)
                    while pending and pending[0][1] <= wq.launched - len(wq.inflight):      # bursts the current stream has waited for
                        g = pending.pop(0)[0]
                        if on_block_done is not None:
                            on_block_done(g)
                waiting = []
        wq.wait()
        if pending and on_block_done is not None:
            on_block_done(pending[-1][0])     # the lowest block still unreported covers the rest
        # ---- pos embed (the patch-embed weight gradient went out with block 0's)
        if "pos_embed" in G:
            ops.reduce_rows(dx.view(B, N * C), G["pos_embed"])
        dimg = None
        if need_input_grad:
            dcols = ops.gemm_nt(dx_act, self._pe[1], self._e(T, ctx["cols"].shape[1]))
            dimg = ops.unpatchify(dcols, self._e(B, Cin, H, W, dtype=F32), self.m.patch_size)
        if on_block_done is not None:
            on_block_done(-1)
        return dimg

    def _scaled_copy(self, dx, scale, N):
        return ops.scale_rows_cast(dx, self._e(*dx.shape), scale, N)
