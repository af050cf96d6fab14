"""Explicit forward / backward kernel schedule of the InternImage backbone on the HIP operators.

Reference: Multi-Task_Pretrain/backbone/intern_image.py ("II") -- StemLayer II:239-276, DownsampleLayer II:279-300, MLPLayer
II:303-333, InternImageLayer II:336-433 (every branch of II:407-427: post- / pre-norm, with / without layer scale, res_post_norm),
InternImageBlock II:436-524 (level norm, level-2 post norms),
InternImage.forward II:690-698 -- and ops_dcnv3/modules/dcnv3.py ("DCNM") DCNv3.forward :318-356.

Everything is channels-last, `rows = N * H * W` rows of C channels:
  * 3x3 convolutions (stem, downsample) = mtp_im2col3x3 + the MFMA NT GEMM; their gradients = TN GEMM (+ unpack) and NT GEMM + col2im;
  * Linear layers = NT GEMMs with fused bias / GELU (+ gelu') / residual epilogues, weight gradients = TN GEMMs with the bias
    gradient as a by-product;
  * depth-wise 3x3, softmax over the 9 points, layer-scale residual: csrc/conv.hip; LayerNorm (+GELU): csrc/layernorm.hip;
  * the DCNv3 core: csrc/dcnv3.hip through mtp_amd.ops_dcnv3 (the reference's own extension interface).
The residual stream is f32; every GEMM operand is the ACT dtype (bf16, or f32 in parity mode).  No torch compute ops: torch
allocates buffers and draws the drop-path masks.
"""
import torch

from . import ops
from .ops_dcnv3 import functions as dcn

F32 = torch.float32


class _Lin:
    __slots__ = ("name", "R", "C", "Rp", "w", "wt", "padded", "bias")


class InternEngine:
    def __init__(self, module, act_dtype=torch.bfloat16):
        self.m = module
        self.act = act_dtype
        self._key = None
        self._ptrs = None
        self._lin = {}
        self._conv = {}
        self._wimg = None
        self._padded = []

    # True: the grouped weight-gradient launches go to a side stream (ops.WgradQueue).  The Linear layers of the 768- / 1536-channel levels are 48-111
    # tiles on 256 CUs: next to them the weight-gradient tiles (one per CU, off the critical path) run on the idle CUs.
    # False: on the current stream; True: a side stream; 2: a side stream of the device's lowest priority (same box: 58.87 / 58.35 / 57.8 ms per step)
    wgrad_side_stream = 2
    wgrad_max_jobs = 0
    wgrad_keep = 3                # bursts that may stay in flight on the side stream when the next one is launched

    def _wgrad_stream(self):
        if not self.wgrad_side_stream:
            return None
        import mtp_amd
        note = mtp_amd.hw_queue_note()        # once per process: the side stream needs a hardware queue of its own (mtp_amd/__init__.py)
        if note:
            import warnings
            warnings.warn(note, RuntimeWarning, stacklevel=2)
        if int(self.wgrad_side_stream) == 2:
            return ops.low_priority_stream(self.dev)
        st = getattr(self, "_wstream", None)
        if st is None or st.device != self.dev:
            st = self._wstream = torch.cuda.Stream(device=self.dev)
        return st

    def mark_images_fresh(self):
        P = self.params()
        self._images_fresh = (self.act,) + tuple((p.data_ptr(), p._version) for p in P.values())
        self._key = None

    def warm_streams(self, device):
        """create and use the weight-gradient side stream now (BackboneEngine.warm_streams: hardware queues go to streams in the order of their first use)"""
        self.dev = torch.device(device)
        st = self._wgrad_stream()
        if st is not None:
            with torch.cuda.stream(st):
                torch.zeros(1, device=self.dev).add_(1.0)
            st.synchronize()

    # ------------------------------------------------------------------ parameters -> GEMM-side images
    def params(self):
        return dict(self.m.named_parameters())

    def prepare_weights(self):
        P = self.params()
        key = (self.act,) + tuple((p.data_ptr(), p._version) for p in P.values())
        if key == self._key:
            return
        ptrs = (self.act,) + tuple(p.data_ptr() for p in P.values())
        if ptrs != self._ptrs:
            self._build(P)
            self._ptrs = ptrs
        if self._wimg is not None and getattr(self, "_images_fresh", None) != key:      # (BackboneEngine.mark_images_fresh: the optimizer launch wrote them)
            self._wimg.refresh()
        self._images_fresh = None
        for L in self._padded:
            ops.pack_rows_padded(P[L.name].detach(), L.w, L.wt)
        for k in range(0, len(self._padded), 12):      # the padded layers' biases: R floats each, 12 per launch
            Ls = self._padded[k:k + 12]
            ops.copy_segments([P[L.name[:-len("weight")] + "bias"].detach() for L in Ls], [L.bias[:L.R] for L in Ls])
        for name, (w2, w2t) in self._conv.items():
            ops.conv3x3_pack(P[name].detach().contiguous(), w2, w2t)
        self._key = key

    def _build(self, P):
        dev = next(iter(P.values())).device
        act = self.act
        entries, self._lin, self._conv, self._padded = [], {}, {}, []
        for name, p in P.items():
            if p.dim() == 2:       # Linear (R, C)
                L = _Lin()
                L.name, (L.R, L.C) = name, p.shape
                L.Rp = ops.pad8(L.R)
                L.padded = L.Rp != L.R
                w2d = p.detach()
                L.bias = None
                if L.padded:
                    # rows not a multiple of 8 (mask head at 12 groups: 108 rows): the GEMM runs on zero-padded images -- Rp output
                    # columns forward (the extra logits are 0 and never read), a 16-byte row pitch for the transposed image and a
                    # contraction over Rp in the data gradient
                    L.w = torch.zeros(L.Rp, L.C, device=dev, dtype=act)
                    L.wt = torch.zeros(L.C, L.Rp, device=dev, dtype=act)
                    L.bias = torch.zeros(L.Rp, device=dev, dtype=F32)
                    self._padded.append(L)
                else:
                    L.wt = torch.empty(L.C, L.R, device=dev, dtype=act)
                    L.w = w2d if act == F32 else torch.empty(L.R, L.C, device=dev, dtype=act)
                    entries.append((w2d, None if act == F32 else L.w, L.wt, False))
                self._lin[name] = L
            elif p.dim() == 4 and p.shape[1] > 1:     # dense 3x3 convolution (Cout, Cin, 3, 3)
                Cout, Cin = p.shape[:2]
                Kp = ops.pad8(9 * Cin)
                self._conv[name] = (torch.empty(Cout, Kp, device=dev, dtype=act), torch.empty(Kp, Cout, device=dev, dtype=act))
        self._wimg = ops.WeightImages(entries, act) if entries else None

    # ------------------------------------------------------------------ helpers
    def _e(self, *shape, dtype=None):
        return torch.empty(*shape, device=self.dev, dtype=dtype or self.act)

    def _linear(self, x, wname, bias, padded_out=False, **kw):
        """x @ W^T + b.  Padded weights: the GEMM writes Rp columns; the result is compacted to R columns unless the caller reads
        it through its pitch (padded_out)."""
        L = self._lin[wname]
        if not L.padded:
            return ops.gemm_nt(x, L.w, self._e(x.shape[0], L.R), bias=bias, **kw)
        out = ops.gemm_nt(x, L.w, self._e(x.shape[0], L.Rp), bias=L.bias, **kw)
        return out if padded_out else ops.copy_rows(out, self._e(x.shape[0], L.R), L.R)

    def _wgrad(self, dy, x, wname, G):
        """dW = dy^T x and db = column sums of dy into G.  Padded layers: dy has Rp columns (the last ones zero); the TN GEMM needs its
        M to be a multiple of 8, so it produces Rp rows and the first R are copied out."""
        L = self._lin[wname]
        bname = wname[:-len("weight")] + "bias"
        if not L.padded:
            self._wq.add(dy, x, G[wname], G[bname])
            return
        tw, tb = self._e(L.Rp, L.C, dtype=F32), torch.zeros(L.Rp, device=self.dev, dtype=F32)
        self._wq.add(dy, x, tw, tb, after=lambda: ops.copy_segments([tw.view(-1)[:L.R * L.C], tb[:L.R]], [G[wname].view(-1), G[bname]]),
                     norm_of=G[wname])      # (the padded rows of dy are zero: the padded image has the norm of the part that is copied out)

    def _ln(self, x, P, key, out_dtype=None, gelu=False):
        rows = x.shape[0]
        y = self._e(rows, x.shape[1], dtype=out_dtype)
        mean, rstd = self._e(rows, dtype=F32), self._e(rows, dtype=F32)
        ops.layernorm_fwd(x, P[key + ".weight"], P[key + ".bias"], y, mean, rstd, eps=1e-6, gelu=gelu)
        return y, mean, rstd

    def _ln_bwd(self, dy, x, mean, rstd, P, G, key, gelu=False):
        dx = self._e(*x.shape, dtype=x.dtype)
        # the dgamma / dbeta partial rows wait in self._ln_parts and are reduced together with the next weight-gradient burst (_ln_flush):
        # three LayerNorms per layer were 240 reduction launches per InternImage-XL step
        ops.layernorm_bwd(dy, x, mean, rstd, P[key + ".weight"], dx, G[key + ".weight"], G[key + ".bias"],
                          beta=P[key + ".bias"] if gelu else None, gelu=gelu, accumulate=True, defer=self._ln_parts)
        return dx

    def _ln_flush(self):
        if self._ln_parts:
            ops.reduce_rows_deferred(self._ln_parts)

    def _to_act(self, t):
        return t if t.dtype == self.act else ops.cast(t, self._e(*t.shape))

    def _conv_fwd(self, x, strides, name, bias, N, H, W, Cin, stride):
        w2, _ = self._conv[name]
        Ho, Wo = ops.conv_out(H, stride), ops.conv_out(W, stride)
        cols = ops.im2col3x3(x, strides, self._e(N * Ho * Wo, w2.shape[1]), N, H, W, Cin, stride)
        y = ops.gemm_nt(cols, w2, self._e(N * Ho * Wo, w2.shape[0]), bias=bias)
        return y, cols, Ho, Wo

    def _conv_bwd(self, dy, cols, name, G, bias_name, dx, strides, N, H, W, Cin, stride, accumulate=False):
        """dy (rows, Cout) ACT.  Weight (+ bias) gradient into G; dx (f32, strided) = / += data gradient when dx is not None"""
        w2, w2t = self._conv[name]
        dw2 = self._e(*w2.shape, dtype=F32)
        self._wq.add(dy, cols, dw2, (G[bias_name] if bias_name else None), after=lambda: ops.conv3x3_unpack_grad(dw2, G[name]),
                     norm_of=G[name])       # (the pad columns of `cols` are zero: same values, re-laid)
        if dx is not None:
            dcols = ops.gemm_nt(dy, w2t, self._e(dy.shape[0], w2.shape[1]))
            ops.col2im3x3(dcols, dx, strides, N, H, W, Cin, stride, accumulate=accumulate)

    def _drop_scales(self, n_layers, N, training):
        rates = self.m.drop_path_rates
        if not training or not any(r > 0 for r in rates):
            return None
        keep = 1.0 - torch.tensor(rates, device=self.dev, dtype=F32).repeat_interleave(2).unsqueeze(1)     # (2 * layers, 1)
        u = torch.rand(2 * n_layers, N, device=self.dev)
        return ((u < keep).to(F32) / keep).contiguous()

    # ------------------------------------------------------------------ one InternImageLayer
    def _vec(self, C, value):
        """cached constant vectors: ones = the layer scale of a configuration without one; zeros = where its (unused) gradient is accumulated"""
        key = (C, value, str(self.dev))
        cache = self.__dict__.setdefault("_vecs", {})
        if key not in cache:
            cache[key] = torch.full((C,), float(value), device=self.dev, dtype=F32)
        return cache[key]

    def _gam(self, pre, k, C):
        return self.P[pre + "gamma%d" % k] if self.m.has_layer_scale else self._vec(C, 1.0)

    def _dgam(self, Gd, pre, k, C):
        return Gd[pre + "gamma%d" % k] if self.m.has_layer_scale else self._vec(C, 0.0)

    def _dcn_fwd(self, pre, xa, N, H, W, C, G):
        """the DCNv3 module (DCNM:318-356) on the ACT rows xa -> h (rows, C) ACT and what its backward needs"""
        P = self.P
        rows = N * H * W
        K = self.m.kernel_size
        Pn = K * K
        d = pre + "dcn."
        xp = self._linear(xa, d + "input_proj.weight", P[d + "input_proj.bias"])
        kd = self.m.dw_kernel_size
        if kd == 3:
            x1c = ops.dwconv3x3_fwd(xa, P[d + "dw_conv.0.weight"], P[d + "dw_conv.0.bias"], self._e(rows, C), N, H, W)
        else:      # InternImage-H/G: a k x k depth-wise kernel (plain kernels)
            x1c = ops.dwconv_fwd(xa, P[d + "dw_conv.0.weight"], P[d + "dw_conv.0.bias"], self._e(rows, C), N, H, W, kd)
        x1, m0, r0 = self._ln(x1c, P, d + "dw_conv.1.1", gelu=True)
        off = self._linear(x1, d + "offset.weight", P[d + "offset.bias"])
        logits = self._linear(x1, d + "mask.weight", P[d + "mask.bias"], padded_out=True)
        mask = ops.softmax_groups_fwd(logits, self._e(rows, G * Pn), G, Pn)
        pad = K // 2
        y = dcn.dcnv3_forward(xp.view(N, H, W, C), off.view(N, H, W, -1), mask.view(N, H, W, -1), K, K, 1, 1, pad, pad, 1, 1, G, C // G,
                              self.m.offset_scale, 256).view(rows, C)
        cfl = y2 = None
        if self.m.center_feature_scale:      # DCNM:209-215: x = x (1 - s) + x_proj s, s = sigmoid(Linear_G(x1)) per group
            cfl = self._linear(x1, d + "center_feature_scale_proj_weight", P[d + "center_feature_scale_proj_bias"], padded_out=True)
            y2 = ops.center_feature_scale_fwd(y, xp, cfl, self._e(rows, C), G)
        h = self._linear(y2 if y2 is not None else y, d + "output_proj.weight", P[d + "output_proj.bias"])
        return h, dict(xa=xa, xp=xp, x1c=x1c, m0=m0, r0=r0, x1=x1, off=off, mask=mask, y=y, h=h, cfl=cfl, y2=y2)

    def _mlp_fwd(self, pre, xab):
        """MLPLayer: fc1 -> GELU -> fc2 (dropout p = 0); fc1 stores gelu'(u) next to gelu(u) for the backward"""
        P = self.P
        rows = xab.shape[0]
        L1 = self._lin[pre + "mlp.fc1.weight"]
        u = self._e(rows, L1.R)
        ug = self._e(rows, L1.R)      # (also when nothing is saved: the GELU epilogues always write their side output -- the inference forward of bench.py's
        ops.gemm_nt(xab, L1.w, u, epi=ops.EPI_BIAS_GELU_DG, bias=P[pre + "mlp.fc1.bias"], aux=ug)      #  `forward_only` found the aux = NULL call of rounds 2-3)
        v = self._linear(u, pre + "mlp.fc2.weight", P[pre + "mlp.fc2.bias"])
        return v, dict(xab=xab, u=u, ug=ug, v=v)

    def _layer_fwd(self, pre, x32, xa, N, H, W, C, G, scales, save):
        """x32 (rows, C) f32 residual stream, xa its ACT copy (None when the previous layer did not produce one) -> (x32', xa', context)"""
        if not self.m.post_norm:
            return self._layer_fwd_pre(pre, x32, N, H, W, C, G, scales, save)
        P = self.P
        rows = N * H * W
        if xa is None:
            xa = self._to_act(x32)
        h, cd = self._dcn_fwd(pre, xa, N, H, W, C, G)
        # x2 = x + s1 * gamma1 * LN1(h): LayerNorm and the layer-scale residual in one pass (round 4: were two launches and a bf16 round trip)
        s1 = scales[0] if scales is not None else None
        x32b, xab = self._e(rows, C, dtype=F32), self._e(rows, C)
        m1, r1 = self._e(rows, dtype=F32), self._e(rows, dtype=F32)
        ops.layernorm_residual_fwd(h, P[pre + "norm1.0.weight"], P[pre + "norm1.0.bias"], x32, self._gam(pre, 1, C), x32b, xab, m1, r1, s1, H * W)
        v, cm = self._mlp_fwd(pre, xab)
        s2 = scales[1] if scales is not None else None
        x32c, xac = self._e(rows, C, dtype=F32), self._e(rows, C)
        m2, r2 = self._e(rows, dtype=F32), self._e(rows, dtype=F32)
        ops.layernorm_residual_fwd(v, P[pre + "norm2.0.weight"], P[pre + "norm2.0.bias"], x32b, self._gam(pre, 2, C), x32c, xac, m2, r2, s2, H * W)
        ctx = None
        if save:
            ctx = dict(m1=m1, r1=r1, m2=m2, r2=r2, s1=s1, s2=s2)
            ctx.update(cd)
            ctx.update(cm)
        return x32c, xac, ctx

    def _layer_fwd_pre(self, pre, x32, N, H, W, C, G, scales, save):
        """the pre-norm branches (II:412-417, 426-427): x += s1 * gamma1 * [res_post_norm1](dcn(norm1(x))); x += s2 * gamma2 * [res_post_norm2](mlp(norm2(x)))"""
        P = self.P
        rows = N * H * W
        rpn = self.m.res_post_norm
        s1 = scales[0] if scales is not None else None
        s2 = scales[1] if scales is not None else None
        xn, ma, ra = self._ln(x32, P, pre + "norm1.0")
        h, cd = self._dcn_fwd(pre, xn, N, H, W, C, G)
        x32b = self._e(rows, C, dtype=F32)
        m1 = r1 = m2 = r2 = None
        if rpn:
            m1, r1 = self._e(rows, dtype=F32), self._e(rows, dtype=F32)
            ops.layernorm_residual_fwd(h, P[pre + "res_post_norm1.0.weight"], P[pre + "res_post_norm1.0.bias"], x32, self._gam(pre, 1, C), x32b, None, m1, r1, s1, H * W)
        else:
            ops.scale_residual_fwd(x32, h, self._gam(pre, 1, C), x32b, None, s1, H * W)
        xn2, mb, rb = self._ln(x32b, P, pre + "norm2.0")
        v, cm = self._mlp_fwd(pre, xn2)
        x32c = self._e(rows, C, dtype=F32)
        if rpn:
            m2, r2 = self._e(rows, dtype=F32), self._e(rows, dtype=F32)
            ops.layernorm_residual_fwd(v, P[pre + "res_post_norm2.0.weight"], P[pre + "res_post_norm2.0.bias"], x32b, self._gam(pre, 2, C), x32c, None, m2, r2, s2, H * W)
        else:
            ops.scale_residual_fwd(x32b, v, self._gam(pre, 2, C), x32c, None, s2, H * W)
        ctx = None
        if save:
            ctx = dict(pre_norm=True, x32=x32, ma=ma, ra=ra, x32b=x32b, mb=mb, rb=rb, m1=m1, r1=r1, m2=m2, r2=r2, s1=s1, s2=s2)
            ctx.update(cd)
            ctx.update(cm)
        return x32c, None, ctx

    def _mlp_bwd(self, pre, c, dv, Gd, res=None, out_dtype=F32):
        """dv (rows, C) ACT = gradient of fc2's output -> gradient of fc1's input ([res] + ..., `out_dtype`); weight gradients queued"""
        rows = dv.shape[0]
        L1, L2 = self._lin[pre + "mlp.fc1.weight"], self._lin[pre + "mlp.fc2.weight"]
        self._wq.add(dv, c["u"], Gd[pre + "mlp.fc2.weight"], Gd[pre + "mlp.fc2.bias"])
        du = ops.gemm_nt(dv, L2.wt, self._e(rows, L1.R), epi=ops.EPI_MUL, aux=c["ug"])
        self._wq.add(du, c["xab"], Gd[pre + "mlp.fc1.weight"], Gd[pre + "mlp.fc1.bias"])
        if res is not None:
            return ops.gemm_nt(du, L1.wt, self._e(rows, L1.C, dtype=F32), epi=ops.EPI_BIAS_RES, res=res)      # + the residual path
        return ops.gemm_nt(du, L1.wt, self._e(rows, L1.C, dtype=out_dtype))

    def _dcn_bwd(self, pre, c, dh, N, H, W, C, G, Gd, res=None):
        """dh (rows, C) ACT = gradient of output_proj's output -> f32 gradient of the module's input ([res] + ...); weight gradients queued"""
        P = self.P
        rows = N * H * W
        K = self.m.kernel_size
        Pn = K * K
        d = pre + "dcn."
        Lo = self._lin[d + "output_proj.weight"]
        self._wq.add(dh, c["y2"] if c.get("y2") is not None else c["y"], Gd[d + "output_proj.weight"], Gd[d + "output_proj.bias"])
        dy = ops.gemm_nt(dh, Lo.wt, self._e(rows, C))
        dxs = dcl = None
        if c.get("cfl") is not None:      # center_feature_scale: dy -> the DCNv3 core's share, the input projection's share (f32, added below) and the gate's logits
            Lc = self._lin[d + "center_feature_scale_proj_weight"]
            dxs, dcl = self._e(rows, C, dtype=F32), self._e(rows, Lc.Rp if Lc.padded else Lc.R)
            dy = ops.center_feature_scale_bwd(dy, c["y"], c["xp"], c["cfl"], self._e(rows, C), dxs, dcl, G)
        pad = K // 2
        Lf, Lm = self._lin[d + "offset.weight"], self._lin[d + "mask.weight"]
        doffa = None
        if self.act != F32:     # grad_offset also as the ACT-dtype, padded GEMM operand, written by the DCNv3 backward itself (round 4: was a cast-and-pad pass)
            dxp, doff, dmask, doffa = dcn.dcnv3_backward_act(c["xp"].view(N, H, W, C), c["off"].view(N, H, W, -1), c["mask"].view(N, H, W, -1), K, K, 1, 1, pad, pad,
                                                             1, 1, G, C // G, self.m.offset_scale, dy.view(N, H, W, C), 256, Lf.Rp)
        else:
            dxp, doff, dmask = dcn.dcnv3_backward(c["xp"].view(N, H, W, C), c["off"].view(N, H, W, -1), c["mask"].view(N, H, W, -1), K, K, 1, 1, pad, pad,
                                                  1, 1, G, C // G, self.m.offset_scale, dy.view(N, H, W, C), 256)
        # offset / mask heads -> d(x1)
        if doffa is None:
            doffa = ops.cast_pad_rows(doff.view(rows, -1), self._e(rows, Lf.Rp)) if (Lf.padded or self.act != F32) else doff.view(rows, -1)
        dlog = ops.softmax_groups_bwd(c["mask"], dmask.view(rows, -1), self._e(rows, Lm.Rp), G, Pn)
        self._wgrad(doffa, c["x1"], d + "offset.weight", Gd)
        self._wgrad(dlog, c["x1"], d + "mask.weight", Gd)
        dx1 = ops.gemm_nt(doffa, Lf.wt, self._e(rows, C, dtype=F32))
        dx1 = ops.gemm_nt(dlog, Lm.wt, self._e(rows, C, dtype=F32), epi=ops.EPI_BIAS_RES, res=dx1)
        if dcl is not None:
            self._wgrad(dcl, c["x1"], d + "center_feature_scale_proj_weight", Gd)
            dx1 = ops.gemm_nt(dcl, self._lin[d + "center_feature_scale_proj_weight"].wt, self._e(rows, C, dtype=F32), epi=ops.EPI_BIAS_RES, res=dx1)
        dx1c = self._ln_bwd(dx1, c["x1c"], c["m0"], c["r0"], P, Gd, d + "dw_conv.1.1", gelu=True)      # (f32 dy, ACT x / dx: no cast pass)
        kd = self.m.dw_kernel_size
        if kd == 3:
            ops.dwconv3x3_bwd_dw(dx1c, c["xa"], Gd[d + "dw_conv.0.weight"], Gd[d + "dw_conv.0.bias"], N, H, W, accumulate=True)
        else:
            ops.dwconv_bwd_dw(dx1c, c["xa"], Gd[d + "dw_conv.0.weight"], Gd[d + "dw_conv.0.bias"], N, H, W, kd)
        # input_proj -> d(x); plus the depth-wise branch and the residual path
        Li = self._lin[d + "input_proj.weight"]
        if dxs is not None:
            ops.axpy(dxp.view(rows, C), dxs)      # + the gate's share of the projected input's gradient
        dxpa = self._to_act(dxp.view(rows, C))
        self._wq.add(dxpa, c["xa"], Gd[d + "input_proj.weight"], Gd[d + "input_proj.bias"])
        if res is not None:
            dxin = ops.gemm_nt(dxpa, Li.wt, self._e(rows, C, dtype=F32), epi=ops.EPI_BIAS_RES, res=res)
        else:
            dxin = ops.gemm_nt(dxpa, Li.wt, self._e(rows, C, dtype=F32))
        if kd == 3:
            ops.dwconv3x3_bwd_dx(dx1c, P[d + "dw_conv.0.weight"], dxin, N, H, W, accumulate=True)
        else:
            ops.dwconv_bwd_dx(dx1c, P[d + "dw_conv.0.weight"], dxin, N, H, W, kd, accumulate=True)
        return dxin

    def _layer_bwd(self, pre, c, dx32, N, H, W, C, G, Gd):
        """dx32 (rows, C) f32: gradient of the layer's output; returns the gradient of its input (f32, new buffer or in place)"""
        if c.get("pre_norm"):
            return self._layer_bwd_pre(pre, c, dx32, N, H, W, C, G, Gd)
        P = self.P
        rows = N * H * W
        # ---- x3 = x2 + s2 * gamma2 * LN2(fc2(gelu(fc1(x2))))
        dv = ops.layernorm_residual_bwd(dx32, c["v"], c["m2"], c["r2"], P[pre + "norm2.0.weight"], P[pre + "norm2.0.bias"], self._gam(pre, 2, C), self._e(rows, C),
                                        Gd[pre + "norm2.0.weight"], Gd[pre + "norm2.0.bias"], self._dgam(Gd, pre, 2, C), c["s2"], H * W, defer=self._ln_parts)
        dx2 = self._mlp_bwd(pre, c, dv, Gd, res=dx32)
        # ---- x2 = x + s1 * gamma1 * LN1(output_proj(dcnv3(...)))
        dh = ops.layernorm_residual_bwd(dx2, c["h"], c["m1"], c["r1"], P[pre + "norm1.0.weight"], P[pre + "norm1.0.bias"], self._gam(pre, 1, C), self._e(rows, C),
                                        Gd[pre + "norm1.0.weight"], Gd[pre + "norm1.0.bias"], self._dgam(Gd, pre, 1, C), c["s1"], H * W, defer=self._ln_parts)
        return self._dcn_bwd(pre, c, dh, N, H, W, C, G, Gd, res=dx2)

    def _branch_bwd(self, pre, k, c, dout, z, mean, rstd, s, HW, C, Gd):
        """gradient of  s * gamma_k * [res_post_norm_k](z)  with respect to z (ACT), parameter gradients accumulated"""
        rows = dout.shape[0]
        if self.m.res_post_norm:
            key = pre + "res_post_norm%d.0" % k
            return ops.layernorm_residual_bwd(dout, z, mean, rstd, self.P[key + ".weight"], self.P[key + ".bias"], self._gam(pre, k, C), self._e(rows, C),
                                              Gd[key + ".weight"], Gd[key + ".bias"], self._dgam(Gd, pre, k, C), s, HW, defer=self._ln_parts)
        return ops.scale_residual_bwd(dout, z, self._gam(pre, k, C), self._e(rows, C), self._dgam(Gd, pre, k, C), s, HW, accumulate=True)

    def _layer_bwd_pre(self, pre, c, dx32, N, H, W, C, G, Gd):
        P = self.P
        HW = H * W
        dv = self._branch_bwd(pre, 2, c, dx32, c["v"], c["m2"], c["r2"], c["s2"], HW, C, Gd)
        dxn2 = self._mlp_bwd(pre, c, dv, Gd, out_dtype=F32)
        dx2 = self._e(*dx32.shape, dtype=F32)
        ops.layernorm_bwd(dxn2, c["x32b"], c["mb"], c["rb"], P[pre + "norm2.0.weight"], dx2, Gd[pre + "norm2.0.weight"], Gd[pre + "norm2.0.bias"], dres=dx32,
                          accumulate=True, defer=self._ln_parts)
        dh = self._branch_bwd(pre, 1, c, dx2, c["h"], c["m1"], c["r1"], c["s1"], HW, C, Gd)
        dxn = self._dcn_bwd(pre, c, dh, N, H, W, C, G, Gd)
        dx = self._e(*dx32.shape, dtype=F32)
        ops.layernorm_bwd(dxn, c["x32"], c["ma"], c["ra"], P[pre + "norm1.0.weight"], dx, Gd[pre + "norm1.0.weight"], Gd[pre + "norm1.0.bias"], dres=dx2,
                          accumulate=True, defer=self._ln_parts)
        return dx

    def _norm_f32(self, x32, key):
        """a LayerNorm of the f32 residual stream itself (the level's closing norm II:516-517, the level-2 post norms II:512-515): f32 in, f32 out"""
        y, mean, rstd = self._ln(x32, self.P, key, out_dtype=F32)
        return y, (x32, mean, rstd, key)

    def _norm_f32_bwd(self, dy, saved, Gd):
        x32, mean, rstd, key = saved
        return self._ln_bwd(dy, x32, mean, rstd, self.P, Gd, key)

    # ------------------------------------------------------------------ whole forward
    def forward(self, img, training=False, need_grad=False, feature_dtype=None):
        m = self.m
        self.dev = img.device
        self.prepare_weights()
        self.P = P = {k: v.detach() for k, v in self.params().items()}
        N, Cin, H, W = img.shape
        if img.dtype not in (F32, torch.bfloat16) or not img.is_contiguous() or (self.act == F32 and img.dtype != F32):
            img = img.float().contiguous()
        save = need_grad
        ckpt = bool(getattr(m, "with_cp", False))
        fdt = feature_dtype or self.act
        ch = m.channels
        # ---- StemLayer: conv3x3 s2 -> LN -> GELU -> conv3x3 s2 -> LN
        y1, cols1, H1, W1 = self._conv_fwd(img, (Cin * H * W, W, 1, H * W), "patch_embed.conv1.weight", P["patch_embed.conv1.bias"], N, H, W, Cin, 2)
        a1, sm1, sr1 = self._ln(y1, P, "patch_embed.norm1.1", gelu=True)
        c2 = ch // 2
        y2, cols2, H2, W2 = self._conv_fwd(a1, (H1 * W1 * c2, W1 * c2, c2, 1), "patch_embed.conv2.weight", P["patch_embed.conv2.bias"], N, H1, W1, c2, 2)
        x32, sm2, sr2 = self._ln(y2, P, "patch_embed.norm2.1", out_dtype=F32)
        xa = self._to_act(x32)
        nl = sum(m.depths)
        scales = self._drop_scales(nl, N, training)
        ctx = dict(stem=(img if save else None, cols1, y1, sm1, sr1, a1, cols2, y2, sm2, sr2, (N, Cin, H, W, H1, W1, H2, W2)), levels=[], scales=scales) if save else None
        feats = []
        Hc, Wc, C = H2, W2, ch
        li = 0
        for i, (depth, G) in enumerate(zip(m.depths, m.groups)):
            lctx = []
            pn_ids = m.post_norm_ids(i)
            pn_saved = {}
            for j in range(depth):
                sc = (scales[2 * li], scales[2 * li + 1]) if scales is not None else None
                if save and ckpt:
                    # with_cp (II:429-430: checkpoint.checkpoint(_inner_forward, x)): keep the layer's input only -- two row tensors instead of twelve and
                    # the 4C-wide GELU pair -- and run its forward again inside the backward (same drop-path factors: they are explicit tensors here)
                    xin32, xina = x32, xa
                    x32, xa, _ = self._layer_fwd("levels.%d.blocks.%d." % (i, j), x32, xa, N, Hc, Wc, C, G, sc, False)
                    c = dict(ckpt=True, x32=xin32, xa=xina, sc=sc)
                else:
                    x32, xa, c = self._layer_fwd("levels.%d.blocks.%d." % (i, j), x32, xa, N, Hc, Wc, C, G, sc, save)
                lctx.append(c)
                li += 1
                if j in pn_ids:       # InternImage-H/G: an extra LayerNorm behind chosen blocks of level 2 (II:512-515)
                    x32, sv = self._norm_f32(x32, "levels.%d.post_norms.%d.0" % (i, pn_ids.index(j)))
                    xa = None
                    pn_saved[j] = sv if save else None
            lnorm = None
            if m.has_level_norm:      # the pre-norm forms (and center_feature_scale models) close a level with its own LayerNorm (II:516-517); the taps are taken behind it
                x32, sv = self._norm_f32(x32, "levels.%d.norm.0" % i)
                xa = None
                lnorm = sv if save else None
            if i in m.out_indices:
                feats.append(ops.tokens_to_nchw(x32, self._e(N, C, Hc, Wc, dtype=fdt), N, Hc, Wc, 0))
            down = None
            if i < len(m.depths) - 1:
                if xa is None:
                    xa = self._to_act(x32)
                pre = "levels.%d.downsample." % i
                yd, colsd, Hn, Wn = self._conv_fwd(xa, (Hc * Wc * C, Wc * C, C, 1), pre + "conv.weight", None, N, Hc, Wc, C, 2)
                x32, dm, dr = self._ln(yd, P, pre + "norm.1", out_dtype=F32)
                xa = self._to_act(x32)
                down = (colsd, yd, dm, dr)
                geom_next = (Hn, Wn, 2 * C)
            if save:
                ctx["levels"].append(dict(layers=lctx, down=down, geom=(Hc, Wc, C, G), norm=lnorm, post_norms=pn_saved))
            if i < len(m.depths) - 1:
                Hc, Wc, C = geom_next
        return feats, ctx

    # ------------------------------------------------------------------ whole backward
    def backward(self, ctx, dfeats, G, need_input_grad=False, on_block_done=None, split_last=False, sqn=None):
        """dfeats: one NCHW cotangent (or None) per entry of out_indices; G: name -> f32 gradient buffer, zero on entry.
        on_block_done(g): every gradient of layer group g (global layer index; InternImage._flat_param_order) and of all later layers is
        complete on the current stream; -1 = the stem (mtp_amd.parallel.GradReducer launches the all-reduces from it)."""
        m = self.m
        P = self.P
        img, cols1, y1, sm1, sr1, a1, cols2, y2, sm2, sr2, (N, Cin, H, W, H1, W1, H2, W2) = ctx["stem"]
        self.dev = cols1.device
        self._wq = wq = ops.WgradQueue(stream=self._wgrad_stream())
        wq.max_jobs = self.wgrad_max_jobs if wq.stream is not None else 0
        wq.sqn = sqn                      # the clipping step's gradient norm as a by-product of the grouped launches (BackboneEngine.backward)
        self.norm_covered = wq.covered
        pending = []        # side-stream mode: the layers whose bursts are in flight (reported once the current stream has waited for them)
        self._ln_parts = []
        taps = {}
        for idx, d in zip([i for i in range(len(m.depths)) if i in m.out_indices], dfeats):
            taps[idx] = d
        dx32 = None
        for i in range(len(m.depths) - 1, -1, -1):
            lv = ctx["levels"][i]
            Hc, Wc, C, Gr = lv["geom"]
            rows = N * Hc * Wc
            if lv["down"] is not None and dx32 is not None:
                # gradient arriving through the downsample of this level: LN -> conv3x3 s2 (no bias)
                colsd, yd, dm, dr = lv["down"]
                pre = "levels.%d.downsample." % i
                dyd = self._ln_bwd(self._to_act(dx32), yd, dm, dr, P, G, pre + "norm.1")
                dprev = self._e(rows, C, dtype=F32)
                self._conv_bwd(dyd, colsd, pre + "conv.weight", G, None, dprev, (Hc * Wc * C, Wc * C, C, 1), N, Hc, Wc, C, 2)
                dx32 = dprev
            elif lv["down"] is not None:      # no gradient arrives through this level's downsample: its overwrite-only weight gradient must read zero
                G["levels.%d.downsample.conv.weight" % i].zero_()
            d = taps.get(i)
            if d is not None:
                d = d if (d.dtype in (F32, torch.bfloat16) and d.is_contiguous()) else d.float().contiguous()
                if dx32 is None:
                    dx32 = ops.nchw_to_tokens(d, self._e(rows, C, dtype=F32), N, Hc, Wc, 0)
                else:
                    ops.axpy(dx32, ops.nchw_to_tokens(d, self._e(rows, C, dtype=F32), N, Hc, Wc, 0))
            if dx32 is None:      # nothing downstream of this level has a gradient: its overwrite-only gradients are not written -- they must read zero
                pre = "levels.%d." % i       # (FlatParams does not clear them, InternImage._overwritten_grads)
                for n, gbuf in G.items():
                    if n.startswith(pre) and m._overwritten_grads(n):
                        gbuf.zero_()
                continue
            if lv.get("norm") is not None:
                dx32 = self._norm_f32_bwd(dx32, lv["norm"], G)
            for j in range(len(lv["layers"]) - 1, -1, -1):
                c = lv["layers"][j]
                if lv.get("post_norms") and lv["post_norms"].get(j) is not None:
                    dx32 = self._norm_f32_bwd(dx32, lv["post_norms"][j], G)
                if c.get("ckpt"):      # with_cp: the layer's forward once more, this time keeping what its backward needs
                    c = self._layer_fwd("levels.%d.blocks.%d." % (i, j), c["x32"], c["xa"], N, Hc, Wc, C, Gr, c["sc"], True)[2]
                dx32 = self._layer_bwd("levels.%d.blocks.%d." % (i, j), c, dx32, N, Hc, Wc, C, Gr, G)
                lv["layers"][j] = None
                # the weight gradients are queued and launched a few layers at a time (ops.WgradQueue: edge tiles for the 192- / 384-channel
                # levels and the offset / mask heads, the contraction of the 131072- / 32768-token levels cut into pieces inside the launch);
                # the layer is reported once they are out
                if j == 0 or wq.should_flush() or not wq.jobs:
                    wq.flush()
                    self._ln_flush()
                    if wq.stream is None:
                        if on_block_done is not None:
                            on_block_done(sum(m.depths[:i]) + j)
                    else:
                        pending.append((sum(m.depths[:i]) + j, wq.launched))
                        wq.wait(keep=self.wgrad_keep)
                        while pending and pending[0][1] <= wq.launched - len(wq.inflight):      # bursts the current stream has waited for
                            g = pending.pop(0)[0]
                            if on_block_done is not None:
                                on_block_done(g)
        if dx32 is None:
            return None
        # ---- stem backward
        c2 = m.channels // 2
        dy2 = self._ln_bwd(self._to_act(dx32), y2, sm2, sr2, P, G, "patch_embed.norm2.1")
        da1 = self._e(N * H1 * W1, c2, dtype=F32)
        self._conv_bwd(dy2, cols2, "patch_embed.conv2.weight", G, "patch_embed.conv2.bias", da1, (H1 * W1 * c2, W1 * c2, c2, 1), N, H1, W1, c2, 2)
        dy1 = self._ln_bwd(self._to_act(da1), y1, sm1, sr1, P, G, "patch_embed.norm1.1", gelu=True)
        dimg = self._e(N, Cin, H, W, dtype=F32) if need_input_grad else None
        self._conv_bwd(dy1, cols1, "patch_embed.conv1.weight", G, "patch_embed.conv1.bias", dimg, (Cin * H * W, W, 1, H * W), N, H, W, Cin, 2)
        wq.flush()      # the stem's two weight gradients
        self._ln_flush()
        wq.wait()
        if on_block_done is not None:
            on_block_done(-1)
        return dimg
