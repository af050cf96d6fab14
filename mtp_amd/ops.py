"""Thin torch-tensor wrappers over the C ABI (include/mtp_hip.h).  PyTorch is plumbing here: device memory,
the current HIP stream, nothing else.  Every function launches hand-written gfx950 kernels from libmtp_hip.so
on torch's current stream; there is no CPU / eager fallback -- CPU tensors raise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_DG, EPI_BIAS_RES, EPI_DGELU, EPI_MUL, MTP_BF16, MTP_F32, GemmArgs,  # noqa: F401
                   check)

_DT = {torch.float32: MTP_F32, torch.bfloat16: MTP_BF16}
def lib():
    return _lib.load()


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("mtp_amd ops take float32 / bfloat16 tensors, got %s" % t.dtype)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mtp_amd ops run only on an MI355X device tensor (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("mtp_amd ops need contiguous tensors")
    # every launch goes to the current stream of the CURRENT device (_s): a tensor living on another GPU (a module moved with
    # .to('cuda:1') while cuda:0 is current, as multi-GPU runners do) would be dereferenced by the wrong device, unordered
    # against its own stream.  One process per GPU with torch.cuda.set_device(local_rank) is the supported setup.
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("mtp_amd ops: tensor on %s but the current device is cuda:%d -- call torch.cuda.set_device(%d) (or use "
                           "`with torch.cuda.device(...)`) before running the backbone" % (t.device, torch.cuda.current_device(), t.device.index))
    return t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise TypeError("expected a float32 tensor")
    return _p(t)


# ------------------------------------------------------------------------------------------------ GEMM
def _nt_args(a, w, out, epi, bias, bias_mod, res, res_mod, rowscale, rows_per_sample, aux, variant, n=None):
    M, K = a.shape
    N = w.shape[0] if n is None else n       # n: only the first n rows of w / columns of a wider `out` (row stride out.shape[1])
    assert w.shape[1] == K and out.shape[0] == M and out.shape[1] >= N and w.shape[0] >= N and a.dtype == w.dtype
    assert n is not None or out.shape[1] == N
    g = GemmArgs()
    g.A, g.B, g.C = _p(a), _p(w), _p(out)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, out.shape[1]
    g.in_dtype, g.out_dtype, g.epilogue = _dt(a), _dt(out), epi
    g.bias, g.bias_mod = _f32(bias), bias_mod
    g.res, g.res_ld, g.res_mod = _f32(res), (res.shape[-1] if res is not None else 0), res_mod
    g.rowscale, g.rows_per_sample = _f32(rowscale), rows_per_sample
    g.aux, g.aux_ld = _p(aux), (aux.shape[-1] if aux is not None else 0)
    if aux is not None:
        assert aux.dtype == out.dtype and tuple(aux.shape) == (M, N) and out.shape[1] == N
    g.split_k, g.variant = 1, variant
    return g


def gemm_nt(a, w, out, epi=EPI_BIAS, bias=None, bias_mod=0, res=None, res_mod=0, rowscale=None, rows_per_sample=0,
            aux=None, variant=0, n=None):
    """out (M,N) = epilogue(a (M,K) @ w (N,K)^T)."""
    g = _nt_args(a, w, out, epi, bias, bias_mod, res, res_mod, rowscale, rows_per_sample, aux, variant, n)
    check(lib().mtp_gemm_nt(C.byref(g), _s()), "mtp_gemm_nt")
    return out


def gemm_nt_tile(a, w, out, epi=EPI_BIAS, bias=None, bias_mod=0, res=None, res_mod=0, rowscale=None, rows_per_sample=0,
                 aux=None, variant=0):
    """tile width (256 / 128) of the kernel family gemm_nt(...) runs these arguments on (no launch)"""
    g = _nt_args(a, w, out, epi, bias, bias_mod, res, res_mod, rowscale, rows_per_sample, aux, variant)
    rc = lib().mtp_gemm_nt_tile(C.byref(g))
    if rc < 0:
        check(rc, "mtp_gemm_nt_tile")
    return rc


def pick_split_k(M, N, K):
    """enough (tile, split) work items for >= 2 resident workgroups on each of the 256 CUs"""
    # measured on MI355X (tools/probes/ab_tn.py): ~512 work items (2 per CU) is the sweet spot; odd splits are consistently slower
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    split = max(1, min(16, round(512 / tiles)))
    if split > 1 and split % 2:
        split += 1
    return max(1, min(split, K // 1024))


def effective_split_k(K, dtype, split):
    """the split launch_tn (csrc/gemm.hip) really runs: at most one K tile (64 bf16 / 32 f32 rows) per split, and no empty
    splits -- the number of partial tiles the workspace holds and a deferred reduction has to sum"""
    k_tiles = -(-K // (64 if dtype == torch.bfloat16 else 32))
    split = max(1, min(int(split), k_tiles))
    per = -(-k_tiles // split)
    return -(-k_tiles // per)


def gemm_tn(a, b, out, split_k=None, use_workspace=True, variant=0, colsum=None, defer=None, m=None):
    """out (M,N) f32 = a (K,M)^T @ b (K,N)   (weight gradient dW = dY^T X).
    colsum (M,) f32, optional: colsum += a.sum(0) -- the bias gradient, produced by the same pass over dY.
    m: use only the first m columns of a wider `a` (row stride a.shape[1])."""
    K, M = a.shape
    lda = M
    if m is not None:
        assert m <= M
        M = m
    N = b.shape[1]
    assert b.shape[0] == K and out.dtype == torch.float32 and out.numel() == M * N and a.dtype == b.dtype
    g = GemmArgs()
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.numel() == M and colsum.is_contiguous()
        g.colsum = _p(colsum)
    g.A, g.B, g.C = _p(a), _p(b), _p(out)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, N, N
    g.in_dtype, g.out_dtype, g.epilogue = _dt(a), MTP_F32, EPI_BIAS
    g.split_k = effective_split_k(K, a.dtype, pick_split_k(M, N, K) if split_k is None else split_k)
    g.variant = variant
    if g.split_k > 1 and use_workspace:
        ws = torch.empty(g.split_k * M * N, device=out.device, dtype=torch.float32)   # split-K partials (summed by the callee)
        g.aux = _p(ws)
        if defer is not None:     # ... or later, together with the other weight gradients of the block (sum_partials)
            g.defer_sum = 1
            defer.append((ws, out, M * N, g.split_k))
    check(lib().mtp_gemm_tn(C.byref(g), _s()), "mtp_gemm_tn")
    return out


MAX_GROUPED = 32      # MTP_MAX_GROUPED_GEMMS


def grouped_tiles(M, N):
    """256 x 256 output tiles (edge tiles included) of one problem in mtp_gemm_tn_grouped, or 0 when it cannot take the problem"""
    return -(-M // 256) * -(-N // 256) if M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8 else 0


def grouped_splits(K, tiles):
    """pieces of the contraction for a problem of `tiles` output tiles: 1 unless the contraction is much longer than a transformer block's
    (InternImage's 192- / 384-channel levels: M, N <= 1536 over 131072 / 32768 tokens; the ViT FPN's second deconvolution: 64 tiles over
    4 x the tokens) -- then pieces of >= 4096 rows, at most enough of them for one round of the 256 CUs, each piece an own workgroup"""
    if K < 16384:
        return 1
    return max(1, min(64, K // 4096, 256 // tiles))


_low_streams = {}


def low_priority_stream(device):
    """a side stream of the device's lowest priority (mtp_stream_create_low_priority), wrapped for torch; one per device, kept for the process"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _low_streams.get(idx)
    if st is None:
        h = C.c_void_p()
        with torch.cuda.device(idx):
            check(lib().mtp_stream_create_low_priority(C.byref(h)), "mtp_stream_create_low_priority")
        st = _low_streams[idx] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
    return st


_mask_streams = {}


def cu_mask_words(kind, part, parts=2, cus_per_xcc=32, xccs=8):
    """the CU bit mask (list of 32-bit words) of partition `part` of `parts`: bit i = XCC i % xccs, CU j = i // xccs of that XCC (gfx950, SPX mode).
    kind "interleaved": CU j belongs to partition j % parts (whole shader engines); "blocked": j * parts // cus_per_xcc (a slice of every engine).
    Every partition holds cus_per_xcc / parts CUs of EVERY XCC: the dispatcher hands each XCC every eighth workgroup whatever the mask says."""
    words = [0] * (cus_per_xcc * xccs // 32)
    for i in range(cus_per_xcc * xccs):
        j = i // xccs
        mine = (j % parts == part) if kind == "interleaved" else (j * parts // cus_per_xcc == part)
        if mine:
            words[i // 32] |= 1 << (i % 32)
    return words


def cu_mask_stream(device, words):
    """a stream restricted to the CUs of `words` (mtp_stream_create_cu_mask), wrapped for torch; one per (device, mask), kept for the process"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, tuple(words))
    st = _mask_streams.get(key)
    if st is None:
        h = C.c_void_p()
        arr = (C.c_uint32 * len(words))(*words)
        with torch.cuda.device(idx):
            check(lib().mtp_stream_create_cu_mask(arr, len(words), C.byref(h)), "mtp_stream_create_cu_mask")
        st = _mask_streams[key] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
    return st


class WgradQueue:
    """Deferred weight gradients dW = dY^T X (+ bias gradient = column sums of dY): the weight gradients of a transformer block
    depend only on tensors the backward pass has anyway, so they are collected and launched together -- ONE
    mtp_gemm_tn_grouped launch whose 256 x 256 tiles fill the 256 CUs in whole rounds (4 ViT-L blocks = 768 tiles = 3 rounds),
    each tile with the full contraction: no split-K partials, no reduction launches.  Problems of a few tiles with a very long
    contraction are cut into pieces inside the same launch (grouped_splits; the kernel's own split_k + one reduction launch).  Problems
    the grouped kernel cannot take (f32 parity mode, sizes that are not multiples of 8 / 128) run immediately through gemm_tn."""

    def __init__(self, cus=256, variant=0, stream=None):
        self.jobs, self.tiles, self.cus, self.variant = [], 0, cus, variant
        # stream: launch on this (side) stream instead of the current one.  The weight gradients are off the backward pass's
        # critical path; next to the chain of data-gradient GEMMs (224 or 672 tiles on 256 CUs: 12.5 % of the CUs idle in the last
        # round) their tiles fill the idle CUs.  flush() orders the launch after everything issued so far; wait() orders the
        # current stream after the launches (call it before the gradients are consumed).
        self.stream, self.inflight = stream, []
        self.max_jobs = 0    # > 0: a burst goes out as soon as it holds this many problems (A/B of the burst size next to the side stream)
        self.launched = 0    # side-stream launches so far; launched - len(inflight) of them have been waited for by the current stream
        self.after = []      # callables run right after the launch (on its stream): e.g. copying a padded result into the gradient buffer
        # squared-norm by-product (round 6): with `sqn` (a 1-element f32 device tensor) every unsplit problem of a grouped launch also adds sum(dW^2) to it; `covered`
        # lists the gradient tensors whose norm has been accounted for that way (the clipping step sums the rest of the buffer only: FlatAdamW.step)
        self.sqn, self.covered = None, []

    def add(self, dy, x, dw, colsum=None, after=None, norm_of=None, norm_ok=True):
        """norm_of: the gradient tensor dw's values end up in when that is not dw itself (a re-laid copy made by `after`); norm_ok=False: `after` changes the values"""
        K, M = dy.shape
        N = x.shape[1]
        t = grouped_tiles(M, N)
        # the grouped kernel's own limits (gemm_tn_p8.hip: 32-bit DMA offsets -> K * ld * 2 < 4 GiB per operand; 16-byte aligned bases):
        # a problem outside them runs now through gemm_tn instead of failing the whole deferred launch several blocks later
        fits = (K * M * 2 < (1 << 32) and K * N * 2 < (1 << 32) and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dw.data_ptr() % 16 == 0
                and dy.is_contiguous() and x.is_contiguous())
        if t == 0 or K % 128 or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not fits:
            gemm_tn(dy, x, dw, colsum=colsum)
            if after is not None:
                after()
            return False
        assert x.shape[0] == K and dw.dtype == torch.float32 and dw.numel() == M * N and dw.is_contiguous()
        splits = grouped_splits(K, t)
        part = torch.empty(splits, M, N, device=dw.device, dtype=torch.float32) if splits > 1 else None
        use_sqn = self.sqn is not None and splits == 1 and norm_ok
        self.jobs.append((dy, x, dw, colsum, splits, part, use_sqn))     # (the references keep dY / X / the partial images alive until the launch)
        if use_sqn:
            self.covered.append(dw if norm_of is None else norm_of)
        if after is not None:
            self.after.append(after)
        self.tiles += t * splits
        return True

    def should_flush(self, next_tiles=0, next_jobs=4):
        """whole rounds of the CUs, or no room for another block's problems"""
        if not self.jobs:
            return False
        if len(self.jobs) + next_jobs > MAX_GROUPED or (self.max_jobs and len(self.jobs) >= self.max_jobs):
            return True
        rounds = -(-self.tiles // self.cus)
        return self.tiles / (rounds * self.cus) >= 0.93

    def flush(self):
        if self.stream is None or not self.jobs:
            return self._launch()
        ready = torch.cuda.Event()
        ready.record()
        held = list(self.jobs)           # dY / X / partial images stay referenced until the current stream has waited for the launch
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self._launch()
            done = torch.cuda.Event()
            done.record()
        self.inflight.append((done, held))
        self.launched += 1

    def wait(self, keep=0):
        """the current stream waits for the side-stream launches, except the `keep` most recent ones"""
        while len(self.inflight) > keep:
            done, _ = self.inflight.pop(0)
            torch.cuda.current_stream().wait_event(done)

    def _launch(self):
        while self.jobs:
            chunk, self.jobs = self.jobs[:MAX_GROUPED], self.jobs[MAX_GROUPED:]
            arr = (GemmArgs * len(chunk))()
            for g, (dy, x, dw, cs, splits, part, use_sqn) in zip(arr, chunk):
                K, M = dy.shape
                N = x.shape[1]
                g.A, g.B, g.C = _p(dy), _p(x), _p(dw)
                g.M, g.N, g.K = M, N, K
                g.lda, g.ldb, g.ldc = M, N, N
                g.in_dtype, g.out_dtype, g.variant = MTP_BF16, MTP_F32, self.variant
                if splits > 1:
                    g.split_k, g.aux = splits, _p(part)
                if use_sqn:
                    g.workspace, g.workspace_bytes = _p(self.sqn), 4
                if cs is not None:
                    assert cs.dtype == torch.float32 and cs.numel() == M and cs.is_contiguous()
                    g.colsum = _p(cs)
            check(lib().mtp_gemm_tn_grouped(arr, len(chunk), _s()), "mtp_gemm_tn_grouped")
        after, self.after = self.after, []
        for fn in after:
            fn()
        self.tiles = 0


def sum_partials(jobs):
    """finish the deferred split-K reductions collected by gemm_tn(..., defer=jobs): one launch per 12 GEMMs"""
    while jobs:
        chunk, jobs[:] = jobs[:12], jobs[12:]
        n = len(chunk)
        P = C.c_void_p * n
        parts, outs = P(*[j[0].data_ptr() for j in chunk]), P(*[j[1].data_ptr() for j in chunk])
        numel, splits = (C.c_int64 * n)(*[j[2] for j in chunk]), (C.c_int * n)(*[j[3] for j in chunk])
        check(lib().mtp_sum_partials_batch(parts, outs, numel, splits, n, _s()), "mtp_sum_partials_batch")


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, y, mean=None, rstd=None, eps=1e-6, gelu=False):
    rows, Cc = x.shape
    check(lib().mtp_layernorm_fwd(_p(x), _dt(x), _f32(gamma), _f32(beta), _p(y), _dt(y), _f32(mean), _f32(rstd),
                                  rows, Cc, eps, int(gelu), _s()), "mtp_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, beta=None, gelu=False, dres=None, extra=None,
                  dx_copy=None, copy_scale=None, rows_per_sample=0, accumulate=False, defer=None, win_add=None, grid=None):
    """dx = [dres] + [extra] + LN'(dy); dgamma/dbeta (C,) f32 are overwritten (or accumulated into).  With `defer` (a list) the
    per-workgroup partials are kept and appended to it instead of being reduced: reduce_rows_deferred(defer) finishes several
    LayerNorms' parameter gradients in one launch."""
    rows, Cc = x.shape
    # (an in-kernel f32-atomic accumulation of dgamma / dbeta was measured: 512 workgroups hitting the same 2C addresses took
    #  the kernel from 61 to 106 us; per-workgroup partials + two 8 us reductions are faster)
    nblk = lib().mtp_layernorm_bwd_partial_rows(rows)
    part = torch.empty(nblk, 2 * Cc, device=x.device, dtype=torch.float32)     # [dgamma partials | dbeta partials] per workgroup
    if win_add is not None:
        # dy_eff = dy + win_add[window(row)]: the RVSA sampling heads' input gradient, added per 7 x 7 window of the (B, Hp, Wp) token grid
        B, Hp, Wp = grid
        nh, nw = rvsa_windows(Hp, Wp)
        assert not gelu and tuple(win_add.shape) == (B * nh * nw, Cc) and B * Hp * Wp == rows
        check(lib().mtp_layernorm_bwd_win(_p(dy), _dt(dy), _p(x), _dt(x), _f32(mean), _f32(rstd), _f32(gamma), _f32(dres), _f32(extra), _p(dx), _dt(dx),
                                          _p(dx_copy), (_dt(dx_copy) if dx_copy is not None else 0), _f32(copy_scale), rows_per_sample,
                                          part.data_ptr(), part.data_ptr() + 4 * Cc, 2 * Cc, rows, Cc, _f32(win_add), B, Hp, Wp, _s()), "mtp_layernorm_bwd_win")
    else:
        check(lib().mtp_layernorm_bwd(_p(dy), _dt(dy), _p(x), _dt(x), _f32(mean), _f32(rstd), _f32(gamma), _f32(beta), int(gelu),
                                      _f32(dres), _f32(extra), _p(dx), _dt(dx), _p(dx_copy), (_dt(dx_copy) if dx_copy is not None else 0),
                                      _f32(copy_scale), rows_per_sample, part.data_ptr(), part.data_ptr() + 4 * Cc, 2 * Cc, rows, Cc, _s()),
              "mtp_layernorm_bwd")
    # weight and bias gradient adjacent in one buffer (the flat gradient buffer of mtp_amd.parallel) -> ONE reduction launch
    if defer is not None and _adjacent(dgamma, dbeta) and dgamma.numel() == Cc:
        defer.append((part, dgamma, accumulate))
    else:
        _reduce_pair(part, Cc, dgamma, dbeta, accumulate)
    return dx


def layernorm_residual_fwd(h, gamma, beta, x, layer_scale, out, out_act, mean, rstd, sample_scale=None, rows_per_sample=0, eps=1e-6):
    """out (f32) = x + sample_scale[row / rows_per_sample] * layer_scale * LayerNorm(h); out_act = its ACT copy (InternImage's post-norm residual)"""
    rows, Cc = h.shape
    check(lib().mtp_layernorm_residual_fwd(_p(h), _dt(h), _f32(gamma), _f32(beta), _f32(x), _f32(layer_scale), _f32(sample_scale), rows_per_sample,
                                           _f32(out), _p(out_act), _f32(mean), _f32(rstd), rows, Cc, eps, _s()), "mtp_layernorm_residual_fwd")
    return out


def layernorm_residual_bwd(dout, h, mean, rstd, gamma, beta, layer_scale, dh, dgamma, dbeta, dls, sample_scale=None, rows_per_sample=0, defer=None):
    """dh = LN'(s * layer_scale * dout); dgamma / dbeta / dls (C,) f32 ACCUMULATE the three parameter gradients (deferred with `defer`)"""
    rows, Cc = h.shape
    nblk = lib().mtp_layernorm_bwd_partial_rows(rows)
    part = torch.empty(nblk, 3 * Cc, device=h.device, dtype=torch.float32)
    check(lib().mtp_layernorm_residual_bwd(_f32(dout), _p(h), _dt(h), _f32(mean), _f32(rstd), _f32(gamma), _f32(beta), _f32(layer_scale), _f32(sample_scale),
                                           rows_per_sample, _p(dh), part.data_ptr(), rows, Cc, _s()), "mtp_layernorm_residual_bwd")
    pair = _adjacent(dgamma, dbeta) and dgamma.numel() == Cc
    if defer is not None and pair:
        defer.append((part[:, :2 * Cc], dgamma, True))
        defer.append((part[:, 2 * Cc:], dls, True))
    else:
        _reduce_pair(part[:, :2 * Cc], Cc, dgamma, dbeta, True)
        reduce_rows(part[:, 2 * Cc:], dls, True)
    return dh


REDUCE_BATCH_MAX = 32


def reduce_rows_deferred(items):
    """items: (part (rows, cols) f32, out (first of `cols` contiguous floats), accumulate[, (R, C)]) tuples queued by layernorm_bwd(defer=...) /
    rvsa_attn_bwd(defer=...); equally-shaped ones go out together, <= REDUCE_BATCH_MAX per launch (mtp_reduce_rows_batched_f32; with (R, C) the
    transposed form mtp_reduce_rows_t_batched_f32: column a * C + b -> out[b * R + a]).  Empties the list."""
    groups = {}
    for it in items:
        part, out, acc = it[:3]
        tr = it[3] if len(it) > 3 else None
        groups.setdefault((tuple(part.shape), part.stride(0), bool(acc), tr), []).append((part, out))
    for ((rows, cols), ld, acc, tr), g in groups.items():
        for i0 in range(0, len(g), REDUCE_BATCH_MAX):
            chunk = g[i0:i0 + REDUCE_BATCH_MAX]
            n = len(chunk)
            P = C.c_void_p * n
            if tr is None:
                check(lib().mtp_reduce_rows_batched_f32(P(*[c[0].data_ptr() for c in chunk]), P(*[c[1].data_ptr() for c in chunk]), n, ld, rows, cols,
                                                        int(acc), _s()), "mtp_reduce_rows_batched_f32")
            else:
                check(lib().mtp_reduce_rows_t_batched_f32(P(*[c[0].data_ptr() for c in chunk]), P(*[c[1].data_ptr() for c in chunk]), n, ld, rows, tr[0], tr[1],
                                                          int(acc), _s()), "mtp_reduce_rows_t_batched_f32")
    del items[:]


def reduce_rows(part, out, accumulate=False):
    """out (+)= part.sum(0); part (rows, cols) f32, a column slice of a wider buffer is fine (row stride = part.stride(0))."""
    if not part.is_cuda:
        raise RuntimeError("mtp_amd ops run only on an MI355X device tensor (no CPU fallback)")
    assert part.dim() == 2 and part.stride(1) == 1 and part.dtype == torch.float32
    rows, cols = part.shape
    assert out.numel() == cols
    check(lib().mtp_reduce_rows_f32(part.data_ptr(), part.stride(0), _f32(out), rows, cols, int(accumulate), _s()), "mtp_reduce_rows_f32")
    return out


def _adjacent(a, b):
    """b starts right where a ends, in the same storage (true for consecutive parameters' gradients in the flat buffer)"""
    return (a.is_contiguous() and b.is_contiguous() and b.data_ptr() == a.data_ptr() + 4 * a.numel()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())


def _reduce_pair(part, n0, out0, out1, accumulate):
    """part (rows, n0 + n1) f32: column sums of [:, :n0] -> out0 and of [:, n0:] -> out1; one launch when out1 follows out0 in memory"""
    if _adjacent(out0, out1) and out0.numel() == n0:
        reduce_rows(part, out0.as_strided((part.shape[1],), (1,)), accumulate)
    else:
        reduce_rows(part[:, :n0], out0, accumulate)
        reduce_rows(part[:, n0:], out1, accumulate)


def copy_segments(srcs, dsts):
    """dsts[i].copy_(srcs[i]) for up to 12 contiguous f32 tensors in one launch."""
    n = len(srcs)
    assert n == len(dsts) and 0 < n <= 12
    for s, d in zip(srcs, dsts):
        assert s.dtype == d.dtype == torch.float32 and s.is_contiguous() and d.is_contiguous() and s.numel() == d.numel()
    P = C.c_void_p * n
    sp, dp = P(*[s.data_ptr() for s in srcs]), P(*[d.data_ptr() for d in dsts])
    cnt = (C.c_int64 * n)(*[s.numel() for s in srcs])
    check(lib().mtp_copy_segments_f32(sp, dp, cnt, n, _s()), "mtp_copy_segments_f32")


def colsum(dy, out, accumulate=False):
    M, N = dy.shape
    fn = lib().mtp_colsum_acc if accumulate else lib().mtp_colsum
    check(fn(_p(dy), _dt(dy), N, _f32(out), M, N, _s()), "mtp_colsum")
    return out


# ------------------------------------------------------------------------------------------------ layout
def patchify(img, cols, P=16):
    B, Cin, H, W = img.shape
    check(lib().mtp_patchify(_f32(img), _p(cols), _dt(cols), B, Cin, H, W, P, _s()), "mtp_patchify")
    return cols


def preprocess_patchify(img_u8, cols, P, mean, std, bgr_to_rgb=True, pad_divisor=32, pad_value=0.0):
    """uint8 (B,H,W,3) HWC image batch -> normalised patch rows `cols` (B*Hp*Wp, 3*P*P): MTP_DataPreprocessor's image
    path (channel flip, (x - mean) / std, pad to a multiple of pad_divisor) fused with the PatchEmbed im2col."""
    if not img_u8.is_cuda:
        raise RuntimeError("mtp_amd ops run only on an MI355X device tensor (no CPU fallback)")
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[-1] == 3 and img_u8.is_contiguous()
    B, H, W, _ = img_u8.shape
    Hp, Wp = padded_grid(H, W, P, pad_divisor)
    assert tuple(cols.shape) == (B * Hp * Wp, 3 * P * P)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    check(lib().mtp_preprocess_patchify(img_u8.data_ptr(), _p(cols), _dt(cols), B, H, W, P, pad_divisor, m3, s3, int(bool(bgr_to_rgb)),
                                        float(pad_value), _s()), "mtp_preprocess_patchify")
    return cols


def padded_grid(H, W, P, pad_divisor):
    """patch grid (Hp, Wp) of an H x W image padded bottom/right to a multiple of pad_divisor"""
    Hpad, Wpad = -(-H // pad_divisor) * pad_divisor, -(-W // pad_divisor) * pad_divisor
    assert Hpad % P == 0 and Wpad % P == 0
    return Hpad // P, Wpad // P


def unpatchify(cols, dimg, P=16):
    B, Cin, H, W = dimg.shape
    check(lib().mtp_unpatchify(_p(cols), _dt(cols), _f32(dimg), B, Cin, H, W, P, _s()), "mtp_unpatchify")
    return dimg


def cast(src, dst):
    assert src.numel() == dst.numel()
    check(lib().mtp_cast(_p(src), _dt(src), _p(dst), _dt(dst), src.numel(), _s()), "mtp_cast")
    return dst


def transpose_cast(src, dst):
    R, Cc = src.shape
    assert tuple(dst.shape) == (Cc, R)
    check(lib().mtp_transpose_cast(_f32(src), _p(dst), _dt(dst), R, Cc, _s()), "mtp_transpose_cast")
    return dst


class WeightImages:
    """Descriptor table (in device memory) for mtp_weight_images: every GEMM-side image of the f32 master weights is
    refreshed by ONE launch per optimizer step.  entries: (src (R,C) f32, w or None, wt or None, f32_out)."""

    def __init__(self, entries, act_dtype):
        n = len(entries)
        arr = (_lib.WimgDesc * n)()
        tile0 = 0
        self.keep = []
        for i, (src, w, wt, f32_out) in enumerate(entries):
            assert src.dim() == 2 and src.dtype == torch.float32
            R, Cc = src.shape
            want = torch.float32 if f32_out else act_dtype
            for img in (w, wt):
                assert img is None or (img.dtype == want and img.numel() == R * Cc)
            d = arr[i]
            d.src, d.w, d.wt = _f32(src), _p(w), _p(wt)
            d.R, d.C, d.tile0, d.f32_out = R, Cc, tile0, int(bool(f32_out))
            tile0 += ((R + 63) // 64) * ((Cc + 63) // 64)
            self.keep.append((src, w, wt))
        self.n, self.total_tiles, self.act = n, tile0, _DT[act_dtype]
        self.entries, self.act_dtype = list(entries), act_dtype
        dev = entries[0][0].device
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def refresh(self):
        check(lib().mtp_weight_images(self.table.data_ptr(), self.n, self.total_tiles, self.act, _s()), "mtp_weight_images")


class AdamWImages:
    """Descriptor table for mtp_adamw_weight_images: ONE launch updates every parameter of the flat buffers (AdamW + gradient clipping, as adamw_flat) and writes the
    GEMM-side images of the parameters that have them from the same registers -- the separate image pass read each f32 master once more.
    flat: mtp_amd.parallel.FlatParams; wimg: the engine's WeightImages (its sources must be the flat parameters themselves); wd_of(name) -> weight decay.
    None is returned by build() when an image source is not a whole flat parameter (layer scale: the images are made from scaled temporaries)."""

    @staticmethod
    def build(flat, wimg, wd_of):
        by_ptr = {}
        for src, w, wt, f32_out in (wimg.entries if wimg is not None else []):
            by_ptr[src.data_ptr()] = (src, w, wt, f32_out)
        base = flat.data.data_ptr()
        rows = []
        used = 0
        for n in flat.names:
            if flat.groups[n] is None:
                continue
            off = flat.offsets[n]
            numel = 1
            for d in flat.shapes[n]:
                numel *= d
            ent = by_ptr.get(base + 4 * off)
            if ent is not None:
                src, w, wt, f32_out = ent
                if src.numel() != numel:
                    return None
                R, Cc = src.shape
                used += 1
            else:
                w = wt = None
                f32_out = False
                padded = (numel + 63) // 64 * 64          # FlatParams pads every parameter to 64 elements: the tail is zero and stays zero
                R, Cc = padded // 64, 64
            rows.append((base + 4 * off, w, wt, R, Cc, f32_out, float(wd_of(n))))
        if used != len(by_ptr):
            return None               # an image whose source is not a flat parameter
        self = AdamWImages()
        arr = (_lib.WimgDesc * len(rows))()
        tile0 = 0
        self.keep = []
        for i, (ptr, w, wt, R, Cc, f32_out, wd) in enumerate(rows):
            d = arr[i]
            d.src, d.w, d.wt = ptr, _p(w), _p(wt)
            d.R, d.C, d.tile0, d.f32_out, d.wd = R, Cc, tile0, int(bool(f32_out)), wd
            tile0 += ((R + 63) // 64) * ((Cc + 63) // 64)
            self.keep.append((w, wt))
        self.n, self.total_tiles = len(rows), tile0
        self.act = _DT[wimg.act_dtype] if wimg is not None else MTP_BF16
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(flat.data.device)
        self.flat = flat
        return self

    def step(self, m, v, hyper, sqn, max_norm, grad_scale):
        f = self.flat
        check(lib().mtp_adamw_weight_images(self.table.data_ptr(), self.n, self.total_tiles, self.act, _f32(f.data), _f32(f.grad), _f32(m), _f32(v), _f32(hyper),
                                            _f32(sqn), max_norm, grad_scale, _s()), "mtp_adamw_weight_images")


def convt_pack(w, wg, wgT):
    Cin, Cout = w.shape[:2]
    dt = _dt(wg if wg is not None else wgT)
    check(lib().mtp_convt_pack(_f32(w), _p(wg), _p(wgT), dt, Cin, Cout, _s()), "mtp_convt_pack")


def convt_unpack_grad(dwg, dw):
    Cin, Cout = dw.shape[:2]
    check(lib().mtp_convt_unpack_grad(_f32(dwg), _f32(dw), Cin, Cout, _s()), "mtp_convt_unpack_grad")
    return dw


def tokens_to_nchw(x, out, B, Hp, Wp, levels):
    Cc = x.shape[-1]
    check(lib().mtp_tokens_to_nchw(_p(x), _dt(x), _p(out), _dt(out), B, Hp, Wp, Cc, levels, _s()), "mtp_tokens_to_nchw")
    return out


def nchw_to_tokens(f, out, B, Hp, Wp, levels):
    Cc = f.shape[1]
    check(lib().mtp_nchw_to_tokens(_p(f), _dt(f), _p(out), _dt(out), B, Hp, Wp, Cc, levels, _s()), "mtp_nchw_to_tokens")
    return out


def maxpool2_tokens_fwd(x, y, B, Hp, Wp):
    check(lib().mtp_maxpool2_tokens_fwd(_f32(x), _p(y), _dt(y), B, Hp, Wp, x.shape[-1], _s()), "mtp_maxpool2_tokens_fwd")
    return y


def maxpool2_tokens_bwd(x, dy, dx, B, Hp, Wp, accumulate=False):
    check(lib().mtp_maxpool2_tokens_bwd(_f32(x), _p(dy), _dt(dy), _f32(dx), int(accumulate), B, Hp, Wp, x.shape[-1], _s()), "mtp_maxpool2_tokens_bwd")
    return dx


def axpy(y, x, alpha=1.0):
    check(lib().mtp_axpy_f32(_f32(y), _f32(x), alpha, y.numel(), _s()), "mtp_axpy_f32")
    return y


def scale_rows_cast(src, dst, scale=None, rows_per_sample=0):
    rows, Cc = src.shape
    check(lib().mtp_scale_rows_cast(_f32(src), _p(dst), _dt(dst), _f32(scale), rows_per_sample, rows, Cc, _s()), "mtp_scale_rows_cast")
    return dst


# ------------------------------------------------------------------------------------------------ attention
def full_attn_fwd(qkv, o, lse, rel_h, rel_w, B, Hp, Wp, heads, scale):
    hd = qkv.shape[1] // (3 * heads)
    check(lib().mtp_full_attn_fwd(_p(qkv), _p(o), _f32(lse), _dt(qkv), _f32(rel_h), _f32(rel_w), B, Hp, Wp, heads, hd, scale, _s()), "mtp_full_attn_fwd")
    return o, lse


def full_attn_bwd(qkv, o, dout, lse, dqkv, rel_h, rel_w, drel_h, drel_w, B, Hp, Wp, heads, scale, accumulate=False, defer=None):
    hd = qkv.shape[1] // (3 * heads)
    rt = (2 * Hp - 1) + (2 * Wp - 1)
    part = torch.empty(B * heads, rt * hd, device=qkv.device, dtype=torch.float32)
    nws = lib().mtp_full_attn_bwd_workspace_floats(B, Hp, Wp, heads)     # > 0 beyond 256 tokens (three-pass backward)
    ws = torch.empty(nws, device=qkv.device, dtype=torch.float32) if nws else None
    check(lib().mtp_full_attn_bwd(_p(qkv), _p(o), _p(dout), _f32(lse), _p(dqkv), _dt(qkv), _f32(rel_h), _f32(rel_w), _p(part), _p(ws),
                                  B, Hp, Wp, heads, hd, scale, _s()), "mtp_full_attn_bwd")
    if defer is not None and _adjacent(drel_h, drel_w) and drel_h.numel() == (2 * Hp - 1) * hd:
        defer.append((part, drel_h, accumulate))       # reduced with the burst's other partial rows (reduce_rows_deferred)
        return dqkv
    _reduce_pair(part, (2 * Hp - 1) * hd, drel_h, drel_w, accumulate)   # per-(image, head) partials -> the two parameters
    return dqkv


def rvsa_windows(Hp, Wp):
    nh = (Hp + (7 - Hp % 7) % 7) // 7
    nw = (Wp + (7 - Wp % 7) % 7) // 7
    return nh, nw


def rvsa_pool_fwd(x, avg, pooled, B, Hp, Wp):
    check(lib().mtp_rvsa_pool_fwd(_p(x), _dt(x), _f32(avg), _f32(pooled), B, Hp, Wp, x.shape[-1], _s()), "mtp_rvsa_pool_fwd")


def rvsa_pool_bwd(dpooled, avg, dx, B, Hp, Wp, accumulate=True):
    check(lib().mtp_rvsa_pool_bwd(_f32(dpooled), _f32(avg), _p(dx), _dt(dx), int(accumulate), B, Hp, Wp, dx.shape[-1], _s()), "mtp_rvsa_pool_bwd")


def rvsa_sampling_fwd(x, w, b, avg, pooled, samp, B, Hp, Wp):
    """pool -> LeakyReLU -> stacked 1x1 heads in one launch (mtp_rvsa_sampling_fwd)"""
    check(lib().mtp_rvsa_sampling_fwd(_p(x), _dt(x), _f32(w), _f32(b), _f32(avg), _f32(pooled), _f32(samp), B, Hp, Wp, x.shape[-1], w.shape[0], _s()),
          "mtp_rvsa_sampling_fwd")
    return samp


def rvsa_sampling_bwd(dsamp, w, avg, dx, B, Hp, Wp):
    """dx += broadcast((dsamp . w) * leaky'(avg) / 49) in one launch (mtp_rvsa_sampling_bwd)"""
    check(lib().mtp_rvsa_sampling_bwd(_f32(dsamp), _f32(w), _f32(avg), _p(dx), _dt(dx), B, Hp, Wp, dx.shape[-1], w.shape[0], _s()), "mtp_rvsa_sampling_bwd")
    return dx


def rvsa_sampling_bwd_win(dsamp, w, avg, g):
    """g (windows, C) f32 = (dsamp . w) * leaky'(avg) / 49 -- the per-window addend layernorm_bwd(win_add=...) spreads over the tokens"""
    R, Cc = avg.shape
    check(lib().mtp_rvsa_sampling_bwd_win(_f32(dsamp), _f32(w), _f32(avg), _f32(g), R, Cc, w.shape[0], _s()), "mtp_rvsa_sampling_bwd_win")
    return g


def small_linear_fwd(x, w, b, y):
    R, K = x.shape
    check(lib().mtp_small_linear_fwd(_f32(x), _f32(w), _f32(b), _f32(y), R, w.shape[0], K, _s()), "mtp_small_linear_fwd")
    return y


def small_linear_bwd(x, w, dy, dx, dw, db):
    """dx may be None (weight / bias gradients only)"""
    R, K = x.shape
    check(lib().mtp_small_linear_bwd(_f32(x), _f32(w), _f32(dy), _f32(dx), _f32(dw), _f32(db), R, w.shape[0], K, _s()), "mtp_small_linear_bwd")


def small_linear_dw_segments(x, dy, dws, dbs):
    """dws[j] (rows_j, K) += dy[:, rows of segment j]^T x, dbs[j] += column sums: the stacked heads' gradients straight into their parameters"""
    R, K = x.shape
    n = len(dws)
    assert 0 < n <= 4 and len(dbs) == n and sum(d.shape[0] for d in dws) == dy.shape[1]
    rows = (C.c_int64 * n)(*[d.shape[0] for d in dws])
    P = C.c_void_p * n
    check(lib().mtp_small_linear_dw_segments(_f32(x), _f32(dy), R, dy.shape[1], K, n, rows, P(*[_f32(d) for d in dws]), P(*[_f32(d) for d in dbs]), _s()),
          "mtp_small_linear_dw_segments")


SL_BATCH = 8


def small_linear_dw_segments_flush(jobs):
    """jobs: (x, dy, dws, dbs) tuples queued instead of small_linear_dw_segments calls (the stacked sampling heads of a burst of RVSA blocks):
    equally-shaped ones go out together, <= SL_BATCH per launch.  Empties the list."""
    groups = {}
    for x, dy, dws, dbs in jobs:
        groups.setdefault((tuple(x.shape), tuple(dy.shape), tuple(d.shape[0] for d in dws)), []).append((x, dy, dws, dbs))
    for ((R, K), (_, N), rows_), g in groups.items():
        nseg = len(rows_)
        rows = (C.c_int64 * nseg)(*rows_)
        for i0 in range(0, len(g), SL_BATCH):
            chunk = g[i0:i0 + SL_BATCH]
            n = len(chunk)
            P, PS = C.c_void_p * n, C.c_void_p * (n * nseg)
            check(lib().mtp_small_linear_dw_segments_batched(P(*[_f32(c[0]) for c in chunk]), P(*[_f32(c[1]) for c in chunk]), n, R, N, K, nseg, rows,
                                                             PS(*[_f32(d) for c in chunk for d in c[2]]), PS(*[_f32(d) for c in chunk for d in c[3]]), _s()),
                  "mtp_small_linear_dw_segments_batched")
    del jobs[:]


def rvsa_attn_fwd(qkv, samp, o, lse, rel_h, rel_w, table, B, Hp, Wp, heads, scale):
    hd = qkv.shape[1] // (3 * heads)
    check(lib().mtp_rvsa_attn_fwd(_p(qkv), _f32(samp), _p(o), _f32(lse), _dt(qkv), _f32(rel_h), _f32(rel_w), _f32(table),
                                  B, Hp, Wp, heads, hd, scale, _s()), "mtp_rvsa_attn_fwd")
    return o, lse


def rvsa_attn_bwd(qkv, samp, o, dout, lse, dqkv, dsamp, rel_h, rel_w, table, drel_h, drel_w, dtable, B, Hp, Wp, heads, scale, accumulate=False, defer=None):
    hd = qkv.shape[1] // (3 * heads)
    T, C3 = qkv.shape
    Cc = C3 // 3
    nh, nw = rvsa_windows(Hp, Wp)
    nblk = B * nh * nw * heads
    dev = qkv.device
    dkv = torch.empty(T, 2 * Cc, device=dev, dtype=torch.float32)
    rel_part = torch.empty(nblk, 26 * hd, device=dev, dtype=torch.float32)
    tab_part = torch.empty(B * nh * nw, heads * 169, device=dev, dtype=torch.float32)     # (window, head, 169): contiguous per workgroup
    check(lib().mtp_rvsa_attn_bwd(_p(qkv), _f32(samp), _p(o), _p(dout), _f32(lse), _p(dqkv), _p(dkv), _f32(dsamp), _p(rel_part), _p(tab_part),
                                  _dt(qkv), _f32(rel_h), _f32(rel_w), _f32(table), B, Hp, Wp, heads, hd, scale, _s()), "mtp_rvsa_attn_bwd")
    if defer is not None and _adjacent(drel_h, drel_w) and drel_h.numel() == 13 * hd:
        # the partial rows wait for the burst's reduction launch (reduce_rows_deferred): one launch per burst of blocks instead of two per block
        defer.append((rel_part, drel_h, accumulate))
        defer.append((tab_part, dtable, accumulate, (heads, 169)))
        return dqkv
    _reduce_pair(rel_part, 13 * hd, drel_h, drel_w, accumulate)   # per-workgroup partials -> the parameters, no staging copies
    # per-(window, head) partials (heads, 169) -> the (169, heads) parameter gradient: one launch, transposed on the way
    check(lib().mtp_reduce_rows_t_f32(_p(tab_part), tab_part.shape[1], _f32(dtable), tab_part.shape[0], heads, 169, int(accumulate), _s()),
          "mtp_reduce_rows_t_f32")
    return dqkv


# ------------------------------------------------------------------------------------------------ optimizer
def zero_segments(base, start, count):
    """base[start[i] : start[i] + count[i]] = 0 for the int64 device tables start / count (one launch, one workgroup per entry)"""
    assert base.dtype == torch.float32 and start.dtype == torch.int64 and count.dtype == torch.int64 and start.numel() == count.numel()
    check(lib().mtp_zero_segments_f32(_p(base), start.data_ptr(), count.data_ptr(), start.numel(), _s()), "mtp_zero_segments_f32")


def sqnorm(g, out):
    check(lib().mtp_sqnorm_f32(_f32(g), _f32(out), g.numel(), _s()), "mtp_sqnorm_f32")
    return out


def sqnorm_segments(base, start, count, out):
    """out += sum(base[start[i] : start[i] + count[i]] ** 2) over the int64 device tables start / count (one launch)"""
    assert base.dtype == torch.float32 and start.dtype == torch.int64 and count.dtype == torch.int64 and start.numel() == count.numel()
    check(lib().mtp_sqnorm_segments_f32(_p(base), start.data_ptr(), count.data_ptr(), start.numel(), _f32(out), _s()), "mtp_sqnorm_segments_f32")
    return out


def adamw_flat(p, g, m, v, seg_start, seg_wd, hyper, sqn=None, max_norm=0.0, grad_scale=1.0):
    assert seg_start.dtype == torch.int64 and seg_start.is_cuda
    check(lib().mtp_adamw_flat(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), seg_start.data_ptr(), _f32(seg_wd), seg_start.numel(),
                               _f32(hyper), _f32(sqn), max_norm, grad_scale, _s()), "mtp_adamw_flat")


# ------------------------------------------------------------------------------------------------ InternImage layers (csrc/conv.hip)
def pad8(n):
    return (n + 7) // 8 * 8


def conv_out(h, stride):
    return (h - 1) // stride + 1


def im2col3x3(x, strides, cols, N, H, W, Cin, stride):
    """x: any tensor addressed by element strides (sN, sH, sW, sC); cols (N*Ho*Wo, Kp) ACT"""
    Kp = cols.shape[1]
    assert cols.shape[0] == N * conv_out(H, stride) * conv_out(W, stride) and Kp >= 9 * Cin
    sN, sH, sW, sC = strides
    if not x.is_cuda:
        raise RuntimeError("mtp_amd ops run only on an MI355X device tensor (no CPU fallback)")
    check(lib().mtp_im2col3x3(x.data_ptr(), _dt(x), sN, sH, sW, sC, _p(cols), _dt(cols), N, H, W, Cin, stride, Kp, _s()), "mtp_im2col3x3")
    return cols


def col2im3x3(dcols, dx, strides, N, H, W, Cin, stride, accumulate=False):
    sN, sH, sW, sC = strides
    assert dx.dtype == torch.float32 and dx.is_cuda
    check(lib().mtp_col2im3x3(_p(dcols), _dt(dcols), dx.data_ptr(), sN, sH, sW, sC, N, H, W, Cin, stride, dcols.shape[1], int(accumulate), _s()), "mtp_col2im3x3")
    return dx


def conv3x3_pack(w, w2, w2t):
    Cout, Cin = w.shape[:2]
    img = w2 if w2 is not None else w2t
    Kp = w2.shape[1] if w2 is not None else w2t.shape[0]
    check(lib().mtp_conv3x3_pack(_f32(w), _p(w2), _p(w2t), _dt(img), Cout, Cin, Kp, _s()), "mtp_conv3x3_pack")


def conv3x3_unpack_grad(dw2, dw):
    Cout, Cin = dw.shape[:2]
    check(lib().mtp_conv3x3_unpack_grad(_f32(dw2), _f32(dw), Cout, Cin, dw2.shape[1], _s()), "mtp_conv3x3_unpack_grad")
    return dw


def pack_rows_padded(w, wp, wpt):
    R, Cc = w.shape
    img = wp if wp is not None else wpt
    Rp = wp.shape[0] if wp is not None else wpt.shape[1]
    check(lib().mtp_pack_rows_padded(_f32(w), _p(wp), _p(wpt), _dt(img), R, Cc, Rp, _s()), "mtp_pack_rows_padded")


def dwconv3x3_fwd(x, w, b, y, N, H, W):
    check(lib().mtp_dwconv3x3_fwd(_p(x), _f32(w), _f32(b), _p(y), _dt(x), N, H, W, x.shape[-1], _s()), "mtp_dwconv3x3_fwd")
    return y


def dwconv3x3_bwd_dx(dy, w, dx, N, H, W, accumulate=False):
    check(lib().mtp_dwconv3x3_bwd_dx(_p(dy), _dt(dy), _f32(w), _f32(dx), int(accumulate), N, H, W, dy.shape[-1], _s()), "mtp_dwconv3x3_bwd_dx")
    return dx


def dwconv3x3_bwd_dw(dy, x, dw, db, N, H, W, accumulate=False):
    """dw (C,1,3,3) f32, db (C,) f32"""
    Cc = dy.shape[-1]
    nb = lib().mtp_dwconv3x3_bwd_dw_partial_rows(N, H, W)
    part = torch.empty(nb, 10 * Cc, device=dy.device, dtype=torch.float32)
    check(lib().mtp_dwconv3x3_bwd_dw(_p(dy), _p(x), _dt(dy), _p(part), N, H, W, Cc, _s()), "mtp_dwconv3x3_bwd_dw")
    _reduce_pair(part, 9 * Cc, dw, db, accumulate)


def dwconv_fwd(x, w, b, y, N, H, W, k):
    """depth-wise k x k (any odd k; InternImage-H/G's dw_kernel_size): w (C, 1, k, k) f32"""
    check(lib().mtp_dwconv_fwd(_p(x), _f32(w), _f32(b), _p(y), _dt(x), N, H, W, x.shape[-1], int(k), _s()), "mtp_dwconv_fwd")
    return y


def dwconv_bwd_dx(dy, w, dx, N, H, W, k, accumulate=False):
    check(lib().mtp_dwconv_bwd_dx(_p(dy), _dt(dy), _f32(w), _f32(dx), int(accumulate), N, H, W, dy.shape[-1], int(k), _s()), "mtp_dwconv_bwd_dx")
    return dx


def dwconv_bwd_dw(dy, x, dw, db, N, H, W, k):
    """dw (C, 1, k, k) / db (C,) f32 += the weight / bias gradient (f32 atomics: accumulates)"""
    check(lib().mtp_dwconv_bwd_dw(_p(dy), _p(x), _dt(dy), _f32(dw), _f32(db), N, H, W, dy.shape[-1], int(k), _s()), "mtp_dwconv_bwd_dw")


def center_feature_scale_fwd(y, xp, logits, out, G):
    rows, Cc = y.shape
    check(lib().mtp_center_feature_scale_fwd(_p(y), _p(xp), _p(logits), logits.shape[1], _p(out), _dt(y), rows, G, Cc // G, _s()), "mtp_center_feature_scale_fwd")
    return out


def center_feature_scale_bwd(dout, y, xp, logits, dy, dxp, dlogits, G):
    rows, Cc = y.shape
    assert dlogits.shape == logits.shape
    check(lib().mtp_center_feature_scale_bwd(_p(dout), _p(y), _p(xp), _p(logits), logits.shape[1], _p(dy), _f32(dxp), _p(dlogits), _dt(y), rows, G, Cc // G, _s()),
          "mtp_center_feature_scale_bwd")
    return dy


def softmax_groups_fwd(logits, prob, G, P):
    rows = logits.shape[0]
    check(lib().mtp_softmax_groups_fwd(_p(logits), logits.shape[1], _p(prob), _dt(logits), rows, G, P, _s()), "mtp_softmax_groups_fwd")
    return prob


def softmax_groups_bwd(prob, dprob, dlogits, G, P):
    rows = dlogits.shape[0]
    check(lib().mtp_softmax_groups_bwd(_p(prob), _f32(dprob), _p(dlogits), dlogits.shape[1], _dt(prob), rows, G, P, _s()), "mtp_softmax_groups_bwd")
    return dlogits


def scale_residual_fwd(x, z, gamma, out, out_act=None, sample_scale=None, rows_per_sample=0):
    rows, Cc = x.shape
    check(lib().mtp_scale_residual_fwd(_f32(x), _p(z), _dt(z), _f32(gamma), _f32(sample_scale), rows_per_sample, _f32(out), _p(out_act), rows, Cc, _s()),
          "mtp_scale_residual_fwd")
    return out


def scale_residual_bwd(dout, z, gamma, dz, dgamma, sample_scale=None, rows_per_sample=0, accumulate=False):
    rows, Cc = dout.shape
    nb = lib().mtp_scale_residual_bwd_partial_rows(rows)
    part = torch.empty(nb, Cc, device=dout.device, dtype=torch.float32)
    check(lib().mtp_scale_residual_bwd(_f32(dout), _p(z), _dt(z), _f32(gamma), _f32(sample_scale), rows_per_sample, _p(dz), _p(part), rows, Cc, _s()),
          "mtp_scale_residual_bwd")
    reduce_rows(part, dgamma, accumulate)
    return dz


def cast_pad_rows(src, dst):
    """src (rows, n) f32 -> dst (rows, ld >= n) ACT, columns n .. ld zero"""
    rows, n = src.shape
    assert dst.shape[0] == rows and dst.shape[1] >= n
    check(lib().mtp_cast_pad_rows(_f32(src), n, _p(dst), _dt(dst), dst.shape[1], rows, _s()), "mtp_cast_pad_rows")
    return dst


def copy_rows(src, dst, n):
    """dst[:, :n] = src[:, :n] (same dtype, 2-D contiguous buffers of different widths)"""
    assert src.dtype == dst.dtype and src.shape[0] == dst.shape[0]
    check(lib().mtp_copy_rows(_p(src), src.shape[1], _p(dst), dst.shape[1], _dt(src), n, src.shape[0], _s()), "mtp_copy_rows")
    return dst
