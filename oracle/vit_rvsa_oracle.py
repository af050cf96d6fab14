"""ORACLE (test infrastructure only) -- CPU restatement of the MTP ViT+RVSA backbone hot path.

Plain torch (CPU, fp32 or fp64), token-major, written from the algorithm -- not from the reference's
code shape.  Every function cites the reference lines it restates ("VIT" =
/root/reference/Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py).

PINNED: checked against golden vectors generated in the build container by importing the reference
itself (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py), forward and
gradients, to <=1e-5 (fp32).  The reference has no tests / known-answer vectors of its own for this
path, so the reference-generated fixtures are the pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (mtp_amd/) never imports it and has no CPU fallback.

Layout conventions shared with the HIP path:
  tokens      (T, C)   T = B*Hp*Wp, row t = (b, y, x) row-major
  qkv         (T, 3C)  channel = [q|k|v][head][64]                       (VIT:97, VIT:390)
  samp        (B*nh*nw, 5*heads) = [off(head,2) | scale(head,2) | angle(head)]  per window (VIT:354-368)
The *_bwd functions are the hand-derived backward formulas the HIP kernels implement; they are
checked against torch.autograd of the forward in tests/test_oracle_golden.py (the "hand-derived backward == autograd"
assertions), next to the comparison with the reference's own gradients.  `preprocess` (the MTP_DataPreprocessor image
path) is the one function that is NOT pinned: its arithmetic lives in mmengine, see its docstring.
"""
import math

import torch
import torch.nn.functional as F

from .index_ops import rvsa_geometry

LN_EPS = 1e-6           # VIT:596
LEAKY_SLOPE = 0.01      # nn.LeakyReLU() default, VIT:230
WS = 7                  # VIT:629


# ----------------------------------------------------------------------------- patch embed
def preprocess(img_u8, mean, std, bgr_to_rgb=True, pad_size_divisor=32, pad_value=0.0):
    """Image path of MTP_DataPreprocessor.forward (Multi-Task_Pretrain/preprocessing.py:145-148 -> `super().forward`), as
    configured at models.py:37-41 (mean [123.675, 116.28, 103.53], std [58.395, 57.12, 57.375], bgr_to_rgb, divisor 32).
    PARITY UNPINNED: the arithmetic lives in mmengine's ImgDataPreprocessor (mmengine is not vendored in the reference and
    not installed here), restated from its published algorithm: per image `x[[2, 1, 0]]` when bgr_to_rgb, `.float()`,
    `(x - mean) / std` with mean/std viewed (3,1,1), then `stack_batch`: pad bottom/right with pad_value up to a multiple
    of pad_size_divisor.  img_u8: (B, H, W, 3) uint8 HWC  ->  (B, 3, Hpad, Wpad) f32."""
    x = img_u8.permute(0, 3, 1, 2)
    if bgr_to_rgb:
        x = x[:, [2, 1, 0]]
    x = x.float()
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    x = (x - m) / s
    H, W = x.shape[-2:]
    Hpad, Wpad = -(-H // pad_size_divisor) * pad_size_divisor, -(-W // pad_size_divisor) * pad_size_divisor
    return torch.nn.functional.pad(x, (0, Wpad - W, 0, Hpad - H), value=pad_value)


def patchify(img, P=16):
    """VIT:529,536-539 -- (B,3,H,W) -> (B*Hp*Wp, 3*P*P), K order (c, ky, kx)."""
    B, Cin, H, W = img.shape
    Hp, Wp = H // P, W // P
    x = img[:, :, :Hp * P, :Wp * P].reshape(B, Cin, Hp, P, Wp, P)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(B * Hp * Wp, Cin * P * P), (Hp, Wp)


def unpatchify(cols, B, Cin, H, W, P=16):
    """adjoint of patchify (pixels outside Hp*P x Wp*P get 0)."""
    Hp, Wp = H // P, W // P
    x = cols.reshape(B, Hp, Wp, Cin, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, Cin, Hp * P, Wp * P)
    out = cols.new_zeros(B, Cin, H, W)
    out[:, :, :Hp * P, :Wp * P] = x
    return out


def patch_embed(img, w, b, pos_embed=None):
    """VIT:531-540 + VIT:793-794 -- conv as GEMM, + abs pos embed (1,N,C) broadcast over batch."""
    cols, (Hp, Wp) = patchify(img, w.shape[-1])
    x = cols @ w.reshape(w.shape[0], -1).t() + b
    if pos_embed is not None:
        x = (x.reshape(img.shape[0], Hp * Wp, -1) + pos_embed).reshape(-1, w.shape[0])
    return x, (Hp, Wp)


# ----------------------------------------------------------------------------- layernorm / gelu
def layernorm_fwd(x, g, b, eps=LN_EPS):
    """nn.LayerNorm(C, eps=1e-6) (VIT:484,496,579,596): biased variance, stats over last dim."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    return (x - mean) * rstd * g + b, mean.squeeze(-1), rstd.squeeze(-1)


def layernorm_bwd(dy, x, mean, rstd, g):
    xhat = (x - mean[:, None]) * rstd[:, None]
    wdy = dy * g
    c1 = (wdy * xhat).mean(-1, keepdim=True)
    c2 = wdy.mean(-1, keepdim=True)
    dx = (wdy - xhat * c1 - c2) * rstd[:, None]
    return dx, (dy * xhat).sum(0), dy.sum(0)


def gelu(x):
    """nn.GELU() exact erf form (VIT:51, VIT:644)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def dgelu(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def mlp(x, w1, b1, w2, b2):
    """VIT:55-62."""
    return gelu(x @ w1.t() + b1) @ w2.t() + b2


# ----------------------------------------------------------------------------- full attention (VIT:90-111, 142-193)
def _rel_index(n_q, n_k, device):
    i = torch.arange(n_q, device=device)[:, None] - torch.arange(n_k, device=device)[None, :] + (n_k - 1)
    return i  # VIT:160-172 with q_shape == k_shape


def full_attn_fwd(qkv, B, Hp, Wp, heads, rel_h, rel_w, scale=None):
    """qkv (T,3C) -> o (T,C), lse (B,heads,N).  logits = (s q).k + (s q).Rh[hq-hk+Hp-1] + (s q).Rw[wq-wk+Wp-1]
    (the reference scales q first, VIT:100, and passes the scaled q to calc_rel_pos_spatial, VIT:103)."""
    T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    N = Hp * Wp
    scale = scale if scale is not None else hd ** -0.5
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)          # (B,heads,N,hd)
    qs = q * scale
    logits = qs @ k.transpose(-1, -2)                                          # (B,heads,N,N)
    Rh = rel_h[_rel_index(Hp, Hp, qkv.device)]                                 # (Hp,Hp,hd)
    Rw = rel_w[_rel_index(Wp, Wp, qkv.device)]
    q5 = qs.reshape(B, heads, Hp, Wp, hd)
    relh = torch.einsum("byhwc,hkc->byhwk", q5, Rh)                            # (B,heads,Hp,Wp,Hp)
    relw = torch.einsum("byhwc,wkc->byhwk", q5, Rw)
    logits = (logits.reshape(B, heads, Hp, Wp, Hp, Wp) + relh[..., :, None] + relw[..., None, :]).reshape(B, heads, N, N)
    lse = torch.logsumexp(logits, dim=-1)
    p = torch.exp(logits - lse[..., None])
    o = (p @ v).transpose(1, 2).reshape(T, C)
    return o, lse


def full_attn_bwd(do, qkv, o, lse, B, Hp, Wp, heads, rel_h, rel_w, scale=None):
    """Hand-derived backward: returns dqkv (T,3C), drel_h, drel_w."""
    T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    N = Hp * Wp
    scale = scale if scale is not None else hd ** -0.5
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    qs = q * scale
    ih, iw = _rel_index(Hp, Hp, qkv.device), _rel_index(Wp, Wp, qkv.device)
    Rh, Rw = rel_h[ih], rel_w[iw]
    q5 = qs.reshape(B, heads, Hp, Wp, hd)
    logits = (qs @ k.transpose(-1, -2)).reshape(B, heads, Hp, Wp, Hp, Wp) \
        + torch.einsum("byhwc,hkc->byhwk", q5, Rh)[..., :, None] + torch.einsum("byhwc,wkc->byhwk", q5, Rw)[..., None, :]
    p = torch.exp(logits.reshape(B, heads, N, N) - lse[..., None])
    dO = do.reshape(B, N, heads, hd).transpose(1, 2)
    O = o.reshape(B, N, heads, hd).transpose(1, 2)
    dv = p.transpose(-1, -2) @ dO
    dp = dO @ v.transpose(-1, -2)
    delta = (dO * O).sum(-1, keepdim=True)
    ds = p * (dp - delta)                                                      # (B,heads,N,N)
    ds6 = ds.reshape(B, heads, Hp, Wp, Hp, Wp)
    d_relh = ds6.sum(-1)                                                       # (B,heads,Hp,Wp,kh)
    d_relw = ds6.sum(-2)                                                       # (B,heads,Hp,Wp,kw)
    dqs = ds @ k + (torch.einsum("byhwk,hkc->byhwc", d_relh, Rh) + torch.einsum("byhwk,wkc->byhwc", d_relw, Rw)).reshape(B, heads, N, hd)
    dk = ds.transpose(-1, -2) @ qs
    dRh = torch.einsum("byhwk,byhwc->hkc", d_relh, q5)                         # (Hp,Hp,hd) grad of gathered table
    dRw = torch.einsum("byhwk,byhwc->wkc", d_relw, q5)
    drel_h = torch.zeros_like(rel_h).index_add_(0, ih.reshape(-1), dRh.reshape(-1, hd))
    drel_w = torch.zeros_like(rel_w).index_add_(0, iw.reshape(-1), dRw.reshape(-1, hd))
    dqkv = torch.stack([dqs * scale, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(T, C3)
    return dqkv, drel_h, drel_w


# ----------------------------------------------------------------------------- RVSA (VIT:195-433)
def rvsa_pool_fwd(x, B, Hp, Wp):
    """VIT:347,354 first two stages of each sampling head: zero-pad to (He,We), AvgPool2d(7,7)
    (always /49), LeakyReLU(0.01).  x (T,C) -> avg (B*nh*nw, C), pooled = leaky(avg)."""
    g = rvsa_geometry(Hp, Wp, WS)
    C = x.shape[1]
    xp = F.pad(x.reshape(B, Hp, Wp, C), (0, 0, g["pad_left"], g["pad_right"], g["pad_top"], g["pad_down"]))
    avg = xp.reshape(B, g["nh"], WS, g["nw"], WS, C).sum(dim=(2, 4)) / float(WS * WS)
    avg = avg.reshape(B * g["nh"] * g["nw"], C)
    return avg, torch.where(avg > 0, avg, avg * LEAKY_SLOPE)


def rvsa_pool_bwd(dpooled, avg, B, Hp, Wp):
    """adjoint: each in-image token of a window receives dpooled * leaky'(avg) / 49."""
    g = rvsa_geometry(Hp, Wp, WS)
    C = avg.shape[1]
    davg = dpooled * torch.where(avg > 0, torch.ones_like(avg), torch.full_like(avg, LEAKY_SLOPE)) / float(WS * WS)
    d = davg.reshape(B, g["nh"], 1, g["nw"], 1, C).expand(B, g["nh"], WS, g["nw"], WS, C).reshape(B, g["He"], g["We"], C)
    return d[:, g["pad_top"]:g["pad_top"] + Hp, g["pad_left"]:g["pad_left"] + Wp].reshape(B * Hp * Wp, C)


def sampling_weight(off_w, off_b, sc_w, sc_b, an_w, an_b):
    """Stack the three 1x1 conv heads (VIT:231,236,242) into one (5*heads, C) matrix / bias."""
    W = torch.cat([off_w.reshape(off_w.shape[0], -1), sc_w.reshape(sc_w.shape[0], -1), an_w.reshape(an_w.shape[0], -1)], 0)
    return W, torch.cat([off_b, sc_b, an_b], 0)


def rvsa_sample_coords(samp, B, Hp, Wp, heads):
    """Closed form of VIT:312-388 + grid_sample's align_corners=True unnormalisation (VIT:397-404).
    samp (B*nh*nw, 5*heads) -> pixel coords ix, iy of shape (B, heads, nh, nw, 7, 7) in the padded (He,We) map."""
    g = rvsa_geometry(Hp, Wp, WS)
    He, We, nh, nw = g["He"], g["We"], g["nh"], g["nw"]
    dt, dev = samp.dtype, samp.device
    s = samp.reshape(B, nh, nw, 5 * heads)
    off = s[..., :2 * heads].reshape(B, nh, nw, heads, 2)
    sc = s[..., 2 * heads:4 * heads].reshape(B, nh, nw, heads, 2)
    ang = s[..., 4 * heads:]
    off_x = off[..., 0] / g["div_x"]            # VIT:359  (x offset / (h // ws))
    off_y = off[..., 1] / g["div_y"]            # VIT:360
    lin_x = torch.linspace(-1, 1, We, dtype=dt, device=dev)
    lin_y = torch.linspace(-1, 1, He, dtype=dt, device=dev)
    cen_x = lin_x.reshape(nw, WS).mean(-1)       # VIT:317
    cen_y = lin_y.reshape(nh, WS).mean(-1)
    t = torch.arange(WS, dtype=dt, device=dev)
    rel_x = t * 2 / (We - 1)
    rel_x = rel_x - rel_x.mean()                 # VIT:328-329
    rel_y = t * 2 / (He - 1)
    rel_y = rel_y - rel_y.mean()
    # broadcast to (B, nh, nw, heads, a, b)
    sx = (sc[..., 0] + 1)[..., None, None]
    sy = (sc[..., 1] + 1)[..., None, None]
    rx = rel_x[None, None, None, None, None, :] * sx        # VIT:372
    ry = rel_y[None, None, None, None, :, None] * sy
    cs, sn = torch.cos(ang)[..., None, None], torch.sin(ang)[..., None, None]
    gx = cen_x[None, None, :, None, None, None] + (rx * cs - ry * sn) + off_x[..., None, None]   # VIT:380,385
    gy = cen_y[None, :, None, None, None, None] + (ry * cs + rx * sn) + off_y[..., None, None]   # VIT:381,385
    ix = (gx + 1) * 0.5 * (We - 1)
    iy = (gy + 1) * 0.5 * (He - 1)
    return ix.permute(0, 3, 1, 2, 4, 5), iy.permute(0, 3, 1, 2, 4, 5)   # (B,heads,nh,nw,7,7)


def _padded_heads(t, B, Hp, Wp, heads, g):
    """(T, C) -> zero-padded (B, heads, He, We, hd)   (VIT:392: pad AFTER the projection)."""
    hd = t.shape[1] // heads
    m = t.reshape(B, Hp, Wp, heads, hd).permute(0, 3, 1, 2, 4)
    return F.pad(m, (0, 0, g["pad_left"], g["pad_right"], g["pad_top"], g["pad_down"]))


def _bilinear_gather(m, ix, iy):
    """F.grid_sample(bilinear, zeros, align_corners=True) on m (B,heads,He,We,hd) at pixel coords
    (B,heads,nh,nw,7,7) -> (B,heads,nh,nw,7,7,hd).  Out-of-range neighbours contribute 0."""
    B, H, He, We, hd = m.shape
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    out = 0
    flat = m.reshape(B, H, He * We, hd)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi <= We - 1) & (yi >= 0) & (yi <= He - 1)
            idx = (yi.clamp(0, He - 1) * We + xi.clamp(0, We - 1)).long()
            val = torch.gather(flat, 2, idx.reshape(B, H, -1, 1).expand(-1, -1, -1, hd)).reshape(*ix.shape, hd)
            out = out + val * (wx * wy * ok.to(m.dtype))[..., None]
    return out


def rvsa_attn_fwd(qkv, samp, B, Hp, Wp, heads, rel_h, rel_w, bias_table, scale=None):
    """VIT:390-428 -- windowed attention with K/V re-sampled at (shifted, scaled, rotated) positions.
    logits = scale*(q.k_sel) + q.Rh[a_q-a_k+6] + q.Rw[b_q-b_k+6] + table[idx, head]   (UNSCALED q in
    the rel-pos terms, VIT:410-412).  Returns o (T,C) and lse (B,heads,nh,nw,49)."""
    T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    scale = scale if scale is not None else hd ** -0.5
    g = rvsa_geometry(Hp, Wp, WS)
    nh, nw = g["nh"], g["nw"]
    qm = _padded_heads(qkv[:, :C], B, Hp, Wp, heads, g)
    km = _padded_heads(qkv[:, C:2 * C], B, Hp, Wp, heads, g)
    vm = _padded_heads(qkv[:, 2 * C:], B, Hp, Wp, heads, g)
    ix, iy = rvsa_sample_coords(samp, B, Hp, Wp, heads)
    ks = _bilinear_gather(km, ix, iy).reshape(B, heads, nh, nw, WS * WS, hd)
    vs = _bilinear_gather(vm, ix, iy).reshape(B, heads, nh, nw, WS * WS, hd)
    qw = qm.reshape(B, heads, nh, WS, nw, WS, hd).permute(0, 1, 2, 4, 3, 5, 6)            # (B,heads,nh,nw,7,7,hd)
    i7 = _rel_index(WS, WS, qkv.device)
    relh = torch.einsum("bhijxyc,xkc->bhijxyk", qw, rel_h[i7])                              # k = key row a_k
    relw = torch.einsum("bhijxyc,ykc->bhijxyk", qw, rel_w[i7])                              # k = key col b_k
    qf = qw.reshape(B, heads, nh, nw, WS * WS, hd)
    logits = (qf @ ks.transpose(-1, -2)) * scale
    logits = (logits.reshape(B, heads, nh, nw, WS, WS, WS, WS) + relh[..., :, None] + relw[..., None, :]).reshape(B, heads, nh, nw, 49, 49)
    a = torch.arange(WS, device=qkv.device)
    an, bn = a.repeat_interleave(WS), a.repeat(WS)
    idx = (an[:, None] - an[None, :] + WS - 1) * (2 * WS - 1) + (bn[:, None] - bn[None, :] + WS - 1)
    logits = logits + bias_table[idx].permute(2, 0, 1)[None, :, None, None]                 # VIT:414-418
    lse = torch.logsumexp(logits, -1)
    p = torch.exp(logits - lse[..., None])
    o = p @ vs                                                                              # (B,heads,nh,nw,49,hd)
    o = o.reshape(B, heads, nh, nw, WS, WS, hd).permute(0, 2, 4, 3, 5, 1, 6).reshape(B, g["He"], g["We"], C)
    o = o[:, g["pad_top"]:g["pad_top"] + Hp, g["pad_left"]:g["pad_left"] + Wp].reshape(T, C)   # VIT:426
    return o, lse


def rvsa_attn_bwd(do, qkv, samp, o, lse, B, Hp, Wp, heads, rel_h, rel_w, bias_table, scale=None):
    """Hand-derived backward of rvsa_attn_fwd: returns dqkv, dsamp, drel_h, drel_w, dbias_table."""
    T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    scale = scale if scale is not None else hd ** -0.5
    g = rvsa_geometry(Hp, Wp, WS)
    nh, nw, He, We = g["nh"], g["nw"], g["He"], g["We"]
    dev, dt = qkv.device, qkv.dtype
    qm = _padded_heads(qkv[:, :C], B, Hp, Wp, heads, g)
    km = _padded_heads(qkv[:, C:2 * C], B, Hp, Wp, heads, g)
    vm = _padded_heads(qkv[:, 2 * C:], B, Hp, Wp, heads, g)
    ix, iy = rvsa_sample_coords(samp, B, Hp, Wp, heads)
    ks = _bilinear_gather(km, ix, iy).reshape(B, heads, nh, nw, 49, hd)
    vs = _bilinear_gather(vm, ix, iy).reshape(B, heads, nh, nw, 49, hd)
    qw = qm.reshape(B, heads, nh, WS, nw, WS, hd).permute(0, 1, 2, 4, 3, 5, 6)
    qf = qw.reshape(B, heads, nh, nw, 49, hd)
    i7 = _rel_index(WS, WS, dev)
    Rh, Rw = rel_h[i7], rel_w[i7]
    relh = torch.einsum("bhijxyc,xkc->bhijxyk", qw, Rh)
    relw = torch.einsum("bhijxyc,ykc->bhijxyk", qw, Rw)
    a = torch.arange(WS, device=dev)
    an, bn = a.repeat_interleave(WS), a.repeat(WS)
    idx = (an[:, None] - an[None, :] + WS - 1) * (2 * WS - 1) + (bn[:, None] - bn[None, :] + WS - 1)
    logits = ((qf @ ks.transpose(-1, -2)) * scale).reshape(B, heads, nh, nw, WS, WS, WS, WS) + relh[..., :, None] + relw[..., None, :]
    logits = logits.reshape(B, heads, nh, nw, 49, 49) + bias_table[idx].permute(2, 0, 1)[None, :, None, None]
    p = torch.exp(logits - lse[..., None])

    def to_win(t):   # (T,C) -> padded windows (B,heads,nh,nw,49,hd), zeros in the padding
        return _padded_heads(t, B, Hp, Wp, heads, g).reshape(B, heads, nh, WS, nw, WS, hd).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, heads, nh, nw, 49, hd)

    dO, O = to_win(do), to_win(o)
    dvs = p.transpose(-1, -2) @ dO
    dp = dO @ vs.transpose(-1, -2)
    delta = (dO * O).sum(-1, keepdim=True)
    ds = p * (dp - delta)                                                     # (B,heads,nh,nw,49,49)
    ds8 = ds.reshape(B, heads, nh, nw, WS, WS, WS, WS)
    d_relh, d_relw = ds8.sum(-1), ds8.sum(-2)
    dq = (ds @ ks) * scale + (torch.einsum("bhijxyk,xkc->bhijxyc", d_relh, Rh) + torch.einsum("bhijxyk,ykc->bhijxyc", d_relw, Rw)).reshape(B, heads, nh, nw, 49, hd)
    dks = (ds.transpose(-1, -2) @ qf) * scale
    drel_h = torch.zeros_like(rel_h).index_add_(0, i7.reshape(-1), torch.einsum("bhijxyk,bhijxyc->xkc", d_relh, qw).reshape(-1, hd))
    drel_w = torch.zeros_like(rel_w).index_add_(0, i7.reshape(-1), torch.einsum("bhijxyk,bhijxyc->ykc", d_relw, qw).reshape(-1, hd))
    dtab = torch.zeros_like(bias_table).index_add_(0, idx.reshape(-1), ds.sum(dim=(0, 2, 3)).permute(1, 2, 0).reshape(49 * 49, heads))

    # ---- scatter dks/dvs through the bilinear weights; coordinate gradients
    dks7 = dks.reshape(B, heads, nh, nw, WS, WS, hd)
    dvs7 = dvs.reshape(B, heads, nh, nw, WS, WS, hd)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    dkm = torch.zeros(B, heads, He * We, hd, dtype=dt, device=dev)
    dvm = torch.zeros_like(dkm)
    kflat, vflat = km.reshape(B, heads, He * We, hd), vm.reshape(B, heads, He * We, hd)
    dix = torch.zeros_like(ix)
    diy = torch.zeros_like(iy)
    for dy_, wy, dwy in ((0, 1 - fy, -1.0), (1, fy, 1.0)):
        for dx_, wx, dwx in ((0, 1 - fx, -1.0), (1, fx, 1.0)):
            xi, yi = x0 + dx_, y0 + dy_
            ok = ((xi >= 0) & (xi <= We - 1) & (yi >= 0) & (yi <= He - 1)).to(dt)
            lin = (yi.clamp(0, He - 1) * We + xi.clamp(0, We - 1)).long().reshape(B, heads, -1, 1).expand(-1, -1, -1, hd)
            wgt = (wx * wy * ok)[..., None]
            dkm.scatter_add_(2, lin, (dks7 * wgt).reshape(B, heads, -1, hd))
            dvm.scatter_add_(2, lin, (dvs7 * wgt).reshape(B, heads, -1, hd))
            kval = torch.gather(kflat, 2, lin).reshape(*ix.shape, hd)
            vval = torch.gather(vflat, 2, lin).reshape(*ix.shape, hd)
            dot = ((dks7 * kval).sum(-1) + (dvs7 * vval).sum(-1)) * ok      # d/d(weight)
            dix = dix + dot * wy * dwx
            diy = diy + dot * wx * dwy
    # pixel coords -> normalised coords (align_corners=True)
    dgx = dix * 0.5 * (We - 1)
    dgy = diy * 0.5 * (He - 1)
    # coords -> (off, scale, angle): recompute the pieces of rvsa_sample_coords
    s = samp.reshape(B, nh, nw, 5 * heads)
    sc = s[..., 2 * heads:4 * heads].reshape(B, nh, nw, heads, 2).permute(0, 3, 1, 2, 4)      # (B,heads,nh,nw,2)
    ang = s[..., 4 * heads:].permute(0, 3, 1, 2)                                              # (B,heads,nh,nw)
    t7 = torch.arange(WS, dtype=dt, device=dev)
    rel_x = t7 * 2 / (We - 1)
    rel_x = rel_x - rel_x.mean()
    rel_y = t7 * 2 / (He - 1)
    rel_y = rel_y - rel_y.mean()
    bx = rel_x[None, None, None, None, None, :]
    by = rel_y[None, None, None, None, :, None]
    rx = bx * (sc[..., 0] + 1)[..., None, None]
    ry = by * (sc[..., 1] + 1)[..., None, None]
    cs, sn = torch.cos(ang)[..., None, None], torch.sin(ang)[..., None, None]
    d_offx = dgx.sum(dim=(-1, -2)) / g["div_x"]
    d_offy = dgy.sum(dim=(-1, -2)) / g["div_y"]
    d_rx = dgx * cs + dgy * sn
    d_ry = -dgx * sn + dgy * cs
    d_sx = (d_rx * bx).sum(dim=(-1, -2))
    d_sy = (d_ry * by).sum(dim=(-1, -2))
    d_ang = (dgx * (-rx * sn - ry * cs) + dgy * (-ry * sn + rx * cs)).sum(dim=(-1, -2))
    dsamp = torch.cat([torch.stack([d_offx, d_offy], -1).permute(0, 2, 3, 1, 4).reshape(B, nh, nw, 2 * heads),
                       torch.stack([d_sx, d_sy], -1).permute(0, 2, 3, 1, 4).reshape(B, nh, nw, 2 * heads),
                       d_ang.permute(0, 2, 3, 1)], -1).reshape(B * nh * nw, 5 * heads)

    def from_map(mp):  # padded (B,heads,He,We,hd) -> (T,C) cropping the padding
        return mp[:, :, g["pad_top"]:g["pad_top"] + Hp, g["pad_left"]:g["pad_left"] + Wp].permute(0, 2, 3, 1, 4).reshape(T, C)

    dq_map = dq.reshape(B, heads, nh, nw, WS, WS, hd).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, heads, He, We, hd)
    dqkv = torch.cat([from_map(dq_map), from_map(dkm.reshape(B, heads, He, We, hd)), from_map(dvm.reshape(B, heads, He, We, hd))], 1)
    return dqkv, dsamp, drel_h, drel_w, dtab


# ----------------------------------------------------------------------------- FPN tail (VIT:640-654, 807-811)
def convT_gemm_weight(w):
    """ConvTranspose2d(C,C,2,2) weight (Cin,Cout,2,2) -> GEMM weight (4*Cout, Cin), row = (ky*2+kx)*Cout + co."""
    Cin, Cout = w.shape[:2]
    return w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin)


def convT_tokens(x, w, b):
    """ConvTranspose2d(k=2,s=2) on token-major rows: (R, Cin) -> (4R, Cout), row = r*4 + ky*2+kx."""
    Cout = w.shape[1]
    y = x @ convT_gemm_weight(w).t() + b.repeat(4)
    return y.reshape(-1, Cout)


def tokens_to_nchw(x, B, Hp, Wp, levels):
    """rows (b, py, px, q_1..q_L) x C -> (B, C, Hp*2^L, Wp*2^L); q_l = ky_l*2+kx_l (VIT:807 for L=0)."""
    C = x.shape[1]
    t = x.reshape(B, Hp, Wp, *([2, 2] * levels), C)
    # dims: 0 b, 1 py, 2 px, then (ky_l, kx_l) pairs, last C
    ydims = [1] + [3 + 2 * l for l in range(levels)]
    xdims = [2] + [4 + 2 * l for l in range(levels)]
    t = t.permute(0, 3 + 2 * levels, *ydims, *xdims)
    return t.reshape(B, C, Hp * 2 ** levels, Wp * 2 ** levels)


def nchw_to_tokens(f, B, Hp, Wp, levels):
    """inverse permutation of tokens_to_nchw."""
    C = f.shape[1]
    t = f.reshape(B, C, Hp, *([2] * levels), Wp, *([2] * levels))
    # dims: 0 b, 1 C, 2 py, 3..2+L ky_l, 3+L px, 4+L.. kx_l
    order = [0, 2, 3 + levels]
    for l in range(levels):
        order += [3 + l, 4 + levels + l]
    order.append(1)
    return t.permute(*order).reshape(B * Hp * Wp * 4 ** levels, C)


def fpn(taps, B, Hp, Wp, p):
    """VIT:807-811.  taps: 4 token-major (T,C) tensors; p: dict of fpn params with reference key names.
    The patch_size == 8 tail (VIT:656-670: one ConvT, identity, MaxPool 2, MaxPool 4) is recognised by the absence of fpn1's second ConvT."""
    if "fpn1.3.weight" not in p:
        f1 = tokens_to_nchw(convT_tokens(taps[0], p["fpn1.0.weight"], p["fpn1.0.bias"]), B, Hp, Wp, 1)
        f2 = tokens_to_nchw(taps[1], B, Hp, Wp, 0)
        f3 = F.max_pool2d(tokens_to_nchw(taps[2], B, Hp, Wp, 0), 2, 2)
        f4 = F.max_pool2d(tokens_to_nchw(taps[3], B, Hp, Wp, 0), 4, 4)
        return [f1, f2, f3, f4]
    y = convT_tokens(taps[0], p["fpn1.0.weight"], p["fpn1.0.bias"])
    y = gelu(layernorm_fwd(y, p["fpn1.1.ln.weight"], p["fpn1.1.ln.bias"])[0])
    y = convT_tokens(y, p["fpn1.3.weight"], p["fpn1.3.bias"])
    f1 = tokens_to_nchw(y, B, Hp, Wp, 2)
    f2 = tokens_to_nchw(convT_tokens(taps[1], p["fpn2.0.weight"], p["fpn2.0.bias"]), B, Hp, Wp, 1)
    f3 = tokens_to_nchw(taps[2], B, Hp, Wp, 0)
    f4 = F.max_pool2d(tokens_to_nchw(taps[3], B, Hp, Wp, 0), 2, 2)
    return [f1, f2, f3, f4]


# ----------------------------------------------------------------------------- blocks and the whole backbone
def drop_path_scale(B, rate, training, generator=None, dtype=torch.float32):
    """VIT:31-42 (timm drop_path): per-sample factor floor(keep + U[0,1)) / keep; 1 when inactive."""
    if rate == 0.0 or not training:
        return None
    keep = 1.0 - rate
    return torch.floor(keep + torch.rand(B, generator=generator, dtype=dtype)) / keep


def block_forward(x, p, pre, window, B, Hp, Wp, heads, dp_scale=None):
    """VIT:506-513 (init_values=None branch).  p: state-dict style mapping, pre = 'blocks.{i}.'."""
    C = x.shape[1]
    N = Hp * Wp
    h1 = layernorm_fwd(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])[0]
    qkv = h1 @ p[pre + "attn.qkv.weight"].t() + p[pre + "attn.qkv.bias"]
    if window:
        _, pooled = rvsa_pool_fwd(h1, B, Hp, Wp)
        Ws, bs = sampling_weight(p[pre + "attn.sampling_offsets.2.weight"], p[pre + "attn.sampling_offsets.2.bias"],
                                 p[pre + "attn.sampling_scales.2.weight"], p[pre + "attn.sampling_scales.2.bias"],
                                 p[pre + "attn.sampling_angles.2.weight"], p[pre + "attn.sampling_angles.2.bias"])
        samp = pooled @ Ws.t() + bs
        a, _ = rvsa_attn_fwd(qkv, samp, B, Hp, Wp, heads, p[pre + "attn.rel_pos_h"], p[pre + "attn.rel_pos_w"],
                             p[pre + "attn.relative_position_bias_table"])
    else:
        a, _ = full_attn_fwd(qkv, B, Hp, Wp, heads, *_full_rel_tables(p, pre, Hp, Wp, x.shape[1] // heads))
    a = a @ p[pre + "attn.proj.weight"].t() + p[pre + "attn.proj.bias"]
    if pre + "gamma_1" in p:          # layer scale (init_values is not None, VIT:500-504, 510-512): per-channel factor on each residual branch
        a = a * p[pre + "gamma_1"]
    if dp_scale is not None:
        a = a * dp_scale.repeat_interleave(N)[:, None]
    x = x + a
    h2 = layernorm_fwd(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])[0]
    m = mlp(h2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    if pre + "gamma_2" in p:
        m = m * p[pre + "gamma_2"]
    if dp_scale is not None:
        m = m * dp_scale.repeat_interleave(N)[:, None]
    return x + m


def _full_rel_tables(p, pre, Hp, Wp, hd):
    """decomposed rel-pos tables of a full-attention block; the ViTDet-style fine-tune copies (mmdet / mmrotate
    vit_rvsa_mtp.py:73-74, 93: parameters and the calc_rel_pos_spatial call commented out) have none -> zero tables."""
    h, w = p.get(pre + "attn.full_attn_rel_pos_h"), p.get(pre + "attn.full_attn_rel_pos_w")
    ref = p[pre + "attn.qkv.weight"]
    if h is None:
        h = torch.zeros(2 * Hp - 1, hd, dtype=ref.dtype)
    if w is None:
        w = torch.zeros(2 * Wp - 1, hd, dtype=ref.dtype)
    return h, w


def backbone_forward(img, p, depth, heads, interval, out_indices, dp_scales=None, vitdet=False, taps_only=False):
    """VIT:787-813 forward_features.  p: reference state-dict (name -> tensor).
    dp_scales: optional list over blocks of (attn_scale, mlp_scale) per-sample factors.
    vitdet=True: forward_features of the mmdet / mmrotate `RVSA_MTP` copies
    (RS_Tasks_Finetune/Horizontal_Detection/mmdet/models/backbones/vit_rvsa_mtp.py:822-844): no taps; the LAST block's
    output goes through the final `norm` (:835) and all four fpn ops are applied to that one map (:841)."""
    B = img.shape[0]
    x, (Hp, Wp) = patch_embed(img, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], p.get("pos_embed"))
    taps = []
    for i in range(depth):
        window = (i + 1) % interval != 0
        pre = "blocks.%d." % i
        if dp_scales is not None and dp_scales[i] is not None:
            # two independent draws per block (attn branch, mlp branch), VIT:508-509
            x = _block_two_scales(x, p, pre, window, B, Hp, Wp, heads, dp_scales[i])
        else:
            x = block_forward(x, p, pre, window, B, Hp, Wp, heads)
        if i in out_indices:
            taps.append(x)
    if vitdet:
        xn = layernorm_fwd(x, p["norm.weight"], p["norm.bias"])[0]
        taps = [xn, xn, xn, xn]
    if taps_only:   # mmpretrain / opencd copies: the taps as NCHW maps, fpn ops commented out (vit_rvsa_mtp.py:836-842)
        return [tokens_to_nchw(t, B, Hp, Wp, 0) for t in taps]
    return fpn(taps, B, Hp, Wp, p)


def _block_two_scales(x, p, pre, window, B, Hp, Wp, heads, scales):
    N = Hp * Wp
    sa, sm = scales
    y = block_forward_parts(x, p, pre, window, B, Hp, Wp, heads)
    a, fn_mlp = y
    x = x + a * sa.repeat_interleave(N)[:, None]
    return x + fn_mlp(x) * sm.repeat_interleave(N)[:, None]


def block_forward_parts(x, p, pre, window, B, Hp, Wp, heads):
    """attention branch output and a closure for the mlp branch (used for drop-path with two draws)."""
    h1 = layernorm_fwd(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])[0]
    qkv = h1 @ p[pre + "attn.qkv.weight"].t() + p[pre + "attn.qkv.bias"]
    if window:
        _, pooled = rvsa_pool_fwd(h1, B, Hp, Wp)
        Ws, bs = sampling_weight(p[pre + "attn.sampling_offsets.2.weight"], p[pre + "attn.sampling_offsets.2.bias"],
                                 p[pre + "attn.sampling_scales.2.weight"], p[pre + "attn.sampling_scales.2.bias"],
                                 p[pre + "attn.sampling_angles.2.weight"], p[pre + "attn.sampling_angles.2.bias"])
        samp = pooled @ Ws.t() + bs
        a, _ = rvsa_attn_fwd(qkv, samp, B, Hp, Wp, heads, p[pre + "attn.rel_pos_h"], p[pre + "attn.rel_pos_w"],
                             p[pre + "attn.relative_position_bias_table"])
    else:
        a, _ = full_attn_fwd(qkv, B, Hp, Wp, heads, *_full_rel_tables(p, pre, Hp, Wp, x.shape[1] // heads))
    a = a @ p[pre + "attn.proj.weight"].t() + p[pre + "attn.proj.bias"]
    if pre + "gamma_1" in p:
        a = a * p[pre + "gamma_1"]

    def fn_mlp(xm):
        h2 = layernorm_fwd(xm, p[pre + "norm2.weight"], p[pre + "norm2.bias"])[0]
        m = mlp(h2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
        return m * p[pre + "gamma_2"] if pre + "gamma_2" in p else m

    return a, fn_mlp
