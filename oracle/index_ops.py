"""ORACLE (test infrastructure only) -- integer / index arithmetic of the MTP ViT+RVSA hot path.

numpy restatement of the reference's integer-exact pieces.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product (mtp_amd/) never does.

Reference = /root/reference/Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py ("VIT").
Pinned against golden vectors generated from the reference itself (tests/golden/make_golden.py,
fixture F1) -- see tests/test_oracle_golden.py.
"""
import numpy as np


def block_schedule(depth: int, interval: int):
    """VIT:629 -- block i is a (RVSA) window block unless (i+1) % interval == 0."""
    return [((i + 1) % interval != 0) for i in range(depth)]


def rvsa_geometry(h: int, w: int, ws: int = 7):
    """VIT:298-310 -- symmetric-ish zero padding of an (h, w) token grid to multiples of ws."""
    pad_td = (ws - h % ws) % ws
    pad_lr = (ws - w % ws) % ws
    top, left = pad_td // 2, pad_lr // 2
    return dict(pad_top=top, pad_down=pad_td - top, pad_left=left, pad_right=pad_lr - left,
                He=h + pad_td, We=w + pad_lr, nh=(h + pad_td) // ws, nw=(w + pad_lr) // ws,
                div_x=h // ws, div_y=w // ws)   # VIT:359-360: x-offset / (h//ws), y-offset / (w//ws)


def relative_position_index(ws: int = 7):
    """VIT:271-281 -- Swin-style pairwise index, closed form (a_n-a_m+ws-1)*(2ws-1) + (b_n-b_m+ws-1)."""
    a = np.repeat(np.arange(ws), ws)        # row of flattened position
    b = np.tile(np.arange(ws), ws)          # col
    return ((a[:, None] - a[None, :] + ws - 1) * (2 * ws - 1)
            + (b[:, None] - b[None, :] + ws - 1)).astype(np.int64)


def rel_pos_dist(q: int, k: int):
    """VIT:158-172 -- decomposed rel-pos row index: dist[i, j] = i*max(k/q,1) - j*max(q/k,1) + (k-1)*max(q/k,1)."""
    qr = max(k / q, 1.0)
    kr = max(q / k, 1.0)
    d = np.arange(q)[:, None] * qr - np.arange(k)[None, :] * kr + (k - 1) * kr
    return d.astype(np.int64)


def window_partition(x: np.ndarray, ws: int):
    """VIT:113-124 -- (B,H,W,C) -> (B*nH*nW, ws, ws, C)."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C)
    return np.ascontiguousarray(x.transpose(0, 1, 3, 2, 4, 5)).reshape(-1, ws, ws, C)


def window_reverse(win: np.ndarray, ws: int, H: int, W: int):
    """VIT:127-140 -- inverse of window_partition."""
    B = int(win.shape[0] / (H * W / ws / ws))
    x = win.reshape(B, H // ws, W // ws, ws, ws, -1)
    return np.ascontiguousarray(x.transpose(0, 1, 3, 2, 4, 5)).reshape(B, H, W, -1)


def patch_token_index(H: int, W: int, P: int = 16):
    """VIT:536-539 -- Conv2d(k=P, s=P) + flatten(2).transpose(1,2): token t = py*Wp + px covers
    pixels [py*P, py*P+P) x [px*P, px*P+P); the GEMM K index is (c, ky, kx) row-major.
    Returns (Hp, Wp, idx) with idx[t, ky, kx] = flat pixel index y*W + x."""
    Hp, Wp = H // P, W // P
    py, px = np.divmod(np.arange(Hp * Wp), Wp)
    y = py[:, None, None] * P + np.arange(P)[None, :, None]
    x = px[:, None, None] * P + np.arange(P)[None, None, :]
    return Hp, Wp, (y * W + x).astype(np.int64)


def fpn_nchw_index(Hp: int, Wp: int, levels: int):
    """ConvTranspose2d(k=2,s=2) applied `levels` times (VIT:642-649): GEMM row (t, q_1, .., q_L) with
    q_l = ky_l*2+kx_l lands at pixel (y, x) = (((py*2+ky_1)*2+ky_2).., ...).  Returns flat pixel index per row."""
    n = Hp * Wp * (4 ** levels)
    r = np.arange(n)
    qs = []
    for _ in range(levels):
        r, q = np.divmod(r, 4)
        qs.append(q)
    py, px = np.divmod(r, Wp)
    y, x = py, px
    for q in reversed(qs):
        y = y * 2 + q // 2
        x = x * 2 + q % 2
    return (y * (Wp * 2 ** levels) + x).astype(np.int64)
