"""Full attention beyond 256 tokens: forward and backward time per call (set MTP_NO_FLASH_ATTN=1 for the three-pass f32-math
backward / generic forward).  usage: python tools/ab_full_attn.py [Hp Wp B heads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
Hp, Wp, B, H = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (28, 28, 16, 16)
hd = 64
C, N = H * hd, Hp * Wp
T = B * N
qkv, do = r(T, 3 * C), r(T, C)
o, lse = torch.empty(T, C, device="cuda", dtype=torch.bfloat16), torch.empty(B * H * N, device="cuda")
rh, rw = torch.randn(2 * Hp - 1, hd, device="cuda") * 0.1, torch.randn(2 * Wp - 1, hd, device="cuda") * 0.1
dqkv, d1, d2 = torch.empty_like(qkv), torch.empty_like(rh), torch.empty_like(rw)
try:
    tf = timeit(lambda: ops.full_attn_fwd(qkv, o, lse, rh, rw, B, Hp, Wp, H, 0.125), iters=5)
    tb = timeit(lambda: ops.full_attn_bwd(qkv, o, do, lse, dqkv, rh, rw, d1, d2, B, Hp, Wp, H, 0.125), iters=5)
    fl = 4.0 * B * H * N * N * hd
    print("%dx%d B=%d heads=%d flash=%s: fwd %.1f us (%.0f TF/s)  bwd %.1f us (%.0f TF/s of 2.5x fwd flops)" % (
        Hp, Wp, B, H, "off" if os.environ.get("MTP_NO_FLASH_ATTN") else "on", tf * 1e6, fl / tf / 1e12, tb * 1e6, 2.5 * fl / tb / 1e12))
except RuntimeError as ex:
    print("%dx%d: %s" % (Hp, Wp, ex))
