"""RVSA forward / backward at the ViT-L B=64 launch geometry (MTP_RVSA_SCATTER=gemm|dense|corner|none, MTP_RVSA_STOP=1..6 for phase timing)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
B, H, hd, Hp, Wp = 64, 16, 64, 14, 14
C, T = H * hd, B * Hp * Wp
qkv = r(T, 3 * C); o = r(T, C); do = r(T, C)
samp = torch.randn(B * 4, 5 * H, device="cuda") * float(os.environ.get("SAMP", "0.2"))
lse = torch.empty(B * 4 * H * 49, device="cuda")
r13, tab = torch.randn(13, 64, device="cuda") * 0.1, torch.randn(169, H, device="cuda") * 0.1
ops.rvsa_attn_fwd(qkv, samp, o, lse, r13, r13, tab, B, Hp, Wp, H, 0.125)
dqkv, dsamp = torch.empty(T, 3 * C, device="cuda", dtype=torch.bfloat16), torch.empty(B * 4, 5 * H, device="cuda")
d1, d2, dt = torch.empty(13, 64, device="cuda"), torch.empty(13, 64, device="cuda"), torch.empty(169, H, device="cuda")
t = timeit(lambda: ops.rvsa_attn_bwd(qkv, samp, o, do, lse, dqkv, dsamp, r13, r13, tab, d1, d2, dt, B, Hp, Wp, H, 0.125), iters=10)
print("rvsa_attn_bwd (kernel + scatter + 2 reduces) SCATTER=%s: %.1f us" % (os.environ.get("MTP_RVSA_SCATTER", "gemm (default)"), t * 1e6))
t = timeit(lambda: ops.rvsa_attn_fwd(qkv, samp, o, lse, r13, r13, tab, B, Hp, Wp, H, 0.125), iters=10)
print("rvsa_attn_fwd: %.1f us" % (t * 1e6))
