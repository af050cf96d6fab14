"""Op-level run-to-run screen of the RVSA backward: same inputs N times, every output against run 0 (max abs diff / max abs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mtp_amd import ops
from tools.bench_ops import r
B, H, hd, Hp, Wp = int(os.environ.get("B", "2")), 4, 64, 14, 14
C, T = H * hd, B * Hp * Wp
torch.manual_seed(0)
qkv = r(T, 3 * C); o = r(T, C); do = r(T, C)
samp = torch.randn(B * 4, 5 * H, device="cuda") * float(os.environ.get("SAMP", "0.2"))
lse = torch.empty(B * 4 * H * 49, device="cuda")
r13, tab = torch.randn(13, 64, device="cuda") * 0.1, torch.randn(169, H, device="cuda") * 0.1
ops.rvsa_attn_fwd(qkv, samp, o, lse, r13, r13, tab, B, Hp, Wp, H, 0.125)
ref = None
worst = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    dqkv, dsamp = torch.empty(T, 3 * C, device="cuda", dtype=torch.bfloat16), torch.empty(B * 4, 5 * H, device="cuda")
    d1, d2, dt = torch.empty(13, 64, device="cuda"), torch.empty(13, 64, device="cuda"), torch.empty(169, H, device="cuda")
    ops.rvsa_attn_bwd(qkv, samp, o, do, lse, dqkv, dsamp, r13, r13, tab, d1, d2, dt, B, Hp, Wp, H, 0.125)
    torch.cuda.synchronize()
    out = {"dq": dqkv[:, :C].float(), "dk": dqkv[:, C:2 * C].float(), "dv": dqkv[:, 2 * C:].float(), "dsamp": dsamp, "drel_h": d1, "drel_w": d2, "dtable": dt}
    if ref is None:
        ref = {k: v.clone() for k, v in out.items()}
        continue
    for k, v in out.items():
        d = float((v - ref[k]).abs().max() / ref[k].abs().max())
        worst[k] = max(worst.get(k, 0.0), d)
print(os.environ.get("MTP_RVSA_SCATTER", "dense"), " ".join("%s=%.2e" % kv for kv in worst.items()))
