"""A/B of NT GEMM kernel variants on the ViT-L shapes (variant bits: see include/mtp_hip.h); checks each variant against variant 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
variants = [int(v) for v in (sys.argv[1:] or ["0", "4"])]
for (M, N, K) in [(T, 3*C, C), (T, C, C), (T, 4*C, C), (T, C, 4*C), (T, C, 3*C)]:
    a, w, out = r(M, K), r(N, K, scale=0.02), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = torch.empty_like(out)
    bias = torch.zeros(N, device="cuda")
    ops.gemm_nt(a, w, ref, bias=bias, variant=0)
    res = []
    for variant in variants:
        ops.gemm_nt(a, w, out, bias=bias, variant=variant)
        ok = torch.equal(out, ref)
        ts = [timeit(lambda: ops.gemm_nt(a, w, out, bias=bias, variant=variant), iters=30) for _ in range(3)]
        res.append("v%d %.0f TF%s" % (variant, 2*M*N*K/min(ts)/1e12, "" if ok else " MISMATCH"))
    print(M, N, K, " | ".join(res), flush=True)
