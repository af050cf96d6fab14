import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
variants = [int(v) for v in (sys.argv[1:] or ["0", "8"])]
for (M, N, K) in [(T, 3*C, C), (T, C, C), (T, 4*C, C), (T, C, 4*C)]:
    a, w, out = r(M, K), r(N, K, scale=0.02), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    res = []
    for variant in variants:
        ts = [timeit(lambda: ops.gemm_nt(a, w, out, bias=bias, variant=variant), iters=30) for _ in range(3)]
        res.append("v%d %.0f TF" % (variant, 2*M*N*K/min(ts)/1e12))
    print(M, N, K, " | ".join(res), flush=True)
