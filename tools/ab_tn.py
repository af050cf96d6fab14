import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
for (M, N) in [(3*C, C), (C, C), (4*C, C), (C, 4*C)]:
    a, b, out = r(T, M), r(T, N), torch.empty(M, N, device="cuda")
    for variant in (0, 8):
        res = []
        for sk in (1, 2, 3, 4, 6, 8, 12):
            ts = [timeit(lambda: ops.gemm_tn(a, b, out, split_k=sk, variant=variant), iters=20) for _ in range(2)]
            res.append("s%d %.0f" % (sk, 2*M*N*T/min(ts)/1e12))
        print(M, N, "v%d" % variant, " | ".join(res), flush=True)
