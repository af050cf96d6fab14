"""A/B of the weight-gradient (TN) GEMM kernels: variant 0 = LDS-DMA + transpose reads, 16 = register transposes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
for (M, N) in [(3*C, C), (C, C), (4*C, C), (C, 4*C)]:
    a, b = r(T, M), r(T, N)
    out, ref = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm_tn(a, b, ref, split_k=4, variant=16)
    ops.gemm_tn(a, b, out, split_k=4, variant=0)
    torch.cuda.synchronize()
    print(M, N, "max |tr - regs| =", float((out - ref).abs().max()), "ref max", float(ref.abs().max()), flush=True)
    for variant in (0, 16):
        res = []
        for sk in (1, 2, 3, 4, 6, 8, 12):
            ts = [timeit(lambda: ops.gemm_tn(a, b, out, split_k=sk, variant=variant), iters=20) for _ in range(2)]
            res.append("s%d %.0f" % (sk, 2*M*N*T/min(ts)/1e12))
        print(M, N, "v%d" % variant, " | ".join(res), flush=True)
