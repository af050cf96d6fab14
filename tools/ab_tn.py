import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
for (M, N) in [(3*C, C), (C, C), (4*C, C), (C, 4*C)]:
    a, b, out = r(T, M), r(T, N), torch.empty(M, N, device="cuda")
    res = []
    for sk in (1, 2, 3, 4, 5, 6, 8, 12):
        ts = [timeit(lambda: ops.gemm_tn(a, b, out, split_k=sk), iters=20) for _ in range(2)]
        res.append("s%d %.0f" % (sk, 2*M*N*T/min(ts)/1e12))
    print(M, N, " | ".join(res), "| auto", ops.pick_split_k(M, N, T), flush=True)
# GELU epilogue cost
a, w = r(T, C), r(4*C, C, scale=0.02)
u, h = torch.empty(T, 4*C, device="cuda", dtype=torch.bfloat16), torch.empty(T, 4*C, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(4*C, device="cuda")
t0 = timeit(lambda: ops.gemm_nt(a, w, h, bias=bias), iters=30)
t1 = timeit(lambda: ops.gemm_nt(a, w, h, epi=ops.EPI_BIAS_GELU, bias=bias, aux=u), iters=30)
t2 = timeit(lambda: ops.gemm_nt(a, w, h, epi=ops.EPI_DGELU, aux=u), iters=30)
print("fc1 plain %.1f us, GELU %.1f us, DGELU-shaped %.1f us" % (t0*1e6, t1*1e6, t2*1e6))
dy = r(T, 3*C)
print("colsum T x 3C: %.1f us" % (timeit(lambda: ops.colsum(dy, torch.empty(3*C, device='cuda')))*1e6))
