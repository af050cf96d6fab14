"""A/B of the weight-gradient (TN) GEMM kernels / tile orders on the ViT-L shapes.
variant 0 = transpose-read kernel, automatic tile order; 6 = same kernel, N-fastest order forced; 4 = M-fastest forced;
16 = register-transposing kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
variants = [int(v) for v in (sys.argv[1:] or ["0", "6", "4"])]
for (M, N) in [(3*C, C), (C, C), (4*C, C), (C, 4*C), (C, 768)]:
    a, b = r(T, M), r(T, N)
    out, ref = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm_tn(a, b, ref, split_k=4, variant=16)
    for variant in variants:
        ops.gemm_tn(a, b, out, split_k=4, variant=variant)
        torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        res = []
        for sk in (2, 4, 8):
            ts = [timeit(lambda: ops.gemm_tn(a, b, out, split_k=sk, variant=variant), iters=20) for _ in range(2)]
            res.append("s%d %.0f" % (sk, 2*M*N*T/min(ts)/1e12))
        print(M, N, "v%d" % variant, " | ".join(res), "" if ok else "MISMATCH", flush=True)
