"""fc1-shaped NT GEMM (12544 x 4096 x 1024) with each epilogue: how much of the in-model slowdown is the epilogue itself?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mtp_amd import ops
from tools.bench_ops import timeit, r
T, C = 12544, 1024
M, N, K = T, 4 * C, C
a, w = r(M, K), r(N, K, scale=0.02)
out, aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(N, device="cuda")
for name, fn in [("bias", lambda: ops.gemm_nt(a, w, out, bias=bias)),
                 ("bias+gelu(+u)", lambda: ops.gemm_nt(a, w, out, epi=ops.EPI_BIAS_GELU, bias=bias, aux=aux)),
                 ("dgelu", lambda: ops.gemm_nt(a, w, out, epi=ops.EPI_DGELU, aux=aux))]:
    ts = [timeit(fn, iters=30) for _ in range(3)]
    print("%-14s %.1f us  %.0f TF" % (name, min(ts) * 1e6, 2 * M * N * K / min(ts) / 1e12), flush=True)
M, N, K = T, C, 4 * C
a, w = r(M, K), r(N, K, scale=0.02)
out32, res = torch.empty(M, N, device="cuda"), torch.randn(M, N, device="cuda")
outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(N, device="cuda")
for name, fn in [("fc2 bias bf16", lambda: ops.gemm_nt(a, w, outb, bias=bias)),
                 ("fc2 bias+res f32", lambda: ops.gemm_nt(a, w, out32, epi=ops.EPI_BIAS_RES, bias=bias, res=res))]:
    ts = [timeit(fn, iters=30) for _ in range(3)]
    print("%-16s %.1f us  %.0f TF" % (name, min(ts) * 1e6, 2 * M * N * K / min(ts) / 1e12), flush=True)
