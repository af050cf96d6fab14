"""Is the NT kernel's epilogue burst bound per CU or by the chip?  Time T(K) of the 8-phase kernel at fixed (M, N) for several K and fit T = a + b K: `a` is everything that is not
the K loop (prologue + epilogue).  Same N, three M: 224 / 112 / 56 tiles of 224 x 256 -- one tile per CU each, on all / half / a quarter of the CUs.  If `a` shrinks with fewer
active CUs the burst is limited by the shared write path (HBM / fabric), not by a CU's own store issue.   python tools/probes/epilogue_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mtp_amd import ops

bf = torch.bfloat16
NROT = 8


def timed(fn, iters=30):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


def main():
    N = 1024
    for epi_name in ("bias_bf16", "bias_res_f32"):
        for M in (12544, 6272, 3136):
            ts = []
            for K in (768, 1024, 1536, 2048, 3072, 4096):
                a = (torch.randn(M, K, device="cuda") * 0.5).to(bf)
                w = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
                bias = torch.randn(N, device="cuda")
                if epi_name == "bias_bf16":
                    outs = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(NROT)]
                    kw = dict(bias=bias)
                else:
                    outs = [torch.empty(M, N, device="cuda") for _ in range(NROT)]
                    res = [torch.randn(M, N, device="cuda") for _ in range(NROT)]
                    kw = dict(bias=bias, epi=ops.EPI_BIAS_RES)
                i = [0]

                def run():
                    i[0] = (i[0] + 1) % NROT
                    k2 = dict(kw)
                    if epi_name != "bias_bf16":
                        k2["res"] = res[i[0]]
                    ops.gemm_nt(a, w, outs[i[0]], variant=512, **k2)
                ts.append((K, timed(run)))
            n = len(ts)
            sx, sy = sum(k for k, _ in ts), sum(t for _, t in ts)
            sxx, sxy = sum(k * k for k, _ in ts), sum(k * t for k, t in ts)
            b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
            a0 = (sy - b * sx) / n
            tiles = -(-M // 224) * (N // 256)
            print("%-13s M=%5d (%3d tiles of 224 x 256) | " % (epi_name, M, tiles) + "  ".join("K=%d %.1f" % kt for kt in ts) + " us | fit: %.1f us + %.2f us per 64 of K" % (a0, b * 64), flush=True)


if __name__ == "__main__":
    main()
