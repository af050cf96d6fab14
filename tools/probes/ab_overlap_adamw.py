"""Does the AdamW pass (HBM-bound, 8.9 GB) hide under a forward-like chain of NT GEMMs when it runs on a second stream?
One process: (a) the GEMM chain alone, (b) AdamW alone, (c) both started together -- wall time of each (HIP events on a third stream would
not see both; host timers around a device synchronize).  usage: python tools/probes/ab_overlap_adamw.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mtp_amd import ops
from tools.bench_ops import r


def main():
    T, C = 12544, 1024
    bf = torch.bfloat16
    x = r(T, C)
    h = r(T, 4 * C)
    ws = [(r(3 * C, C, scale=0.02), x, torch.empty(T, 3 * C, device="cuda", dtype=bf)), (r(C, C, scale=0.02), x, torch.empty(T, C, device="cuda", dtype=bf)),
          (r(4 * C, C, scale=0.02), x, torch.empty(T, 4 * C, device="cuda", dtype=bf)), (r(C, 4 * C, scale=0.02), h, torch.empty(T, C, device="cuda", dtype=bf))]
    n = 304_000_000
    p, g, m, v = (torch.zeros(n, device="cuda") for _ in range(4))
    g.normal_()
    seg_start = torch.tensor([0], device="cuda", dtype=torch.int64)
    seg_wd = torch.tensor([0.05], device="cuda")
    hyper = torch.tensor([6e-5, 0.9, 0.999, 1e-8, 0.1, 0.001], device="cuda")
    side = torch.cuda.Stream()

    def chain():
        for _ in range(24):
            for w, a, o in ws:
                ops.gemm_nt(a, w, o)

    def adamw():
        ops.adamw_flat(p, g, m, v, seg_start, seg_wd, hyper)

    def wall(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    def both():
        with torch.cuda.stream(side):
            adamw()
        chain()

    def both_chunked(k=8):
        step = n // k // 1024 * 1024
        with torch.cuda.stream(side):
            for i in range(k):
                a, b = i * step, (n if i == k - 1 else (i + 1) * step)
                ops.adamw_flat(p[a:b], g[a:b], m[a:b], v[a:b], seg_start, seg_wd, hyper)
        chain()
    chain(); adamw()
    ta, tb, tc, td = wall(chain), wall(adamw), wall(both), wall(both_chunked)
    print("GEMM chain alone %.2f ms | AdamW alone %.2f ms | together %.2f ms (sum %.2f) | together, AdamW in 8 launches %.2f ms" % (ta, tb, tc, ta + tb, td))


if __name__ == "__main__":
    main()
