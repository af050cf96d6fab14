"""Weight gradients of N ViT-L blocks: ONE grouped launch (gemm_tn_p8.hip: 256 x 256 tiles, full contraction per workgroup)
vs the split-K kernels of gemm.hip (one launch per gradient + the batched partial-tile reduction), same inputs, interleaved
rounds.  usage: python tools/ab_wgrad.py [rounds] [blocks ...]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mtp_amd import ops
from tools.bench_ops import r

T, C = 12544, 1024
SHAPES = [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]      # qkv, proj, fc1, fc2: dW (M, N) = dY (T, M)^T X (T, N)


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for nblk in [int(v) for v in sys.argv[2:]] or [1, 2, 4]:
        probs = []
        for _ in range(nblk):
            for (M, N) in SHAPES:
                probs.append((r(T, M), r(T, N), torch.empty(M, N, device="cuda"), torch.zeros(M, device="cuda")))
        fl = sum(2.0 * T * a.shape[1] * b.shape[1] for a, b, _, _ in probs)

        def grouped(variant=0):
            q = ops.WgradQueue(variant=variant)
            for a, b, dw, cs in probs:
                q.add(a, b, dw, cs)
            q.flush()

        def split():
            pend = []
            for a, b, dw, cs in probs:
                ops.gemm_tn(a, b, dw, colsum=cs, defer=pend)
            ops.sum_partials(pend)

        grouped(64); split()
        ref = [p[2].clone() for p in probs]
        grouped(32)
        err = max(float((p[2] - x).abs().max() / x.abs().max()) for p, x in zip(probs, ref))
        tg, tw, ts, abl = [], [], [], {"nostagger": [], "nomfma": []}
        for _ in range(rounds):
            tg.append(timed(lambda: grouped(64), 5))
            tw.append(timed(lambda: grouped(32), 5))
            ts.append(timed(split, 5))
            abl["nostagger"].append(timed(lambda: grouped(64 + (2 << 11)), 3))
            abl["nomfma"].append(timed(lambda: grouped(64 + (8 << 11)), 3))
        print("   ablations: " + "  ".join("%s %.1f us" % (k, statistics.median(v) * 1e6) for k, v in abl.items()))
        print("   4-wave 32x32x16 form (gemm_tn_w4.hip): %.1f us  %.0f TF/s (min %.1f us), max rel. difference to the 8-wave form %.2e" % (
            statistics.median(tw) * 1e6, fl / statistics.median(tw) / 1e12, min(tw) * 1e6, err))
        print("%d block(s), %d tiles: grouped %.1f us  %.0f TF/s (min %.1f us) | split-K %.1f us  %.0f TF/s" % (
            nblk, sum((a.shape[1] // 256) * (b.shape[1] // 256) for a, b, _, _ in probs), statistics.median(tg) * 1e6, fl / statistics.median(tg) / 1e12,
            min(tg) * 1e6, statistics.median(ts) * 1e6, fl / statistics.median(ts) / 1e12), flush=True)


if __name__ == "__main__":
    main()
