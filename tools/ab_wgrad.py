"""A/B of the grouped weight-gradient launch (gemm_tn_p8.hip) on the weight gradients of N ViT-L blocks: the plain phases of rounds 2-4 (variant bit 19)
against the read-ahead phases (default since round 5: the next phase's transpose reads issued under the current phase's MFMAs), interleaved rounds in one process,
results compared bit for bit.  usage: python tools/ab_wgrad.py [rounds] [blocks ...]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mtp_amd import ops
from tools.bench_ops import r

T, C = 12544, 1024
SHAPES = [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]      # qkv, proj, fc1, fc2: dW (M, N) = dY (T, M)^T X (T, N)
PLAIN = 1 << 19


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
    for nblk in [int(v) for v in sys.argv[2:]] or [2, 4]:
        probs = []
        for _ in range(nblk):
            for (M, N) in SHAPES:
                probs.append((r(T, M), r(T, N), torch.empty(M, N, device="cuda"), torch.zeros(M, device="cuda")))
        fl = sum(2.0 * T * a.shape[1] * b.shape[1] for a, b, _, _ in probs)

        def grouped(variant):
            q = ops.WgradQueue(variant=variant)
            for a, b, dw, cs in probs:
                q.add(a, b, dw, cs)
            q.flush()

        grouped(PLAIN)
        ref = [p[2].clone() for p in probs]
        grouped(0)
        same = all(torch.equal(p[2], x) for p, x in zip(probs, ref))
        t0, t1 = [], []
        for _ in range(rounds):
            t0.append(timed(lambda: grouped(PLAIN), 5))
            t1.append(timed(lambda: grouped(0), 5))
        tiles = sum((a.shape[1] // 256) * (b.shape[1] // 256) for a, b, _, _ in probs)
        print("%d block(s), %d tiles: plain %.1f us %.0f TF/s (min %.1f) | read-ahead %.1f us %.0f TF/s (min %.1f) %s" % (
            nblk, tiles, statistics.median(t0) * 1e6, fl / statistics.median(t0) / 1e12, min(t0) * 1e6,
            statistics.median(t1) * 1e6, fl / statistics.median(t1) / 1e12, min(t1) * 1e6, "bit-identical" if same else "MISMATCH"), flush=True)


if __name__ == "__main__":
    main()
