"""Round 6 probe: would the optimizer launch (HBM-bound, ~10 GB) hide under the NEXT step's forward if it ran on a side stream?
The real ViT-L + RVSA training-mode forward (GEMMs AND the LayerNorm / RVSA / attention / layout kernels between them) on the compute stream, an AdamW pass of the
model's size on a side stream (ordinary, low priority), whole or in 6 chunks; wall time of forward alone, AdamW alone, both.   python tools/probes/overlap_adamw_forward_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import mtp_amd
from mtp_amd import ops


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = mtp_amd.vit_l_rvsa(type("A", (), dict(image_size=224, use_ckpt=False))()).to(dev).train()
    eng = net._engine()
    img = torch.randn(64, 3, 224, 224, device=dev)
    n = 317_000_000 // 1024 * 1024
    p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    g.normal_()
    seg_start, seg_wd = torch.tensor([0], device=dev, dtype=torch.int64), torch.tensor([0.05], device=dev)
    hyper = torch.tensor([6e-5, 0.9, 0.999, 1e-8, 0.1, 0.001], device=dev)
    sides = {"ordinary side stream": torch.cuda.Stream(), "lowest-priority side stream": ops.low_priority_stream(dev)}

    def fwd():
        eng.forward(img, training=True, need_grad=True)

    def adamw(k=1):
        step = n // k // 1024 * 1024
        for i in range(k):
            a, b = i * step, (n if i == k - 1 else (i + 1) * step)
            ops.adamw_flat(p[a:b], g[a:b], m[a:b], v[a:b], seg_start, seg_wd, hyper)

    def wall(fn, reps=6):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3
    fwd(); adamw()
    tf, ta = wall(fwd), wall(adamw)
    print("forward alone %.2f ms, AdamW-sized pass alone %.2f ms, one after the other %.2f ms" % (tf, ta, wall(lambda: (adamw(), fwd()))))
    for name, side in sides.items():
        for k in (1, 6, 24):
            def both():
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    adamw(k)
                fwd()
                cur.wait_stream(side)
            print("%-30s AdamW in %2d launches beside the forward: %.2f ms" % (name, k, wall(both)))


if __name__ == "__main__":
    main()
