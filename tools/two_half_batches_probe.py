"""Probe (round 5): the forward of ONE batch of 64 against the forwards of its two halves on two streams.  With two independent launch queues the workgroups of the two halves' GEMMs
interleave on the CUs as they free up, so one half's epilogue bursts and HBM-bound kernels meet the other half's K loops (profiles/r05_epilogue_probe.txt).  Inference mode only
(engine.forward keeps no state between calls besides the read-only weight images).   python tools/two_half_batches_probe.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtp_amd


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = mtp_amd.vit_l_rvsa(type("A", (), dict(image_size=224, use_ckpt=False))()).to(dev)
    eng = net._engine()
    B = 64
    img = torch.randn(B, 3, 224, 224, device=dev)
    halves = [img[:B // 2].contiguous(), img[B // 2:].contiguous()]
    quarters = [img[i * 16:(i + 1) * 16].contiguous() for i in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]

    def whole():
        eng.forward(img, training=False, need_grad=False)

    def split(parts):
        cur = torch.cuda.current_stream()
        for p, s in zip(parts, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                eng.forward(p, training=False, need_grad=False)
        for p, s in zip(parts, streams):
            cur.wait_stream(s)

    def sequential(parts):
        for p in parts:
            eng.forward(p, training=False, need_grad=False)

    def t(fn, n=10):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {k: [] for k in ("whole", "2 halves, 2 streams", "2 halves, 1 stream", "4 quarters, 4 streams")}
    for _ in range(rounds):
        res["whole"].append(t(whole))
        res["2 halves, 2 streams"].append(t(lambda: split(halves)))
        res["2 halves, 1 stream"].append(t(lambda: sequential(halves)))
        res["4 quarters, 4 streams"].append(t(lambda: split(quarters)))
    print("# ViT-L + RVSA forward (inference mode), 64 images, ms per pass: min over %d interleaved rounds of 10 passes" % rounds)
    for k, v in res.items():
        print("%-24s %.3f ms  (%s)" % (k, min(v), " ".join("%.2f" % x for x in v)))


if __name__ == "__main__":
    main()
