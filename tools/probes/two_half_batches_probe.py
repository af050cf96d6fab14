"""Probe: the forward of ONE batch of 64 against the forwards of its two halves on two streams -- plain streams (round 5: the hardware interleaves the
workgroups of the two halves on all CUs) and CU-masked streams (round 6, VERDICT r05 #1: each half owns 128 CUs, 16 of every XCC, so M/2 rows on half the CUs keep the
whole-batch tile quantisation and one half's HBM-bound kernels meet the other half's K loops).  Inference mode only (engine.forward keeps no state between calls
besides the read-only weight images).   python tools/probes/two_half_batches_probe.py [rounds] [--cu-mask]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import mtp_amd
from mtp_amd import ops


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(args[0]) if args else 5
    masked = "--cu-mask" in sys.argv
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

    def split(parts, sts, stagger_blocks=0):
        cur = torch.cuda.current_stream()
        for s in sts[:len(parts)]:
            s.wait_stream(cur)
        for k, (p, s) in enumerate(zip(parts, sts)):
            with torch.cuda.stream(s):
                if k and stagger_blocks:
                    torch.cuda._sleep(int(stagger_blocks))      # delay the second half by about this many clocks
                eng.forward(p, training=False, need_grad=False)
        for p, s in zip(parts, sts):
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

    cases = {"whole": whole, "2 halves, 2 streams": lambda: split(halves, streams), "2 halves, 1 stream": lambda: sequential(halves)}
    if masked:
        for kind in ("interleaved", "blocked"):
            ms = [ops.cu_mask_stream(dev, ops.cu_mask_words(kind, k)) for k in range(2)]
            cases["2 halves, 2 CU-masked streams (%s)" % kind] = (lambda ms=ms: split(halves, ms))
            cases["2 halves, 2 CU-masked streams (%s), second half 200 us late" % kind] = (lambda ms=ms: split(halves, ms, 400000))
        ms = [ops.cu_mask_stream(dev, ops.cu_mask_words("interleaved", k)) for k in range(2)]
        with torch.cuda.stream(ms[0]):
            pass
        cases["half a batch alone on a 128-CU stream"] = (lambda ms=ms: split(halves[:1], ms[:1]))
        cases["half a batch alone on the whole chip"] = (lambda: sequential(halves[:1]))
        q = [ops.cu_mask_stream(dev, ops.cu_mask_words("interleaved", k, parts=4)) for k in range(4)]
        cases["4 quarters, 4 CU-masked streams (64 CUs each)"] = (lambda q=q: split(quarters, q))
    else:
        cases["4 quarters, 4 streams"] = lambda: split(quarters, streams)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if only:      # for rocprofv3 --kernel-trace --stats: a few passes of the named cases only
        for k, fn in cases.items():
            if k in only:
                print(k, "%.3f ms" % t(fn, 5))
        return
    res = {k: [] for k in cases}
    for _ in range(rounds):
        for k, fn in cases.items():
            res[k].append(t(fn))
    print("# ViT-L + RVSA forward (inference mode), 64 images, ms per pass: min over %d interleaved rounds of 10 passes" % rounds)
    for k, v in res.items():
        print("%-78s %.3f ms  (%s)" % (k, min(v), " ".join("%.2f" % x for x in v)))


if __name__ == "__main__":
    main()
