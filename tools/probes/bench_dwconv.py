"""depth-wise 3x3 convolution kernels at InternImage-XL's level shapes (N = 8, 512^2 input): forward, data gradient, weight-gradient partials.  python tools/probes/bench_dwconv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mtp_amd import ops

bf = torch.bfloat16


def t(fn, k=30):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(k):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / k * 1e3)
    return best


def main():
    print("# lib = %s" % (os.environ.get("MTP_HIP_LIB") or "this tree"))
    for (N, H, W, C) in ((8, 128, 128, 192), (8, 64, 64, 384), (8, 32, 32, 768), (8, 16, 16, 1536)):
        x = (torch.randn(N * H * W, C, device="cuda")).to(bf)
        dy = (torch.randn(N * H * W, C, device="cuda")).to(bf)
        w, b = torch.randn(C, 1, 3, 3, device="cuda"), torch.randn(C, device="cuda")
        y, dx = torch.empty_like(x), torch.empty(N * H * W, C, device="cuda")
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        print("N=%d %3dx%-3d C=%4d | fwd %6.1f us | bwd_dx %6.1f us | bwd_dw (+ reduction) %6.1f us" % (
            N, H, W, C, t(lambda: ops.dwconv3x3_fwd(x, w, b, y, N, H, W)), t(lambda: ops.dwconv3x3_bwd_dx(dy, w, dx, N, H, W)),
            t(lambda: ops.dwconv3x3_bwd_dw(dy, x, dw, db, N, H, W))), flush=True)


if __name__ == "__main__":
    main()
