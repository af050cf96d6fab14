"""A/B of the DCNv3 forward kernels at InternImage-XL's four levels (512^2, batch 8, bf16): the unrolled 9-point forward (default) against the generic one
(MTP_DCNV3_VARIANT=8), interleaved; outputs compared.  python tools/probes/dcn_fwd_ab.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mtp_amd.ops_dcnv3 import dcnv3_forward

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


for dt in (torch.bfloat16, torch.float32):
    for (N, HW, M) in [(8, 128, 12), (8, 64, 24), (8, 32, 48), (8, 16, 96)]:
        x = torch.randn(N, HW, HW, M * 16, device=dev).to(dt)
        m = torch.softmax(torch.randn(N, HW, HW, M, 9, device=dev), -1).reshape(N, HW, HW, M * 9).to(dt)
        off = (torch.randn(N, HW, HW, M * 18, device=dev) * 0.02 * (M * 16) ** 0.5).to(dt)
        a = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, 2.0)
        res = {}
        ts = {"0": [], "8": []}
        for rnd in range(3):
            for var in ("0", "8"):
                os.environ["MTP_DCNV3_VARIANT"] = var
                ts[var].append(timeit(lambda: dcnv3_forward(x, off, m, *a, 256, 0)) * 1e6)
                res[var] = dcnv3_forward(x, off, m, *a, 256, 0).float()
        os.environ["MTP_DCNV3_VARIANT"] = "0"
        d = float((res["0"] - res["8"]).abs().max() / res["8"].abs().max())
        print("%s %dx%d groups %d: unrolled %s us   generic %s us   max rel diff %.1e" % (str(dt)[6:], HW, HW, M, " ".join("%.1f" % t for t in ts["0"]),
                                                                                    " ".join("%.1f" % t for t in ts["8"]), d), flush=True)
