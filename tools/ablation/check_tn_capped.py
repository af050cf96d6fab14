"""the capped (persistent) form of mtp_gemm_tn_grouped against the one-tile-per-workgroup form: same problems, bit-identical results, both timed"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mtp_amd import ops

torch.manual_seed(0)
T, C = 12544, 1024
shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C), (264, 1000), (8, 8)]
jobs = []
for M, N in shapes:
    dy = (torch.randn(T, M, device="cuda") * 0.1).bfloat16()
    x = (torch.randn(T, N, device="cuda") * 0.1).bfloat16()
    jobs.append((dy, x))


def run(variant):
    q = ops.WgradQueue(variant=variant)
    outs, sums = [], []
    for dy, x in jobs:
        dw = torch.empty(dy.shape[1], x.shape[1], device="cuda")
        cs = torch.zeros(dy.shape[1], device="cuda")
        q.add(dy, x, dw, cs)
        outs.append(dw); sums.append(cs)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); q.flush(); e.record(); torch.cuda.synchronize()
    return outs, sums, s.elapsed_time(e)


ref, refs, t0 = run(0)
ref, refs, t0 = run(0)
for cap in (32, 7, 1000):
    out, sums, t = run(cap << 8)
    same = all(torch.equal(a, b) for a, b in zip(ref, out))
    cs_ok = all(torch.allclose(a, b, rtol=1e-4, atol=1e-3) for a, b in zip(refs, sums))
    print("cap %4d: %8.1f us (uncapped %.1f us)  bit-identical %s  colsum ok %s" % (cap, t * 1e3, t0 * 1e3, same, cs_ok))
    assert same and cs_ok
