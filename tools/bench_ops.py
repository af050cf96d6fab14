"""Per-kernel timings at BASELINE config-3 shapes (ViT-L, B=64, 224^2): TF/s for the GEMMs, GB/s for HBM-bound ops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtp_amd import ops

dev = "cuda"
T, C, H = 64 * 196, 1024, 16
bf = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def r(*shape, dtype=bf, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def main():
    which = sys.argv[1:] or ["gemm", "ln", "attn", "misc"]
    if "gemm" in which:
        for dt in (bf, torch.float32):
            for (M, N, K) in [(T, 3 * C, C), (T, C, C), (T, 4 * C, C), (T, C, 4 * C)]:
                a, w, out = r(M, K, dtype=dt), r(N, K, dtype=dt, scale=0.02), torch.empty(M, N, device=dev, dtype=dt)
                bias = torch.zeros(N, device=dev)
                for variant in (0, 1):
                    t = timeit(lambda: ops.gemm_nt(a, w, out, bias=bias, variant=variant))
                    print("gemm_nt %s v%d M=%d N=%d K=%d: %.3f ms  %.1f TF/s" % (str(dt)[6:], variant, M, N, K, t * 1e3, 2 * M * N * K / t / 1e12), flush=True)
            a, w = r(T, C), r(4 * C, C, scale=0.02)
            u, h = torch.empty(T, 4 * C, device=dev, dtype=bf), torch.empty(T, 4 * C, device=dev, dtype=bf)
            if dt == bf:
                t = timeit(lambda: ops.gemm_nt(a, w, h, epi=ops.EPI_BIAS_GELU, bias=torch.zeros(4 * C, device=dev), aux=u))
                print("gemm_nt bf16 GELU fc1: %.3f ms %.1f TF/s" % (t * 1e3, 2 * T * 4 * C * C / t / 1e12))
                res, xo = torch.randn(T, C, device=dev), torch.empty(T, C, device=dev)
                t = timeit(lambda: ops.gemm_nt(h, r(C, 4 * C, scale=0.02), xo, epi=ops.EPI_BIAS_RES, bias=torch.zeros(C, device=dev), res=res))
                print("gemm_nt bf16 RES fc2: %.3f ms %.1f TF/s" % (t * 1e3, 2 * T * 4 * C * C / t / 1e12))
            for (M, N) in [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]:
                a, b, out = r(T, M, dtype=dt), r(T, N, dtype=dt), torch.empty(M, N, device=dev)
                for sk in (None, 1, 4, 8):
                    t = timeit(lambda: ops.gemm_tn(a, b, out, split_k=sk))
                    print("gemm_tn %s M=%d N=%d K=%d split=%s: %.3f ms  %.1f TF/s" % (str(dt)[6:], M, N, T, sk, t * 1e3, 2 * M * N * T / t / 1e12), flush=True)
    if "ln" in which:
        x, g, b = torch.randn(T, C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
        y, mean, rstd = torch.empty(T, C, device=dev, dtype=bf), torch.empty(T, device=dev), torch.empty(T, device=dev)
        t = timeit(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd))
        print("ln_fwd f32->bf16: %.1f us  %.0f GB/s" % (t * 1e6, T * C * 6 / t / 1e9))
        dy, dx, dxc, dg, db = r(T, C), torch.empty(T, C, device=dev), torch.empty(T, C, device=dev, dtype=bf), torch.empty(C, device=dev), torch.empty(C, device=dev)
        dres = torch.randn(T, C, device=dev)
        t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, dres=dres, dx_copy=dxc))
        print("ln_bwd (+2 reduces): %.1f us  %.0f GB/s" % (t * 1e6, T * C * (2 + 4 + 4 + 4 + 2) / t / 1e9))
        dyb = r(T, 4 * C)
        t = timeit(lambda: ops.colsum(dyb, torch.empty(4 * C, device=dev)))
        print("colsum T x 4C bf16: %.1f us %.0f GB/s" % (t * 1e6, T * 4 * C * 2 / t / 1e9))
    if "samp" in which:
        # the RVSA sampling heads of one block: window pooling + LeakyReLU + the stacked (5 * heads, C) linear layer; rotating inputs (in the step ln1 is fresh every block)
        xs = [r(T, C) for _ in range(6)]
        wS, bS = torch.randn(5 * H, C, device=dev) * 0.02, torch.zeros(5 * H, device=dev)
        avg, pooled, samp = torch.empty(256, C, device=dev), torch.empty(256, C, device=dev), torch.empty(256, 5 * H, device=dev)
        it = [0]

        def f():
            it[0] += 1
            ops.rvsa_sampling_fwd(xs[it[0] % 6], wS, bS, avg, pooled, samp, 64, 14, 14)
        t = timeit(f, iters=60)
        print("rvsa_sampling_fwd: %.1f us  (MTP_SAMPLING_YSPLIT=%s)" % (t * 1e6, __import__("os").environ.get("MTP_SAMPLING_YSPLIT", "default")))
    if "attn" in which:
        qkv = r(T, 3 * C)
        o, lse = torch.empty(T, C, device=dev, dtype=bf), torch.empty(64 * H * 196, device=dev)
        rh, rw = torch.randn(27, 64, device=dev) * 0.1, torch.randn(27, 64, device=dev) * 0.1
        t = timeit(lambda: ops.full_attn_fwd(qkv, o, lse, rh, rw, 64, 14, 14, H, 0.125))
        fl = 4 * 196 * 196 * 64 * 64 * H
        print("full_attn_fwd: %.3f ms  %.1f TF/s" % (t * 1e3, fl / t / 1e12))
        do, dqkv = r(T, C), torch.empty(T, 3 * C, device=dev, dtype=bf)
        drh, drw = torch.empty(27, 64, device=dev), torch.empty(27, 64, device=dev)
        t = timeit(lambda: ops.full_attn_bwd(qkv, o, do, lse, dqkv, rh, rw, drh, drw, 64, 14, 14, H, 0.125), iters=5)
        print("full_attn_bwd: %.3f ms" % (t * 1e3))
        samp = torch.randn(256, 5 * H, device=dev) * 0.2
        lse2 = torch.empty(256 * H * 49, device=dev)
        r13, tab = torch.randn(13, 64, device=dev) * 0.1, torch.randn(169, H, device=dev) * 0.1
        t = timeit(lambda: ops.rvsa_attn_fwd(qkv, samp, o, lse2, r13, r13, tab, 64, 14, 14, H, 0.125))
        print("rvsa_attn_fwd: %.3f ms  %.1f TF/s" % (t * 1e3, 4 * 49 * 49 * 64 * 256 * H / t / 1e12))
        dsamp, d13a, d13b, dtab = torch.empty(256, 5 * H, device=dev), torch.empty(13, 64, device=dev), torch.empty(13, 64, device=dev), torch.empty(169, H, device=dev)
        t = timeit(lambda: ops.rvsa_attn_bwd(qkv, samp, o, do, lse2, dqkv, dsamp, r13, r13, tab, d13a, d13b, dtab, 64, 14, 14, H, 0.125), iters=5)
        print("rvsa_attn_bwd: %.3f ms" % (t * 1e3))
    if "misc" in which:
        x = r(16 * T, C)
        f = torch.empty(64, C, 56, 56, device=dev, dtype=bf)
        t = timeit(lambda: ops.tokens_to_nchw(x, f, 64, 14, 14, 2))
        print("tokens_to_nchw L2 bf16: %.1f us %.0f GB/s" % (t * 1e6, 16 * T * C * 4 / t / 1e9))
        t = timeit(lambda: ops.nchw_to_tokens(f, x, 64, 14, 14, 2))
        print("nchw_to_tokens L2 bf16: %.1f us %.0f GB/s" % (t * 1e6, 16 * T * C * 4 / t / 1e9))
        w = torch.randn(4 * C, C, device=dev)
        wt = torch.empty(C, 4 * C, device=dev, dtype=bf)
        t = timeit(lambda: ops.transpose_cast(w, wt))
        print("transpose_cast 4CxC: %.1f us %.0f GB/s" % (t * 1e6, 4 * C * C * 6 / t / 1e9))
        xs = r(T, C)
        avg, pooled = torch.empty(256, C, device=dev), torch.empty(256, C, device=dev)
        t = timeit(lambda: ops.rvsa_pool_fwd(xs, avg, pooled, 64, 14, 14))
        print("rvsa_pool_fwd: %.1f us" % (t * 1e6))
        img = torch.randn(64, 3, 224, 224, device=dev)
        cols = torch.empty(T, 768, device=dev, dtype=bf)
        t = timeit(lambda: ops.patchify(img, cols))
        print("patchify: %.1f us %.0f GB/s" % (t * 1e6, T * 768 * 6 / t / 1e9))
    if "dcnv3" in which:
        # InternImage-XL at 512^2 (channels 192, groups 12/24/48/96 x 16 channels, 3x3, offset_scale 2), batch 8;
        # also the reference's own timing case (ops_dcnv3/test.py:check_time_cost: N=512, 64x64, 4 groups x 16).
        # GB/s = algorithmic bytes (input + offset + mask + output once; backward: + grad_output, the three f32 gradients, grad_input twice)
        from mtp_amd.ops_dcnv3 import dcnv3_backward, dcnv3_forward
        # offsets: "zero" = a freshly initialised network (the offset head is zero-initialised, ops_dcnv3/modules/dcnv3.py:176-179) -- what bench.py
        # runs; "smooth" = +-0.25 px; "random" = +-2 px (x offset_scale 2: most samples leave the gather form's reach and take the atomic path).
        # "scatter" = the per-corner f32-atomic backward (MTP_DCNV3_VARIANT=2), the form used for every geometry before round 3; "3x3" = the narrow-reach
        # form of the gather backward (MTP_DCNV3_VARIANT=4); plus offsets drawn like bench.py's heads (sigma 0.55 ... 1.57 px after the offset scale).
        import os
        for (N, HW, M) in [(8, 128, 12), (8, 64, 24), (8, 32, 48), (8, 16, 96), (512, 64, 4)]:
            for dt in (bf, torch.float32):
                e = 2 if dt == bf else 4
                x = r(N, HW, HW, M * 16, dtype=dt)
                m = torch.softmax(torch.randn(N, HW, HW, M, 9, device=dev), -1).reshape(N, HW, HW, M * 9).to(dt)
                G = r(N, HW, HW, M * 16, dtype=dt)
                a = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, 2.0)
                px = N * HW * HW
                fb = px * M * (16 * e * 2 + 27 * e)
                bb = px * M * (16 * e * 2 + 27 * e + 16 * 4 + 27 * 4)
                for kind, amp in (("zero", 0.0), ("smooth", 0.5), ("bench", -1.0), ("random", 4.0)):
                    if amp < 0:      # what bench.py's re-drawn heads produce: N(0, (0.02 sqrt(C))^2) before the offset scale
                        off = (torch.randn(N, HW, HW, M * 18, device=dev) * 0.02 * (M * 16) ** 0.5).to(dt)
                    else:
                        off = ((torch.rand(N, HW, HW, M * 18, device=dev) - 0.5) * amp).to(dt)
                    t = timeit(lambda: dcnv3_forward(x, off, m, *a, 256, 0), iters=10)
                    cells = ["fwd %.1f us %.0f GB/s" % (t * 1e6, fb / t / 1e9)]
                    for name, var in (("bwd", "0"), ("bwd-3x3", "4"), ("bwd-scatter", "2")):
                        os.environ["MTP_DCNV3_VARIANT"] = var
                        t = timeit(lambda: dcnv3_backward(x, off, m, *a, G, 256, 0), iters=10)
                        cells.append("%s %.1f us %.0f GB/s" % (name, t * 1e6, bb / t / 1e9))
                    os.environ["MTP_DCNV3_VARIANT"] = "0"
                    print("dcnv3 %s N=%d %dx%d groups=%d offsets=%s: %s" % (str(dt)[6:], N, HW, HW, M, kind, " | ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
