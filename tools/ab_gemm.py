"""A/B of the NT GEMM kernel families / variants on the ViT-L training shapes (variant bits: include/mtp_hip.h).
Interleaved rounds in ONE process (guide rule 24), random operands (rule 25), every variant checked bit for bit against the
128-wide kernels.  usage: python tools/ab_gemm.py [rounds] [variant ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import statistics

import torch

from mtp_amd import ops
from tools.bench_ops import r

T, C = 12544, 1024
NT_, SC1, PLAIN = 1 << 20, 2 << 20, 3 << 20
S8 = 1 << 17
NAMES = {S8: "s8-strip", -1: "hipBLASLt(torch.addmm)",  512 + NT_: "p8-224-nt", 512 + SC1: "p8-224-sc1", 512 + PLAIN: "p8-224-plain", 512 + 65536: "p8-224-oneshot", 1024: "w128", 256 + 32768: "p8-persist", 512 + 32768: "p8-224-persist", 768 + 32768: "p8-256-persist", 512 + 65536: "p8-224-oneshot", 768 + 65536: "p8-256-oneshot", 256: "p8-auto", 512: "p8-224", 768: "p8-256", 258: "p8-plain"}


def time_many(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    variants = [int(v) for v in sys.argv[2:]] or [1024, 512, 768]
    bf = torch.bfloat16
    cases = []
    shapes = [(T, 3 * C, C), (T, C, C), (T, 4 * C, C), (T, C, 4 * C), (T, C, 3 * C), (4 * T, 4 * C, C), (T, C, 768)]
    if os.environ.get("MTP_AB_SHAPES") == "mid":      # ViT-B at batch 32 and InternImage-XL's 768- / 1536-channel levels
        shapes = [(6272, 2304, 768), (6272, 768, 768), (6272, 3072, 768), (6272, 768, 3072), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (2048, 1536, 1536), (2048, 6144, 1536), (2048, 1536, 6144)]
    for (M, N, K) in shapes:
        cases.append(("bias", M, N, K))
    cases += [("gelu", T, 4 * C, C), ("dgelu", T, 4 * C, C), ("gelu_dg", T, 4 * C, C), ("mul", T, 4 * C, C), ("res", T, C, C), ("res", T, C, 4 * C)]
    for (epi, M, N, K) in cases:
        a, w = r(M, K), r(N, K, scale=0.02)
        bias = torch.randn(N, device="cuda")
        kw = dict(bias=bias)
        odt = bf
        if epi == "gelu":
            kw.update(epi=ops.EPI_BIAS_GELU, aux=torch.empty(M, N, device="cuda", dtype=bf))
        elif epi == "dgelu":
            kw = dict(epi=ops.EPI_DGELU, aux=r(M, N))
        elif epi == "gelu_dg":      # what the engine's fc1 forward runs (gelu + gelu' out)
            kw.update(epi=ops.EPI_BIAS_GELU_DG, aux=torch.empty(M, N, device="cuda", dtype=bf))
        elif epi == "mul":          # ... and its fc2 dgrad
            kw = dict(epi=ops.EPI_MUL, aux=r(M, N))
        elif epi == "res":
            kw.update(epi=ops.EPI_BIAS_RES, res=torch.randn(M, N, device="cuda"))
            odt = torch.float32
        out, ref = torch.empty(M, N, device="cuda", dtype=odt), torch.empty(M, N, device="cuda", dtype=odt)
        # MTP_AB_ROTATE=n: cycle through n output (and aux-out) buffers, as the training step does -- every GEMM there writes fresh memory,
        # while a benchmark that rewrites ONE buffer keeps its output lines resident in the 256-MB infinity cache
        nrot = int(os.environ.get("MTP_AB_ROTATE", "1"))
        outs = [out] + [torch.empty_like(out) for _ in range(nrot - 1)]
        auxs = [kw.get("aux")] + [torch.empty_like(kw["aux"]) if (epi in ("gelu", "gelu_dg")) else kw.get("aux") for _ in range(nrot - 1)]
        rot = [0]

        bias_bf = bias.to(bf)

        def launch(v):
            i = rot[0] = (rot[0] + 1) % nrot
            if v == -1:                   # the library GEMM (torch -> hipBLASLt) on the same operands: a yardstick, not a product path
                torch.addmm(bias_bf, a, w.t(), out=outs[i])
                return
            k2 = dict(kw)
            if auxs[i] is not None:
                k2["aux"] = auxs[i]
            ops.gemm_nt(a, w, outs[i], variant=v, **k2)
        ops.gemm_nt(a, w, ref, variant=1024, **kw)
        ts = {v: [] for v in variants}
        okv = {}
        iters = 10 if M > T else 20
        for v in variants:
            if v == -1:
                if epi == "bias":
                    okv[v] = True
                    time_many(lambda: launch(v), 3)
                continue
            if v == S8 and epi in ("gelu", "dgelu"):
                continue
            if v not in (S8, 1024, 256, 512, 768, 258, 256 + 32768, 512 + 32768, 768 + 32768, 512 + 65536, 768 + 65536, 512 + NT_, 512 + SC1, 512 + PLAIN) and epi != "bias":
                continue
            out.zero_()
            ops.gemm_nt(a, w, out, variant=v, **kw)
            okv[v] = torch.equal(out, ref) or ((v >> 11) & 12) != 0     # the nostore / nomfma ablations compute nothing to compare
            time_many(lambda: launch(v), 3)
        for _ in range(rounds):
            for v in okv:
                ts[v].append(time_many(lambda: launch(v), iters))
        fl = 2.0 * M * N * K
        cells = []
        for v in okv:
            cells.append("%s %.1fus %.0f/%.0fTF%s" % (NAMES.get(v, str(v)), min(ts[v]) * 1e6, fl / statistics.median(ts[v]) / 1e12, fl / min(ts[v]) / 1e12,
                                                      "" if okv[v] else " MISMATCH"))
        print("%-5s M=%d N=%d K=%d | " % (epi, M, N, K) + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
