"""A/B of the NT kernel families on mid-size problems (InternImage-XL levels 2 / 3: 48-128 tiles of 256 x 256 for 256 CUs): the 128-wide kernels of
gemm.hip (variant 1024 = never the 8-phase kernel) against the 8-phase kernel forced (256 = tile height picked, 512 = 224 rows, 768 = 256 rows).
usage: python tools/probes/ab_gemm_mid.py [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mtp_amd import ops
from tools.ab_gemm import time_many
from tools.bench_ops import r


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    bf = torch.bfloat16
    shapes = [(8192, 768, 768), (8192, 768, 3072), (8192, 3072, 768), (8192, 864, 768), (8192, 432, 768), (2048, 1536, 1536), (2048, 1536, 6144), (2048, 6144, 1536),
              (32768, 384, 384), (32768, 384, 1536), (32768, 1536, 384), (131072, 192, 768), (6272, 768, 768), (6272, 2304, 768), (6272, 3072, 768), (6272, 768, 3072)]
    for (M, N, K) in shapes:
        a, w = r(M, K), r(N, K, scale=0.02)
        bias = torch.randn(N, device="cuda")
        outs = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(4)]
        ref = torch.empty(M, N, device="cuda", dtype=bf)
        ops.gemm_nt(a, w, ref, bias=bias, variant=1024)
        ts, ok = {}, {}
        C2 = 1 << 22      # the co-resident 4-wave form (tools/ablation/gemm_c2.hip): only in an ablation build (MTP_HIP_LIB=tools/_abl/libmtp_hip_c2.so)
        for v in (0, 1024, 256, 512, 768) + ((C2,) if os.environ.get("MTP_AB_C2") else ()):
            try:
                ops.gemm_nt(a, w, outs[0], bias=bias, variant=v)
            except Exception as e:
                ok[v] = "n/a"
                continue
            ok[v] = "" if torch.equal(outs[0], ref) else " MISMATCH"
            ts[v] = []
            i = [0]

            def go():
                i[0] = (i[0] + 1) % 4
                ops.gemm_nt(a, w, outs[i[0]], bias=bias, variant=v)
            time_many(go, 3)
        for _ in range(rounds):
            for v in ts:
                ts[v].append(time_many(lambda: (ops.gemm_nt(a, w, outs[0], bias=bias, variant=v)), 20))
        fl = 2.0 * M * N * K
        names = {0: "default", 1024: "128-wide", 256: "p8-auto", 512: "p8-224", 768: "p8-256", 1 << 22: "c2-256x128"}
        print("M=%d N=%d K=%d tiles256=%d | " % (M, N, K, -(-M // 256) * -(-N // 256)) +
              " | ".join("%s %.1fus %.0fTF%s" % (names[v], min(ts[v]) * 1e6, fl / statistics.median(ts[v]) / 1e12, ok[v]) for v in ts), flush=True)


if __name__ == "__main__":
    main()
