"""Fit t = a * GFLOP + b * MB + c to the NT-GEMM dispatches of a rocprofv3 kernel trace.

    python tools/probes/gemm_time_model.py profiles/r01_gemm_nt_trace.csv

Input: rows (kernel, workgroups, duration_us) of every gemm_nt dispatch of `bench.py --steps 4 --warmup 2` (ViT-L, B = 64,
224^2, bf16), extracted from the kernel trace.  The trace does not carry M, N, K: each (kernel template, grid) group is split
into its known shapes (the engine's launch list below; equal counts per step) by sorting the durations -- a group's launches of
one shape are its k-th quantile band.  MB = bytes the epilogue moves (outputs written + aux / residual read); operand reads are
part of the main loop.
"""
import collections
import csv
import sys

import numpy as np

T, C = 12544, 1024
GF = lambda m, n, k: 2.0 * m * n * k / 1e9
# (kernel template, workgroups) -> shapes in ascending expected duration: (label, GFLOP, epilogue MB), equal counts
GROUPS = {
    ("gemm_nt_sb_kernel<bf16_t, bf16_t, 1>", 3136): [("fc1 + GELU (u and h out)", GF(T, 4 * C, C), 2 * T * 4 * C * 2 / 1e6)],
    ("gemm_nt_sb_kernel<bf16_t, bf16_t, 3>", 3136): [("dH = dY W2 * GELU'(u) (u in, dU out)", GF(T, 4 * C, C), 2 * T * 4 * C * 2 / 1e6)],
    ("gemm_nt_sb_kernel<bf16_t, bf16_t, 0>", 2352): [("qkv", GF(T, 3 * C, C), T * 3 * C * 2 / 1e6)],
    ("gemm_nt_sb8_kernel<bf16_t, float, 2>", 392): [("proj + residual (f32 in / out)", GF(T, C, C), 2 * T * C * 4 / 1e6),
                                                     ("fc2 + residual (f32 in / out)", GF(T, C, 4 * C), 2 * T * C * 4 / 1e6)],
    ("gemm_nt_sb8_kernel<bf16_t, bf16_t, 0>", 392): [("dX, K = 1024 (proj)", GF(T, C, C), T * C * 2 / 1e6), ("dX, K = 3072 (qkv)", GF(T, C, 3 * C), T * C * 2 / 1e6),
                                                      ("dX, K = 4096 (fc1)", GF(T, C, 4 * C), T * C * 2 / 1e6)],
    ("gemm_nt_sb_kernel<bf16_t, bf16_t, 0>", 12544): [("FPN ConvT on 4T rows", GF(4 * T, 4 * C, C), 4 * T * 4 * C * 2 / 1e6)],
}


def main(path):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        by[(r["kernel"], int(r["workgroups"]))].append(float(r["duration_us"]))
    X, y, rows = [], [], []
    for key, shapes in GROUPS.items():
        d = sorted(by.get(key, []))
        if not d:
            continue
        n = len(d) // len(shapes)
        for i, (label, gf, mb) in enumerate(shapes):
            band = d[i * n:(i + 1) * n]
            med = float(np.median(band))
            X.append([gf, mb, 1.0])
            y.append(med)
            rows.append((label, key[0].split("<")[0], key[1], len(band), gf, mb, med))
    X, y = np.array(X), np.array(y)
    (a, b, c), *_ = np.linalg.lstsq(X, y, rcond=None)
    print("fit over %d shape groups:  t[us] = %.4f * GFLOP + %.4f * MB + %.1f" % (len(y), a, b, c))
    print("  main loop  %.0f TFLOP/s   epilogue traffic  %.2f TB/s   fixed  %.1f us" % (1e3 / a, 1.0 / b, c))
    print("%-40s %-22s %6s %5s %8s %8s %9s %9s %7s" % ("shape", "kernel", "WGs", "n", "GFLOP", "epi MB", "median us", "model us", "err"))
    for (label, k, wg, n, gf, mb, med), pred in zip(rows, X @ np.array([a, b, c])):
        print("%-40s %-22s %6d %5d %8.1f %8.1f %9.1f %9.1f %6.1f%%" % (label, k, wg, n, gf, mb, med, pred, 100 * (pred - med) / med))
    tot = sum(r[3] * r[6] for r in rows)
    epi = sum(r[3] * r[5] * b for r in rows)
    print("epilogue term = %.1f %% of the fitted NT time of these groups" % (100 * epi / tot))


if __name__ == "__main__":
    main(sys.argv[1])
