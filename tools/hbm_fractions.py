"""GB/s against the 8 TB/s HBM peak for every HBM-bound kernel of the headline step (ViT-L + RVSA, B = 64, 224^2, bf16).

Joins, per kernel family,
  * the rocprofv3 kernel statistics of a SINGLE-STREAM trace (average duration, launches),
  * the PMC passes (tools/pmc_hbm.py: FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction) -- bytes that crossed the L2 <-> fabric boundary,
  * the ALGORITHMIC bytes of one launch (every operand and result once), written out below,
into one table: us per launch, algorithmic MB, counter MB, counter / algorithmic, TB/s on the algorithmic bytes, fraction of 8 TB/s.

    python tools/hbm_fractions.py profiles/r05_rocprofv3_kernel_stats_single_stream.csv profiles/r05_pmc_hbm.json > profiles/r05_hbm_fractions.txt
(also writes <out>.json next to the PMC file when --json PATH is given; bench.py attaches it to the `roofline` object when its csrc hash matches.)"""
import csv
import json
import sys

T, C, H, B = 12544, 1024, 16, 64          # tokens, channels, heads, images
NWIN = B * 4                               # 7 x 7 windows per launch (x heads = workgroups)
PARAMS = 317_628_800 - 2 * C               # trainable parameters in the flat buffers (norm.* excluded)
PEAK = 8.0e12

# family -> (needle in the kernel name, statistics key in the PMC json, algorithmic bytes per launch, what they are)
FAMILIES = [
    ("ln_fwd (norm1 / norm2: f32 x in, bf16 y out, 2 f32 stats per row)", "ln_fwd_kernel<float", "ln_fwd_kernel", T * C * (4 + 2) + T * 8, "x f32 + y bf16 + mean / rstd"),
    ("ln_bwd (dy bf16 + x f32 + residual-gradient f32 in; dx f32 + dx bf16 copy out; partial d-gamma / d-beta rows)", "ln_bwd_kernel", "ln_bwd_kernel", T * C * (2 + 4 + 4 + 4 + 2), "dy + x + dres + dx + dx_act"),
    ("rvsa_bwd4 (qkv rows gathered, o, do, lse in; dq rows + dK_sel / dV_sel rows + table partials out)", "rvsa_bwd", "rvsa_bwd4", T * 3 * C * 2 + 2 * T * C * 2 + T * C * 2 + 2 * NWIN * H * 49 * 64 * 2 + NWIN * H * (26 * 64 + 169) * 4, "qkv + o + do + dq + dKs|dVs + partials"),
    ("rvsa_scatter_gemm (dK_sel / dV_sel rows in, dk / dv token rows out)", "rvsa_scatter_gemm_kernel", "rvsa_scatter_gemm", 2 * NWIN * H * 49 * 64 * 2 + 2 * T * C * 2, "dKs|dVs + dk|dv"),
    ("rvsa_fwd4 (qkv rows in, o + lse out)", "rvsa_fwd4_mfma_kernel", "rvsa_fwd4", T * 3 * C * 2 + T * C * 2 + NWIN * H * 49 * 4, "qkv + o + lse"),
    ("adamw (p, g, m, v in; p, m, v out: 28 B per parameter)", "adamw_kernel", "adamw", 28 * PARAMS, "28 B x parameters"),
    ("adamw_images (round 6: AdamW + both bf16 images of every GEMM weight in one launch)", "adamw_images_kernel", "adamw", 28 * PARAMS + 4 * 303_000_000, "28 B x parameters + 4 B x GEMM weights"),
    ("weight_images (f32 masters in, bf16 W and W^T images out)", "weight_images_kernel", "weight_images", 303_000_000 * (4 + 2 + 2), "GEMM weights x (4 + 2 + 2) B"),
    ("sqnorm (gradient norm: one read of the flat gradient)", "sqnorm_kernel", "sqnorm", 4 * PARAMS, "4 B x parameters"),
    ("v3_fwd (full attention, <= 16 x 16 grids: qkv in, o + lse out)", "v3_fwd_kernel", "full_v3_fwd", T * 3 * C * 2 + T * C * 2 + B * H * 196 * 4, "qkv + o + lse"),
    ("v3_bwd_a", "v3_bwd_a_kernel", "full_v3_bwd_a", T * 3 * C * 2 + 2 * T * C * 2 + T * C * 2, "qkv + o + do + dq"),
    ("v3_bwd_b", "v3_bwd_b_kernel", "full_v3_bwd_b", T * 3 * C * 2 + T * C * 2 + 2 * T * C * 2, "qkv + do + dk|dv"),
    ("rvsa_sampling_fwd (x rows in: pooled grid + heads)", "rvsa_sampling_fwd_kernel", "rvsa_sampling_fwd", T * C * 2, "x bf16"),
    ("transpose (token <-> NCHW layout changes of the FPN tail, per launch average)", "transpose", "transpose", None, "--"),
]
MFMA = [("gemm_nt (NT family: forward + data gradients)", "gemm_nt_p8_kernel", "gemm_nt_kernel"), ("gemm_tn_p8 (grouped weight gradients)", "gemm_tn_p8_kernel", "gemm_tn_kernel")]


def gemm_algorithmic_bytes_per_step():
    """every operand and result of every GEMM launch of one step once (bf16 activations 2 B, f32 residual stream / weight gradients 4 B, bf16 weight images):
    returns (NT bytes, NT launches, TN bytes) per step.  VERDICT r05 #5a: the traffic table printed '--' for the GEMM families."""
    a, w = T * C * 2, C * C * 2                                   # one (T, C) bf16 activation, one (C, C) bf16 weight image
    x = T * C * 4                                                 # one (T, C) f32 residual-stream tensor
    blk_nt = [a + 3 * w + 3 * a,                                  # qkv:      ln1 | W | qkv
              a + w + x + x,                                      # proj:     o | W | residual in | x1 out (f32)
              a + 4 * w + 4 * a + 4 * a,                          # fc1:      ln2 | W | h | gelu' image u
              4 * a + 4 * w + x + x,                              # fc2:      h | W | residual in | x2 out
              a + 4 * w + 4 * a + 4 * a,                          # dgrad fc2: dx2 | W^T | u | du
              4 * a + 4 * w + a,                                  # dgrad fc1: du | W^T | dln2
              a + w + a,                                          # dgrad proj
              3 * a + 3 * w + a]                                  # dgrad qkv
    fpn_nt = [a + 4 * w + 4 * a, 4 * a + 4 * w + 16 * a, a + 4 * w + 4 * a,            # fpn1.0, fpn1.3, fpn2.0 forward (ConvT as GEMM, bf16 out)
              16 * a + 4 * w + 4 * a, 4 * a + 4 * w + x, 4 * a + 4 * w + x,            # their data gradients (the two that reach the residual stream: f32 out)
              T * 768 * 2 + 768 * C * 2 + x]                                           # patch embedding forward
    nt = 24 * sum(blk_nt) + sum(fpn_nt)
    dw = C * C * 4
    blk_tn = [a + 4 * a + 4 * dw, 4 * a + a + 4 * dw, a + a + dw, 3 * a + a + 3 * dw]  # fc2, fc1, proj, qkv weight gradients: dY | X | dW (f32)
    fpn_tn = [16 * a + 4 * a + 4 * dw, 4 * a + a + 4 * dw, 4 * a + a + 4 * dw, a + T * 768 * 2 + 768 * C * 4]
    return nt, 24 * 8 + 7, 24 * sum(blk_tn) + sum(fpn_tn)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    jpath = None
    if "--json" in sys.argv:
        jpath = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != jpath]
    stats, pmc = args[0], json.load(open(args[1]))
    rows = list(csv.DictReader(open(stats)))

    def agg(needle):
        n, tot = 0, 0.0
        for r in rows:
            if needle in r["Name"]:
                n += int(r["Calls"])
                tot += float(r["TotalDurationNs"])
        return n, tot
    print("# HBM-bound kernels of the headline step (ViT-L + RVSA, B = 64, 224^2, bf16, 1x MI355X): rate against the 8 TB/s HBM3E peak (SURVEY 8d)")
    print("# durations: %s (single-stream trace, all traced steps); counters: %s (csrc %s, commit %s)" % (stats, args[1], pmc.get("_csrc_sha"), pmc.get("_commit")))
    print("# algorithmic = every operand and result of one launch once (DESIGN.md section 3); counter = (2 x FETCH_SIZE + WRITE_SIZE) KB of the PMC passes = bytes across")
    print("# the L2 <-> fabric boundary (Infinity-Cache hits included: an upper bound on HBM bytes); TB/s and the fraction are on the ALGORITHMIC bytes")
    print("%-46s %8s %9s %10s %10s %8s %7s %6s" % ("kernel", "launches", "us/launch", "algo MB", "counter MB", "ctr/algo", "TB/s", "frac"))
    out = {"_csrc_sha": pmc.get("_csrc_sha"), "_commit": pmc.get("_commit"), "_stats": stats, "_pmc": args[1], "peak_TBps": PEAK / 1e12, "kernels": {}}
    for label, needle, key, algo, what in FAMILIES:
        n, tot = agg(needle)
        if not n:
            continue
        us = tot / n / 1e3
        ctr = pmc.get(key, {}).get("hbm_bytes_per_launch") if key else None
        if algo is None:
            print("%-46s %8d %9.1f %10s %10s %8s %7s %6s   %s" % (label.split(" (")[0], n, us, "--", "%.1f" % (ctr / 1e6) if ctr else "--", "--", "--", "--", label))
            continue
        tbps = algo / (us * 1e-6) / 1e12
        print("%-46s %8d %9.1f %10.1f %10s %8s %7.2f %6.2f   %s" % (label.split(" (")[0], n, us, algo / 1e6, "%.1f" % (ctr / 1e6) if ctr else "--",
                                                                 "%.2f" % (ctr / algo) if ctr else "--", tbps, tbps / (PEAK / 1e12), what))
        out["kernels"][label.split(" (")[0]] = dict(launches=n, us_per_launch=round(us, 1), algorithmic_bytes=int(algo), counter_bytes=ctr, TBps=round(tbps, 2), frac=round(tbps / (PEAK / 1e12), 3))
    print("# MFMA-bound families, for the traffic column only (their roofline is the matrix pipe: bench.py `roofline`).  algo MB = the step's GEMM operands and results once,")
    print("# divided by the launches of the family in the PMC pass (the grouped TN launch count depends on the burst size: per-step totals are in the json)")
    nt_b, nt_n, tn_b = gemm_algorithmic_bytes_per_step()
    for (label, needle, key), per_step in zip(MFMA, (nt_b, tn_b)):
        n, tot = agg(needle)
        if n:
            e = pmc.get(key, {})
            ctr, nl = e.get("hbm_bytes_per_launch"), e.get("launches")
            per_launch = per_step / (nl / 2.0) if nl else None          # the PMC passes hold two steps (--steps 1 --warmup 1)
            print("%-46s %8d %9.1f %10s %10s %8s" % (label.split(" (")[0], n, tot / n / 1e3, "%.1f" % (per_launch / 1e6) if per_launch else "--",
                                                      "%.1f" % (ctr / 1e6) if ctr else "--", "%.2f" % (ctr / per_launch) if ctr and per_launch else "--"))
            out["kernels"][label.split(" (")[0] + " [mfma-bound]"] = dict(launches=n, us_per_launch=round(tot / n / 1e3, 1), algorithmic_bytes=int(per_launch) if per_launch else None,
                                                                        algorithmic_bytes_per_step=int(per_step), counter_bytes=ctr, TBps=None, frac=None)
    print("# NT ratio: one XCD's 32 CUs hold 32 tiles, whose operand panels (0.5 MiB each at K = 1024) exceed its 4-MiB L2 for any tile order, so each of the 8 L2s fetches")
    print("# the panels of its own tile range: the re-fetches are L2 misses served by the 256-MiB Infinity Cache (the operands of a launch total 32 MB), counted by FETCH_SIZE;")
    print("# the HBM side of a launch stays at its algorithmic bytes (DESIGN.md section 4, 'XCD-aware tile order').")
    if jpath:
        json.dump(out, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
