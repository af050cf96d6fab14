"""HBM traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer
    python tools/pmc_hbm.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv > profiles/r01_pmc_hbm.json

Both counters are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts 64 B per
128-B request, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The AdamW kernel is the built-in check: its algorithmic traffic
is 28 B / parameter.  Keys `gemm_nt_kernel` / `gemm_tn_kernel` are what bench.py reads for `roofline.traffic`."""
import collections
import csv
import json
import os
import sys

FAMILIES = [("gemm_nt", "gemm_nt_kernel"), ("gemm_tn_p8", "gemm_tn_kernel"), ("gemm_tn", "gemm_tn_split_kernel"), ("sum_partials", "sum_partials"), ("ln_fwd", "ln_fwd_kernel"),
            ("ln_bwd", "ln_bwd_kernel"), ("rvsa_bwd4", "rvsa_bwd4"), ("rvsa_bwd5", "rvsa_bwd4"), ("rvsa_fwd4", "rvsa_fwd4"), ("full_bwd_a", "full_bwd_a"),
            ("full_bwd_b", "full_bwd_b"), ("full_fwd", "full_fwd"), ("v3_bwd_a", "full_v3_bwd_a"), ("v3_bwd_b", "full_v3_bwd_b"), ("v3_fwd", "full_v3_fwd"), ("adamw", "adamw"), ("weight_images", "weight_images"),
            ("reduce_rows", "reduce_rows"), ("colsum", "colsum"), ("dkv_convert", "dkv_convert"), ("small_linear", "small_linear"),
            ("rvsa_scatter_gemm", "rvsa_scatter_gemm"), ("sqnorm_segments", "sqnorm_segments"), ("sqnorm", "sqnorm"), ("rvsa_sampling_fwd", "rvsa_sampling_fwd"), ("rvsa_sampling_bwd", "rvsa_sampling_bwd"),
            ("transpose_kernel", "transpose"), ("transpose8_bf16_kernel", "transpose")]


def csrc_sha():
    """hash of the kernel sources (mtp_amd/csrc/*.hip, *.h): bench.py quotes a PMC file only when this matches the running tree"""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "mtp_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mtp_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def collect(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for needle, key in FAMILIES:
            if needle in name:
                tot[key] += float(r["Counter_Value"])
                n[key].add(r["Dispatch_Id"])
                break
    return {k: (tot[k] / len(n[k]), len(n[k])) for k in tot}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    import subprocess
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).decode().strip()
    except Exception:
        commit = os.environ.get("MTP_COMMIT", "?")
    out = {"_commit": commit, "_csrc_sha": csrc_sha(), "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `bench.py --steps 1 --warmup 1` (ViT-L, B=64, bf16, "
                    "1x MI355X); per-launch averages in KB as reported; hbm_bytes_per_launch applies the gfx950 correction of "
                    "MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request: x2) : (2*FETCH + WRITE) * 1024  [tools/pmc_hbm.py]"}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {"launches": max(nf, nw), "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
