#!/bin/bash
# after tools/gpu_final_r06.sh: copy the summaries that are judged from gpurun_out/final_r06/ into profiles/ (tracked)
cd "$(dirname "$0")/.."
O=gpurun_out/final_r06
cp $O/r06_hbm_fractions.json $O/r06_hbm_fractions.txt $O/r06_pmc_hbm.json $O/r06_rocprofv3_kernel_stats_single_stream.csv profiles/
cp $O/kernel_stats.csv profiles/r06_rocprofv3_kernel_stats.csv
cp $O/kernel_trace_last_step.csv profiles/r06_kernel_trace_last_step.csv
cp $O/steady_step.txt profiles/r06_steady_step.txt
cp $O/parity_errors.json profiles/r06_parity_errors.json
cp $O/pytest_gpu.log profiles/r06_gputest_summary.txt
cp $O/vitl_gemm_shapes.txt profiles/r06_vitl_gemm_shapes.txt
cp $O/internimage_xl_kernel_stats_single_stream.csv profiles/r06_internimage_xl_kernel_stats_single_stream.csv
cp $O/vitb_b32_standin3_kernel_stats_single_stream.csv profiles/r06_vitb_b32_standin3_kernel_stats_single_stream.csv
for f in bench_n1 bench_n1_200steps bench_n1_forced_comm bench_n1_forced_comm_rs_ag_bf16 bench_config2_vitb_b32_standin3 bench_vitl_448_b16 bench_vitl_448_b16_ckpt bench_internimage_xl_512_b8 bench_internimage_xl_512_b8_with_cp bench_internimage_xl_512_b8_standin_seg; do cp $O/$f.json profiles/r06_$f.json; done
cp $O/ab_vs_round5.txt profiles/r06_ab_vs_round5_all_configs.txt
# (the GPU box has no .git: the commit the set was taken at is filled in here)
python - <<PY
import json, subprocess
p = "profiles/r06_pmc_hbm.json"
d = json.load(open(p))
d["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
json.dump(d, open(p, "w"), indent=1)
print("csrc", d["_csrc_sha"], "commit", d["_commit"])
PY
