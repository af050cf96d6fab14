#!/bin/bash
# the round's profile set, one box: GPU test suite, bench line, kernel stats + last-step trace, both PMC passes, forced-comm variants, the other configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/final_r04
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_single -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/trace_single.log 2>&1
rm -f $O/trace_single/t_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_write.log 2>&1
cd $R
rm -f $O/pmc_fetch/p_kernel_trace.csv $O/pmc_write/p_kernel_trace.csv
python - <<'PY'
import csv, os
o = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/final_r04/trace/"
rows = list(csv.DictReader(open(o + "t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6
open(o + "steady_step.txt", "w").write("last steady-state step of the trace: %d launches, span %.3f ms, kernel time %.3f ms\n" % (len(step), span, busy))
with open(o + "t_kernel_trace_last_step.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size"])
    w.writeheader()
    for r in step:
        w.writerow({k: r[k] for k in w.fieldnames})
os.remove(o + "t_kernel_trace.csv")
PY
MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/bench_n1_forced_comm.json 2>> $O/bench_n1.err
MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --comm-mode rs_ag --comm-bf16 > $O/bench_n1_forced_comm_rs_ag_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config2_vitb_b32_standin3.json 2>> $O/bench_n1.err
timeout 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vitl_448_b16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --image-size 448 --batch 16 --use-ckpt --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vitl_448_b16_ckpt.json 2>> $O/bench_n1.err
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_internimage_xl_512_b8.json 2>> $O/bench_n1.err
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --heads standin_seg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_internimage_xl_512_b8_standin_seg.json 2>> $O/bench_n1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_intern -o t -- python $R/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/trace_intern.log 2>&1
rm -f $O/trace_intern/t_kernel_trace.csv
cd $R
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-forward-only --timer-every 20 > $O/bench_n1_200steps.json 2>> $O/bench_n1.err
for i in 1 2 3; do
  (cd $R/_base && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_base_$i.json 2>> $O/ab.err)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/ab_new_$i.json 2>> $O/ab.err
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream 0 > $O/ab_single_$i.json 2>> $O/ab.err
done
python - <<PY > $O/ab_step_vs_round3.txt
import json, glob
print("# same box, 20 steps each: the round-3 tree (git b249d9b, its own libmtp_hip.so) interleaved with this tree; ab_single = this tree with the weight-gradient bursts on the compute stream; ms per step")
for tag in ("ab_base", "ab_new", "ab_single"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
cat $O/pytest_gpu.log; cut -c1-220 $O/bench_n1.json $O/bench_n1_forced_comm.json $O/bench_internimage_xl_512_b8.json $O/bench_vitl_448_b16.json $O/bench_vitl_448_b16_ckpt.json $O/bench_config2_vitb_b32_standin3.json $O/bench_n1_200steps.json; cat $O/trace/steady_step.txt $O/ab_step_vs_round3.txt; tail -3 $O/bench_n1.err; du -sh $O
