#!/bin/bash
# The round's FINAL profile set in one gpurun call, at the last kernel sources (everything lands in gpurun_out/final_r06/; the summaries that are judged get copied to profiles/):
#   GPU test suite + parity errors; one-stream kernel statistics + both PMC passes -> pmc_hbm.json + the HBM table (written BEFORE the bench line so that the line quotes
#   them); the driver-command bench line; as-run kernel statistics + last-step trace; forced-comm lines (C-ABI communicator; rs_ag + bf16); the other configurations;
#   InternImage-XL kernel statistics; the 200-step line; same-box pairs against round 5's tree (_base/ = git e465476 with its own library) for all three configurations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/final_r06
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_single -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/trace_single.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_write.log 2>&1
cd $R
FETCH=$(find $O/pmc_fetch -name "p_counter_collection.csv" | head -1); WRITE=$(find $O/pmc_write -name "p_counter_collection.csv" | head -1)
python tools/pmc_hbm.py $FETCH $WRITE > profiles/r06_pmc_hbm.json 2> $O/pmc_hbm.err
cp $(find $O/trace_single -name "t_kernel_stats.csv" | head -1) profiles/r06_rocprofv3_kernel_stats_single_stream.csv
python tools/hbm_fractions.py profiles/r06_rocprofv3_kernel_stats_single_stream.csv profiles/r06_pmc_hbm.json --json profiles/r06_hbm_fractions.json > profiles/r06_hbm_fractions.txt 2>> $O/pmc_hbm.err
cp profiles/r06_pmc_hbm.json profiles/r06_rocprofv3_kernel_stats_single_stream.csv profiles/r06_hbm_fractions.json profiles/r06_hbm_fractions.txt $O/
rm -rf $O/trace_single $O/pmc_fetch $O/pmc_write
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --gemm-shapes > /dev/null 2> $O/vitl_gemm_shapes.txt
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os
o = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/final_r06/"
f = glob.glob(o + "trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6
open(o + "steady_step.txt", "w").write("last steady-state step of the trace: %d launches, span %.3f ms, kernel time %.3f ms\n" % (len(step), span, busy))
with open(o + "kernel_trace_last_step.csv", "w") as g:
    w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size"])
    w.writeheader()
    for r in step:
        w.writerow({k: r[k] for k in w.fieldnames})
os.remove(f)
PY
cp $(find $O/trace -name "t_kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/trace
MTP_RCCL_LOG_COPY=$O/rccl_rank0.log MTP_FORCE_COMM=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/bench_n1_forced_comm.json 2>> $O/bench_n1.err
MTP_FORCE_COMM=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --comm-mode rs_ag --comm-bf16 > $O/bench_n1_forced_comm_rs_ag_bf16.json 2>> $O/bench_n1.err
rm -f $O/rccl_rank0.log
timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config2_vitb_b32_standin3.json 2>> $O/bench_n1.err
timeout -s KILL 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vitl_448_b16.json 2>> $O/bench_n1.err
timeout -s KILL 300 python bench.py --image-size 448 --batch 16 --use-ckpt --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vitl_448_b16_ckpt.json 2>> $O/bench_n1.err
timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_internimage_xl_512_b8.json 2>> $O/bench_n1.err
timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --use-ckpt --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/bench_internimage_xl_512_b8_with_cp.json 2>> $O/bench_n1.err
timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --heads standin_seg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_internimage_xl_512_b8_standin_seg.json 2>> $O/bench_n1.err
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_intern1 -o t -- python $R/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/trace_intern1.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_vitb -o t -- python $R/bench.py --model vit_b --batch 32 --heads standin3 --steps 6 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/trace_vitb.log 2>&1
cd $R
cp $(find $O/trace_intern1 -name "t_kernel_stats.csv" | head -1) $O/internimage_xl_kernel_stats_single_stream.csv
cp $(find $O/trace_vitb -name "t_kernel_stats.csv" | head -1) $O/vitb_b32_standin3_kernel_stats_single_stream.csv
rm -rf $O/trace_intern1 $O/trace_vitb
timeout -s KILL 400 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-forward-only > $O/bench_n1_200steps.json 2>> $O/bench_n1.err
for i in 1 2 3; do
  (cd $R/_base && timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/ab_base_$i.json 2>> $O/ab.err)
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/ab_new_$i.json 2>> $O/ab.err
done
for i in 1 2; do
  (cd $R/_base && timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/vitb_base_$i.json 2>> $O/ab.err)
  timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/vitb_new_$i.json 2>> $O/ab.err
  (cd $R/_base && timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/ii_base_$i.json 2>> $O/ab.err)
  timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/ii_new_$i.json 2>> $O/ab.err
done
python - <<PY > $O/ab_vs_round5.txt
import json, glob
print("# same box, interleaved: the round-5 tree (git e465476, its own libmtp_hip.so, _base/) against this tree.  (ms per step, images per second, NT family TF/s, step_mfma_frac)")
for tag in ("ab_base", "ab_new", "vitb_base", "vitb_new", "ii_base", "ii_new"):
    rows = []
    for f in sorted(glob.glob("$O/%s_*.json" % tag)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            rows.append(("unreadable", str(e)[:40])); continue
        fam = (d.get("roofline") or {}).get("families", {})
        rows.append((d["ms_per_step"], d["value"], fam.get("gemm_nt", {}).get("tflops"), d.get("step_mfma_frac")))
    good = [r[0] for r in rows if isinstance(r[0], float)]
    print(tag, rows, ("min %.3f ms" % min(good)) if good else "")
PY
cat $O/pytest_gpu.log; cut -c1-200 $O/bench_n1.json $O/bench_n1_forced_comm.json $O/bench_internimage_xl_512_b8.json $O/bench_vitl_448_b16.json $O/bench_config2_vitb_b32_standin3.json $O/bench_n1_200steps.json; cat $O/steady_step.txt $O/ab_vs_round5.txt; head -24 $O/hbm_fractions.txt | cut -c1-150
