#!/bin/bash
# HISTORICAL (how profiles/r05_ab_late_16byte_stores.txt was taken): needs tools/_abl/libmtp_hip_prev.so = the library of commit b7f9c7c (git archive b7f9c7c mtp_amd/csrc include | tar -x -C /tmp/x && make -C /tmp/x/mtp_amd/csrc)
if [ "$MTP_RUN_HISTORICAL" != "1" ]; then echo "tools/gpu_r05_late_16byte_stores.sh: historical record of a measurement -- see its header; set MTP_RUN_HISTORICAL=1 to run it anyway" >&2; exit 1; fi
# round 5, late (2): 16-byte stores in weight_images_kernel and the bf16 transposes.  (1) their tests + the FPN / backbone tests that run them; (2) step A/B against the previous
# library, interleaved; (3) one-stream kernel statistics: ViT-L (the two kernels' own times) and InternImage-XL (what reduce_rows_batched costs by itself); (4) forced-comm line with
# RCCL's INFO log kept (comm.rccl)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05_o; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py tests/test_hip_parallel.py -m gpu -q -x --timeout 600 -k "weight_images or tokens_nchw or transpose or forced_comm_single_rank_rccl or vit_l_headline or small_model_forward or vit_b_config1 or patch_size_8" 2>&1 | tail -5 | tee $O/pytest.log
for i in 1 2 3; do
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_prev.so timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/prev_$i.json 2>> $O/prev.err
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
done
python - <<PY | tee $O/ab.txt
import json, glob
print("# same box, 20 steps each, interleaved; ms per step.  prev = library of commit b7f9c7c; new = 16-byte stores in weight_images_kernel and transpose8_bf16_kernel")
for tag in ("prev", "new"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
cd /tmp
for lib in prev new; do
  L=""; [ $lib = prev ] && L=$R/tools/_abl/libmtp_hip_prev.so
  MTP_HIP_LIB=$L timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vitl_$lib -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/prof_vitl_$lib.log 2>&1
  cp $(find $O/prof_vitl_$lib -name "t_kernel_stats.csv" | head -1) $O/vitl_${lib}_kernel_stats.csv; rm -rf $O/prof_vitl_$lib
done
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ii -o t -- python $R/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/prof_ii.log 2>&1
cp $(find $O/prof_ii -name "t_kernel_stats.csv" | head -1) $O/ii_single_stream_kernel_stats.csv; rm -rf $O/prof_ii
cd $R
MTP_RCCL_LOG_COPY=$O/rccl_rank0.log MTP_FORCE_COMM=1 timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/forced_comm.json 2> $O/forced_comm.err
tail -n 3 $O/new.err; tail -n 3 $O/forced_comm.err
grep -h "weight_images\|transpose\|reduce_rows_batched" $O/vitl_prev_kernel_stats.csv $O/vitl_new_kernel_stats.csv $O/ii_single_stream_kernel_stats.csv | cut -c1-200
