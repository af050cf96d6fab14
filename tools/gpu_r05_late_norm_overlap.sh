#!/bin/bash
# HISTORICAL (how profiles/r05_ab_late_weight_images_norm_overlap.txt was taken): needs tools/ablation/grad_norm_overlap.patch.txt applied (MTP_NORM_OVERLAP) and tools/_abl/libmtp_hip_prev.so = the library of commit b7f9c7c
if [ "$MTP_RUN_HISTORICAL" != "1" ]; then echo "tools/gpu_r05_late_norm_overlap.sh: historical record of a measurement -- see its header; set MTP_RUN_HISTORICAL=1 to run it anyway" >&2; exit 1; fi
# round 5, late: (1) tests of the 16-byte weight-image stores and the overlapped gradient norm; (2) step A/B, interleaved: previous library + single-pass norm / new library +
# single-pass norm / new library + norm in pieces; (3) InternImage-XL kernel statistics on ONE stream (what reduce_rows_batched costs by itself); (4) forced-comm line (comm.rccl)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05_n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_hip_ops.py tests/test_hip_parallel.py -m gpu -q -x --timeout 600 -k "weight_images or norm or forced_comm" 2>&1 | tail -5 | tee $O/pytest.log
for i in 1 2 3; do
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_prev.so MTP_NORM_OVERLAP=0 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/prev_$i.json 2>> $O/prev.err
  MTP_NORM_OVERLAP=0 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/newlib_$i.json 2>> $O/newlib.err
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/overlap_$i.json 2>> $O/overlap.err
done
python - <<PY | tee $O/ab.txt
import json, glob
print("# same box, 20 steps each, interleaved; ms per step.  prev = library of commit b7f9c7c + single-pass norm; newlib = 16-byte weight-image stores + single-pass norm; overlap = newlib + gradient norm in pieces next to the backward (default)")
for tag in ("prev", "newlib", "overlap"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/prof_vitl -o vitl -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-forward-only --wgrad-side-stream 0 > $O/prof_vitl.log 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/prof_ii -o ii -- python $R/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-forward-only --wgrad-side-stream 0 > $O/prof_ii.log 2>&1
cd $R
for d in prof_vitl prof_ii; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; find $O/$d -name "*.csv" ! -name "*stats*" -delete; find $O/$d -name "*.db" -delete; done
MTP_FORCE_COMM=1 timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/forced_comm.json 2> $O/forced_comm.err
tail -3 $O/overlap.err $O/forced_comm.err
grep -h "weight_images\|reduce_rows_batched\|sqnorm" $O/prof_vitl_kernel_stats.csv $O/prof_ii_kernel_stats.csv | cut -c1-200
