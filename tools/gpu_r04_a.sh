#!/bin/bash
# round 4, call A: the co-resident NT kernel (gemm_c2.hip) -- parity, per-shape A/B, whole-step A/B; InternImage bf16 diagnostics
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
C2=$((256 + (1 << 22)))
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "p8 or gemm_nt" --timeout 600 2>&1 | tail -8 > $O/pytest_c2.log
cat $O/pytest_c2.log
MTP_AB_ROTATE=8 timeout 420 python tools/ab_gemm.py 3 512 $C2 $((C2 + 2)) > $O/ab_gemm_c2.txt 2>&1
cat $O/ab_gemm_c2.txt
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
MTP_NT_VARIANT=$((1 << 22)) timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_default2.json 2>> $O/bench_default.err
cut -c1-260 $O/bench_default.json $O/bench_c2.json $O/bench_default2.json; tail -3 $O/bench_c2.err
# InternImage bf16: is the 0.29 of dw_conv.1.1.weight the gather-form backward or noise on a 2 x 2 map?
for v in 0 2; do
  MTP_DCNV3_VARIANT=$v timeout 300 python -m pytest tests/test_hip_internimage.py -q -k "every_gradient and bf16" --timeout 300 2>&1 | tail -2
  python - <<PY
import json
d = json.load(open("gpurun_out/parity_errors.json"))["internimage_small_bf16"]
top = sorted(d.items(), key=lambda kv: -kv[1])[:5]
print("variant $v:", [(k, round(x, 3)) for k, x in top])
PY
done 2>&1 | tee $O/intern_bf16_variants.txt
