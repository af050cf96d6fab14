#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_dcnv3.py -m gpu -q -x --timeout 600 2>&1 | tail -2 > gpurun_out/r3aa.txt
for i in 1 2; do
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('intern', d['value'], d['ms_per_step'])" >> gpurun_out/r3aa.txt
done
timeout 300 python tools/bench_ops.py dcnv3 > gpurun_out/r3aa_dcnv3_ops.txt 2>&1
cat gpurun_out/r3aa.txt; grep "N=8" gpurun_out/r3aa_dcnv3_ops.txt | grep bfloat16 | grep "zero\|bench"
