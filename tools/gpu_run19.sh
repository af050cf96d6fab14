#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_internimage.py tests/test_hip_ops.py -m gpu -q -x --timeout 600 -k "internimage or gemm" 2>&1 | tail -4 > gpurun_out/r3t.txt
for i in 1 2; do
MTP_DEFER_LN_REDUCE=0 timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('intern nodefer', d['value'], d['ms_per_step'])" >> gpurun_out/r3t.txt
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('intern defer', d['value'], d['ms_per_step'])" >> gpurun_out/r3t.txt
done
timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('vitb', d['value'], d['ms_per_step'])" >> gpurun_out/r3t.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('vitl', d['value'], d['ms_per_step'])" >> gpurun_out/r3t.txt
cat gpurun_out/r3t.txt
