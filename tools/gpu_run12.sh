#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --timer-every 10" "--steps 20 --warmup 5 --no-gemm-timer" "--steps 20 --warmup 50 --no-gemm-timer" "--steps 20 --warmup 5" "--steps 100 --warmup 5 --timer-every 50"; do
  echo "== $cfg" >> gpurun_out/r3l_bench_variants.txt
  timeout 300 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families'] if d['roofline'] else None, d.get('clocks'))" >> gpurun_out/r3l_bench_variants.txt
done
cat gpurun_out/r3l_bench_variants.txt
