#!/bin/bash
# round 5, call A: the strip GEMM kernel -- parity tests, micro-benchmark against the 8-wave kernel, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "s8" --timeout 300 2>&1 | tail -15 > $O/pytest_s8.log
cat $O/pytest_s8.log
MTP_AB_ROTATE=8 timeout -s KILL 600 python tools/ab_gemm.py 5 256 131072 > $O/ab_gemm_s8.txt 2>&1
cat $O/ab_gemm_s8.txt
for i in 1 2; do
  MTP_NT_S8=0 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/step_p8_$i.json 2>> $O/err.log
  MTP_NT_S8=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/step_s8_$i.json 2>> $O/err.log
done
MTP_NT_S8=1 timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --gemm-shapes > $O/shapes_s8.txt 2>> $O/err.log
MTP_NT_S8=0 timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --gemm-shapes > $O/shapes_p8.txt 2>> $O/err.log
for f in $O/step_*.json; do echo $f; cut -c1-200 $f; done
tail -5 $O/err.log
