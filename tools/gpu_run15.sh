#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -15 > gpurun_out/r3p_pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/r3p_bench_vitl.json 2>> gpurun_out/r3p_bench.err
done
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --timer-every 1 --gemm-shapes > /dev/null 2> gpurun_out/r3p_shapes.txt
cat gpurun_out/r3p_pytest.log; cut -c1-200 gpurun_out/r3p_bench_vitl.json; tail -3 gpurun_out/r3p_bench.err; head -24 gpurun_out/r3p_shapes.txt
