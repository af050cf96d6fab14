#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_backbone.py tests/test_hip_engine.py -m gpu -q -x --timeout 900 2>&1 | tail -5 > gpurun_out/r3q_pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/r3q_bench_vitl.json 2>> gpurun_out/r3q_bench.err
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r3q -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > $GRAFT_REPO_ROOT/gpurun_out/r3q_trace.log 2>&1
cp /tmp/r3q/t_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/r3q_kernel_trace.csv
cd $GRAFT_REPO_ROOT
cat gpurun_out/r3q_pytest.log; cut -c1-200 gpurun_out/r3q_bench_vitl.json; tail -3 gpurun_out/r3q_bench.err
