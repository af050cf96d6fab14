#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -k "rvsa" --timeout 600 2>&1 | tail -3 > gpurun_out/r3h_pytest.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r3h_trace" -o t -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > "$R/gpurun_out/r3h_trace.log" 2>&1
cd $R; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c60-200
cat gpurun_out/r3h_pytest.log
