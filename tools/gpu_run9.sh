#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "tn_grouped" --timeout 300 2>&1 | tail -15 > gpurun_out/r3i_pytest.log
timeout 300 python tools/ab_wgrad.py 4 1 4 > gpurun_out/r3i_ab_wgrad.txt 2>&1
cat gpurun_out/r3i_pytest.log gpurun_out/r3i_ab_wgrad.txt
