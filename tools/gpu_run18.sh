#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_gemm_mid.py 3 > gpurun_out/r3s_ab_gemm_mid.txt 2>&1
timeout 600 python -m pytest tests/test_hip_backbone.py -m gpu -q -x --timeout 600 -k "non_square" 2>&1 | tail -5 >> gpurun_out/r3s_ab_gemm_mid.txt
cat gpurun_out/r3s_ab_gemm_mid.txt
