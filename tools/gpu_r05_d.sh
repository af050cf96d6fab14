#!/bin/bash
# round 5, call D: read-ahead phases (fragments of phase g+1 read under the MFMAs of phase g) -- parity, NT and grouped-TN micro-benchmarks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" --timeout 600 2>&1 | tail -8 > $O/pytest_gemm.log
cat $O/pytest_gemm.log
timeout -s KILL 300 python tools/ab_wgrad.py 5 2 4 > $O/ab_wgrad_ra.txt 2>&1
cat $O/ab_wgrad_ra.txt
MTP_AB_ROTATE=8 timeout -s KILL 600 python tools/ab_gemm.py 5 256 $((256 + 524288)) > $O/ab_gemm_ra.txt 2>&1
cat $O/ab_gemm_ra.txt
