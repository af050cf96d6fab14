#!/bin/bash
# round 5, call E: split read-ahead phases on the grouped TN kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_tn" --timeout 600 2>&1 | tail -5 > $O/pytest_tn.log
cat $O/pytest_tn.log
timeout -s KILL 300 python tools/ab_wgrad.py 5 2 4 > $O/ab_wgrad_ra2.txt 2>&1
cat $O/ab_wgrad_ra2.txt
