#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
for i in 1 2 3; do timeout -s KILL 600 python -m pytest tests/test_hip_internimage.py -m gpu -q --timeout 600 -k "with_cp" 2>&1 | grep -E "assert|Error|passed|failed" | head -8; done
