#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 -k "patch_size_8 or with_cp" 2>&1 | tail -40 > $O/pytest.log
cat $O/pytest.log
