#!/bin/bash
# round 5, call H: the new parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 -k "448 or 512 or stand_in or batched_small or flash_large or cell_edge or side_stream" 2>&1 | tail -25 > $O/pytest.log
cat $O/pytest.log
