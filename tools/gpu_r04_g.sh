#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-200 $O/bench.json; tail -2 $O/bench.err
