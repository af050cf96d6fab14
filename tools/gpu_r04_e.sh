#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "rvsa" --timeout 600 2>&1 | tail -3 | tee $O/pytest.log
for w4 in 1 0; do echo "MTP_RVSA_SCATTER_W4=$w4"; MTP_RVSA_SCATTER_W4=$w4 timeout 300 python tools/bench_ops.py attn 2>&1 | grep -i rvsa; done | tee $O/bench_ops_rvsa.txt
MTP_RVSA_SCATTER_W4=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_w4.json 2> $O/bench.err
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_w8.json 2>> $O/bench.err
cut -c1-200 $O/bench_w4.json $O/bench_w8.json; tail -3 $O/bench.err
