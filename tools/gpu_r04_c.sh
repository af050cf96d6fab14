#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py -q -k "rvsa or vit_l or small or f8 or f13" --timeout 600 2>&1 | tail -8 | tee $O/pytest_rvsa.log
for v in 0 1; do echo "MTP_RVSA_V5=$v"; MTP_RVSA_V5=$v timeout 300 python tools/bench_ops.py attn 2>&1 | grep -i rvsa; done | tee $O/bench_ops_rvsa.txt
MTP_RVSA_V5=0 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_v5off.json 2> $O/bench.err
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_v5on.json 2>> $O/bench.err
cut -c1-200 $O/bench_v5off.json $O/bench_v5on.json; tail -3 $O/bench.err
