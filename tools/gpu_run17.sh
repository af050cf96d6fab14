#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py -m gpu -q -x --timeout 900 -k "layernorm or backbone or headline or vit or block or f7 or f8 or f13" 2>&1 | tail -5 > gpurun_out/r3r_pytest.log
for i in 1 2 3; do
MTP_LN_BWD_1WAVE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1wave', d['value'], d['ms_per_step'])" >> gpurun_out/r3r_ab_ln.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('2wave', d['value'], d['ms_per_step'])" >> gpurun_out/r3r_ab_ln.txt
done
cat gpurun_out/r3r_pytest.log gpurun_out/r3r_ab_ln.txt
