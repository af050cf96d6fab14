#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py tests/test_abi.py -q -k "layernorm or window or small or vit_l or f8 or f13 or abi or checkpoint" --timeout 900 2>&1 | tail -8 | tee $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/bench2.json 2>> $O/bench.err
cut -c1-200 $O/bench.json $O/bench2.json; tail -2 $O/bench.err
