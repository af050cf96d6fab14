#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_internimage.py tests/test_hip_ops.py -q -k "internimage or rvsa or nt" --timeout 900 2>&1 | tail -8 | tee $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --image-size 448 --batch 16 --use-ckpt --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_448_ckpt.json 2>> $O/bench.err
cut -c1-260 $O/bench.json $O/bench_448_ckpt.json; tail -2 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d.get('forward_only')); d=json.load(open('$O/bench_448_ckpt.json')); print(d['config'], d.get('forward_only'))"
