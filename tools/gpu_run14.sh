#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_internimage.py tests/test_hip_dcnv3.py -m gpu -q -x --timeout 600 -k "tn or wgrad or internimage or dcnv3 or gather" 2>&1 | tail -25 > gpurun_out/r3o_pytest.log
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3o_bench_intern.json 2> gpurun_out/r3o_bench_intern.err
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --timer-every 1 --gemm-shapes > gpurun_out/r3o_shapes.json 2> gpurun_out/r3o_shapes.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3o_bench_vitl.json 2>> gpurun_out/r3o_bench_intern.err
cat gpurun_out/r3o_pytest.log; cut -c1-300 gpurun_out/r3o_bench_intern.json gpurun_out/r3o_bench_vitl.json; tail -5 gpurun_out/r3o_bench_intern.err; head -30 gpurun_out/r3o_shapes.txt
