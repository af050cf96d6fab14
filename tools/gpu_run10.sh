#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dcnv3.py -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r3n_pytest.log
timeout 600 python tools/bench_ops.py dcnv3 > gpurun_out/r3n_dcnv3_ops.txt 2>&1
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3n_bench_intern.json 2> gpurun_out/r3n_bench_intern.err
MTP_DCNV3_VARIANT=2 timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3n_bench_intern_scatter.json 2>> gpurun_out/r3n_bench_intern.err
cat gpurun_out/r3n_pytest.log; cat gpurun_out/r3n_dcnv3_ops.txt | tail -32; cut -c1-300 gpurun_out/r3n_bench_intern.json gpurun_out/r3n_bench_intern_scatter.json; tail -5 gpurun_out/r3n_bench_intern.err
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --timer-every 1 --gemm-shapes > gpurun_out/r3n_shapes.json 2> gpurun_out/r3n_shapes.txt; head -40 gpurun_out/r3n_shapes.txt
