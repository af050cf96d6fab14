#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --timer-every 20 > gpurun_out/r3k_bench_200.json 2> gpurun_out/r3k_bench_200.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3k_intern_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer > $GRAFT_REPO_ROOT/gpurun_out/r3k_intern_trace.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r3k_intern_trace/t_kernel_trace.csv
cut -c1-1500 gpurun_out/r3k_bench_200.json; tail -3 gpurun_out/r3k_bench_200.err
head -45 gpurun_out/r3k_intern_trace/t_kernel_stats.csv | cut -c1-180
