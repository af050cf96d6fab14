#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r3m -o t -- python $GRAFT_REPO_ROOT/bench.py --model internimage_xl --image-size 512 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timer > $GRAFT_REPO_ROOT/gpurun_out/r3m.log 2>&1
head -1 /tmp/r3m/t_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/r3m_dcn_trace.csv
grep -i "dcnv3" /tmp/r3m/t_kernel_trace.csv >> $GRAFT_REPO_ROOT/gpurun_out/r3m_dcn_trace.csv
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --timer-every 1 --gemm-shapes > gpurun_out/r3m_shapes.json 2> gpurun_out/r3m_shapes.txt
head -50 gpurun_out/r3m_shapes.txt
