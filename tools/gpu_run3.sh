#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "p8_bit_identical" --timeout 600 2>&1 | tail -5 > gpurun_out/r3c_pytest.log
MTP_AB_ROTATE=8 timeout 400 python tools/ab_gemm.py 4 512 $((512+131072)) $((512+1048576)) $((512+2097152)) > gpurun_out/r3c_ab_gemm_rot8.txt 2>&1
for v in 0 1048576 2097152 0 2097152; do
  MTP_NT_VARIANT=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-330 >> gpurun_out/r3c_bench_variants.txt
done
cat gpurun_out/r3c_pytest.log gpurun_out/r3c_ab_gemm_rot8.txt; cut -c60-200 gpurun_out/r3c_bench_variants.txt
