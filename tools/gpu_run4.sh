#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "full_attention" --timeout 600 2>&1 | tail -25 > gpurun_out/r3d_pytest.log
for v in 0 1 0 1; do MTP_ATTN_V3=$v timeout 120 python tools/ab_full_attn.py 14 14 64 16 2>&1 | tail -1 | sed "s/^/v3=$v /" >> gpurun_out/r3d_ab_attn.txt; done
for v in 0 1 0 1; do
  MTP_ATTN_V3=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c60-200 | sed "s/^/v3=$v /" >> gpurun_out/r3d_bench.txt
done
cat gpurun_out/r3d_pytest.log gpurun_out/r3d_ab_attn.txt gpurun_out/r3d_bench.txt
