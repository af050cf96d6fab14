#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "full_attention" --timeout 600 2>&1 | tail -4 > gpurun_out/r3g_pytest.log
for v in 0 1 1; do MTP_ATTN_V3=$v timeout 120 python tools/ab_full_attn.py 14 14 64 16 2>&1 | tail -1 | sed "s/^/v3=$v /" >> gpurun_out/r3g_ab_attn.txt; done
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3g_attn_stats -o t -- python $R/tools/ab_full_attn.py 14 14 64 16 > $R/gpurun_out/r3g_a.log 2>&1
cd $R; cat gpurun_out/r3g_pytest.log gpurun_out/r3g_ab_attn.txt; grep -h "v3_" gpurun_out/r3g_attn_stats/t_kernel_stats.csv | awk -F, '{print substr($1,1,60), $(NF-4)}'
