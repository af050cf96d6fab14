#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e_attn_stats -o t -- python $R/tools/ab_full_attn.py 14 14 64 16 > $R/gpurun_out/r3e_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r3e_attn_pmc1 -o t -- python $R/tools/ab_full_attn.py 14 14 64 16 > $R/gpurun_out/r3e_b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/r3e_attn_pmc2 -o t -- python $R/tools/ab_full_attn.py 14 14 64 16 > $R/gpurun_out/r3e_c.log 2>&1
cd $R; ls gpurun_out/r3e_attn_stats gpurun_out/r3e_attn_pmc1 | head -20
grep -h "v3_\|Name" gpurun_out/r3e_attn_stats/*kernel_stats.csv | cut -c1-200
