#!/bin/bash
# round 4: SQ counters of the RVSA kernels at the ViT-L geometry (evidence for "VALU-issue / chain-latency bound")
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04w; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/a -o p -- python $R/tools/bench_ops.py attn > $O/a.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -o p -- python $R/tools/bench_ops.py attn > $O/b.log 2>&1
cd $R
python tools/pmc_summary.py $O/a/p_counter_collection.csv $O/b/p_counter_collection.csv 2>&1 | grep -i "rvsa" > $O/sq_rvsa.txt
rm -f $O/a/p_kernel_trace.csv $O/b/p_kernel_trace.csv $O/a/p_counter_collection.csv $O/b/p_counter_collection.csv
cat $O/sq_rvsa.txt | cut -c1-400; tail -3 $O/a.log | cut -c1-200
