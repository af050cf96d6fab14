#!/bin/bash
# HISTORICAL (how profiles/r05_ab_p8_loop_ablations.txt was taken): needs the ablation libraries tools/_abl/libmtp_hip_{nomfma,noreads,nodma,nobarrier,...}.so described in tools/ablation/README.md
if [ "$MTP_RUN_HISTORICAL" != "1" ]; then echo "tools/gpu_r05_p8_loop_ablations.sh: historical record of a measurement -- see its header; set MTP_RUN_HISTORICAL=1 to run it anyway" >&2; exit 1; fi
# round 5, call C: where the 8-wave NT loop's time goes -- ablation builds of the phase (no ds_reads / no DMA / no barriers / no MFMAs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for l in "" nomfma noreads nodma nobarrier nomfma_noreads noreads_nodma ""; do
  echo "lib=$l" >> $O/ab_p8_loop.txt
  if [ -n "$l" ]; then export MTP_HIP_LIB=tools/_abl/libmtp_hip_p8_$l.so; else unset MTP_HIP_LIB; fi
  MTP_AB_ROTATE=8 timeout -s KILL 300 python tools/ab_gemm.py 3 256 2>&1 | grep -E "^bias" | head -5 >> $O/ab_p8_loop.txt
done
cat $O/ab_p8_loop.txt
