#!/bin/bash
# round 5, call B: strip kernel on the mid-size shapes (ViT-B at batch 32, InternImage-XL levels); in-flight sensitivity of the 8-wave kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
MTP_AB_SHAPES=mid MTP_AB_ROTATE=8 timeout -s KILL 600 python tools/ab_gemm.py 5 0 1024 256 131072 > $O/ab_gemm_mid.txt 2>&1
cat $O/ab_gemm_mid.txt
for l in "" tools/_abl/libmtp_hip_p8vm10.so tools/_abl/libmtp_hip_p8vm8.so ""; do
  echo "lib=$l" >> $O/ab_p8_vm.txt
  MTP_HIP_LIB=$l MTP_AB_ROTATE=8 timeout -s KILL 300 python tools/ab_gemm.py 3 256 2>&1 | head -7 >> $O/ab_p8_vm.txt
done
cat $O/ab_p8_vm.txt
