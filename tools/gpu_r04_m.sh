#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
echo "== product library, 256-row tiles (768) and persistent (768+32768)"; MTP_AB_ROTATE=8 timeout 300 python tools/ab_gemm.py 3 768 $((768 + 32768)) 2>&1 | grep -v amdgpu.ids | grep "^bias\|^gelu_dg\|^mul"
echo "== 32x32x16 timing probe (results wrong by construction)"; MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_mfma32probe.so MTP_AB_ROTATE=8 timeout 300 python tools/ab_gemm.py 3 768 $((768 + 32768)) 2>&1 | grep -v amdgpu.ids | grep "^bias\|^gelu_dg\|^mul"
done | tee $O/ab_mfma32_probe.txt
