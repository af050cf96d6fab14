#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04r; mkdir -p $O
export TMPDIR=/tmp
MTP_AB_C2=1 MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_c2.so timeout 600 python tools/ab_gemm_mid.py 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_gemm_mid_c2.txt
