#!/bin/bash
# round 5, call K: RVSA kernels after the VALU diet -- parity, op timing before / after, step A/B; the f14 / with_cp tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 -k "rvsa or patch_size_8 or with_cp or vit_l_headline or small_model or near_edge" 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
for l in _base/mtp_amd/libmtp_hip.so "" _base/mtp_amd/libmtp_hip.so ""; do
  echo "lib=${l:-this tree}" >> $O/bench_ops_attn.txt
  MTP_HIP_LIB=$l timeout -s KILL 300 python tools/bench_ops.py attn 2>&1 | grep -E "rvsa" >> $O/bench_ops_attn.txt
done
cat $O/bench_ops_attn.txt
bash tools/gpu_ab_step.sh r05_k_ab
