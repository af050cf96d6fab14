#!/bin/bash
# round 5, call I: the persistent NT kernel without the per-tile vmcnt(0) drain after the epilogue (ablation library): bit-identity, micro-benchmark, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
MTP_HIP_LIB=tools/_abl/libmtp_hip_p8_nodrain.so timeout -s KILL 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_nt" --timeout 600 2>&1 | tail -4 > $O/pytest_nodrain.log
cat $O/pytest_nodrain.log
for l in "" tools/_abl/libmtp_hip_p8_nodrain.so "" tools/_abl/libmtp_hip_p8_nodrain.so; do
  echo "lib=$l" >> $O/ab_p8_nodrain.txt
  MTP_HIP_LIB=$l MTP_AB_ROTATE=8 timeout -s KILL 300 python tools/ab_gemm.py 3 256 2>&1 | grep -E "^(bias|gelu_dg|mul|res)" >> $O/ab_p8_nodrain.txt
done
cat $O/ab_p8_nodrain.txt
for i in 1 2 3; do
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/step_drain_$i.json 2>> $O/err.log
  MTP_HIP_LIB=tools/_abl/libmtp_hip_p8_nodrain.so timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/step_nodrain_$i.json 2>> $O/err.log
done
python - <<PY | tee $O/ab_step.txt
import json, glob
for tag in ("drain", "nodrain"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/step_%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
