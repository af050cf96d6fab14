#!/bin/bash
# round 4: DPP lane exchanges in the RVSA backward (60 -> 10 ds_bpermute): RVSA parity tests, then interleaved A/B against the previous library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04y; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py -m gpu -q -x --timeout 90 -k "rvsa or vit_l_forward or f13" 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2; do
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_old.so timeout 60 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-forward-only --no-gemm-timer 2>>$O/err.log | tail -1 > $O/old_$i.json
  timeout 60 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-forward-only --no-gemm-timer 2>>$O/err.log | tail -1 > $O/new_$i.json
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04y/"
for tag in ("old", "new"):
    print(tag, [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob(O + tag + "_?.json"))])
PY
