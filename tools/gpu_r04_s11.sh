#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s11; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x --timeout 600 -k "low_priority or reduce" 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2 3; do
  for x in 0 1; do
    MTP_X_LNFIRST=$x timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer 2>>$O/err.log | tail -1 > $O/vitl_lnfirst${x}_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s11/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    print(tag, [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob(O + tag + "_?.json"))])
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
