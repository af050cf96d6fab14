#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s9; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  for cfg in "4 3" "4 6" "2 6" "4 2" "8 3"; do
    set -- $cfg
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-max-jobs $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/vitl_mj$1_k$2_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s9/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try:
            d = json.load(open(f)); v.append(d["ms_per_step"])
        except Exception as e: v.append(str(e)[:40])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
