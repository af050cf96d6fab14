#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04t2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  for tr in "" "fc1" "fc1,proj" "fc1,fc2,proj,qkv"; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-trickle "$tr" 2>>$O/err.log | tail -1 > "$O/vitl_tr_${tr//,/+}_$i.json"
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04t2/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try: v.append(json.load(open(f))["ms_per_step"])
        except Exception as e: v.append(str(e)[:60])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -8 | cut -c1-300
