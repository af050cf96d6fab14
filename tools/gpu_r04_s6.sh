#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s6; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  for q in 4 8; do
    for cfg in "0 0" "1 0" "1 1"; do
      set -- $cfg
      GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $1 --wgrad-tail $2 2>>$O/err.log | tail -1 > $O/vitl_q${q}_ss$1_t$2_$i.json
    done
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s6/"
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
