#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s4; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  for ss in 0 1; do
    MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/comm_q4_ss${ss}_$i.json
    GPU_MAX_HW_QUEUES=8 MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/comm_q8_ss${ss}_$i.json
  done
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream 1 2>>$O/err.log | tail -1 > $O/plain_q8_ss1_$i.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream 1 2>>$O/err.log | tail -1 > $O/plain_q4_ss1_$i.json
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s4/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try:
            d = json.load(open(f)); v.append(d["ms_per_step"])
            if "comm" in d: v.append(("nocomm", d["comm"].get("ms_per_step_without_comm")))
        except Exception as e: v.append(str(e)[:40])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
