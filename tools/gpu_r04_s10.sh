#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s10; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_backbone.py tests/test_hip_internimage.py tests/test_hip_parallel.py -m gpu -q -x --timeout 600 -k "side_stream or trainer or parallel or reducer" 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2; do
  for cfg in "0 3" "14 3" "7 4" "24 3"; do
    set -- $cfg
    timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-max-jobs $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/intern_mj$1_k$2_$i.json
  done
  for cfg in "0 1" "8 2"; do
    set -- $cfg
    timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-max-jobs $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/vitb_mj$1_k$2_$i.json
    MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-max-jobs $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/comm_mj$1_k$2_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s10/"
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
