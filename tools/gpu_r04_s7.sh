#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s7; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_backbone.py tests/test_hip_parallel.py -m gpu -q -x --timeout 900 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2 3; do
  for ax in 0 1; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gemm-timer --aux-side-stream $ax 2>>$O/err.log | tail -1 > $O/vitl_aux${ax}_$i.json
  done
done
for ax in 0 1; do
  timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-timer --aux-side-stream $ax 2>>$O/err.log | tail -1 > $O/vitb_aux${ax}_1.json
  timeout 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-timer --aux-side-stream $ax 2>>$O/err.log | tail -1 > $O/v448_aux${ax}_1.json
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s7/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try:
            d = json.load(open(f)); v.append((d["ms_per_step"], (d.get("forward_only") or {}).get("ms_per_pass")))
        except Exception as e: v.append(str(e)[:40])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
