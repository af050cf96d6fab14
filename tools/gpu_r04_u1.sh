#!/bin/bash
# round 4: weight images off the critical path (side-stream refresh of the tables the forward pass does not need at once): full GPU suite + interleaved A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04u1; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2 3; do
  for x in 0 1; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --images-side-stream $x 2>>$O/err.log | tail -1 > $O/vitl_img${x}_$i.json
  done
done
for i in 1 2; do
  for x in 0 1; do
    timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-gemm-timer --images-side-stream $x 2>>$O/err.log | tail -1 > $O/intern_img${x}_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04u1/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try: v.append(json.load(open(f))["ms_per_step"])
        except Exception as e: v.append(str(e)[:60])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
