#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s12; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  for ss in 0 -1; do
    timeout 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/v448_ss${ss}_$i.json
    timeout 300 python bench.py --image-size 448 --batch 16 --use-ckpt --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/v448ckpt_ss${ss}_$i.json
    timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/vitb_ss${ss}_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s12/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try: v.append(json.load(open(f))["ms_per_step"])
        except Exception as e: v.append(str(e)[:60])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | grep -v socket | tail -5 | cut -c1-300
