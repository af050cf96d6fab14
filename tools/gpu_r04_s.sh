#!/bin/bash
# round 4: weight gradients on a side stream for InternImage-XL / ViT-B (under-filled NT GEMMs leave CUs idle), interleaved A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_internimage.py  -m gpu -q -x --timeout 600 -k "side_stream or one_step or padded" 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2; do
  for cfg in "0 1" "1 1" "2 1" "2 2" "2 3"; do
    set -- $cfg
    timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --wgrad-side-stream $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/intern_ss$1_k$2_$i.json
  done
  for ss in 0 1 2; do
    timeout 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/vitb_ss${ss}_$i.json
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/vitl_ss${ss}_$i.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try: v.append(json.load(open(f))["ms_per_step"])
        except Exception as e: v.append(str(e)[:40])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | tail -5 | cut -c1-300
