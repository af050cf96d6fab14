#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_backbone.py tests/test_hip_internimage.py -m gpu -q -x --timeout 600 -k "side_stream" 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2; do
  for cfg in "0 1" "1 1" "1 2" "1 3" "2 2"; do
    set -- $cfg
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/vitl_ss$1_k$2_$i.json
  done
  for cfg in "0 1" "1 3" "2 3" "2 5"; do
    set -- $cfg
    timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --wgrad-side-stream $1 --wgrad-keep $2 2>>$O/err.log | tail -1 > $O/intern_ss$1_k$2_$i.json
  done
done
MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream 0 2>>$O/err.log | tail -1 > $O/vitlcomm_ss0_k1_1.json
MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream 1 2>>$O/err.log | tail -1 > $O/vitlcomm_ss1_k1_1.json
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s2/"
tags = sorted(set(os.path.basename(f).rsplit("_", 1)[0] for f in glob.glob(O + "*.json")))
for tag in tags:
    v = []
    for f in sorted(glob.glob(O + tag + "_?.json")):
        try:
            d = json.load(open(f)); v.append(d["ms_per_step"])
            if "comm" in d: v.append(("exposed", d["comm"].get("exposed_comm_ms")))
        except Exception as e: v.append(str(e)[:40])
    print(tag, v)
PY
grep -v amdgpu.ids $O/err.log | tail -5 | cut -c1-300
