#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04s5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2; do
  for ss in 0 -1; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/vitl_ss${ss}_$i.json
    timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/intern_ss${ss}_$i.json
    MTP_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --wgrad-side-stream $ss 2>>$O/err.log | tail -1 > $O/comm_ss${ss}_$i.json
  done
done
timeout 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --wgrad-side-stream 0 2>>$O/err.log | tail -1 > $O/v448_ss0_1.json
timeout 300 python bench.py --image-size 448 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only 2>>$O/err.log | tail -1 > $O/v448_ss-1_1.json
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04s5/"
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
