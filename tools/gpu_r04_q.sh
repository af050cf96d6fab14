#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_internimage.py tests/test_hip_ops.py tests/test_hip_backbone.py -q --timeout 900 -k "not xl_at_512" 2>&1 | tail -6 | tee $O/pytest.log
timeout 300 python tools/bench_ops.py attn 2>&1 | grep -i "rvsa"
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/ii.json 2> $O/ii.err; cut -c1-200 $O/ii.json; tail -2 $O/ii.err
for i in 1 2 3; do
  (cd $R/_base && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base_$i.json 2>> $O/base.err)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
done
python - <<PY
import json, glob
for tag in ("base", "new"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
