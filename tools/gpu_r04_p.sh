#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_internimage.py tests/test_hip_dcnv3.py -q --timeout 900 -k "not xl_at_512" 2>&1 | tail -6 | tee $O/pytest.log
for i in 1 2; do
  (cd $R/_base && timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/ii_base_$i.json 2>> $O/base.err)
  timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/ii_new_$i.json 2>> $O/new.err
done
python - <<PY
import json, glob
for tag in ("ii_base", "ii_new"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
tail -2 $O/new.err
