#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py tests/test_abi.py -q -k "small_linear or rvsa or full_attention or small or vit_l or f8 or f13 or abi or reduce or checkpoint or det or taps" --timeout 900 2>&1 | tail -8 | tee $O/pytest.log
for i in 1 2; do
  (cd $R/_base && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base_$i.json 2>> $O/base.err)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_tn_nostagger.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/tnns_$i.json 2>> $O/new.err
done
python - <<PY
import json, glob
for tag in ("base", "new", "tnns"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
tail -2 $O/new.err
