#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_tn_nostagger.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/tnns_$i.json 2>> $O/new.err
done
python - <<PY
import json, glob
for tag in ("new", "tnns"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
tail -2 $O/new.err
