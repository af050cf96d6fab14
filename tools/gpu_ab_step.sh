#!/bin/bash
# same-box A/B of the whole step: round-3 baseline tree (_base/, a git worktree of b249d9b with its own libmtp_hip.so) vs the working tree, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/${1:-ab}; mkdir -p $O
export TMPDIR=/tmp
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
tail -2 $O/new.err
