#!/bin/bash
# same-box A/B of the whole step: baseline tree (_base/, a git worktree of the previous round's last commit with its own libmtp_hip.so) vs the working tree, interleaved
# usage: bash tools/gpu_ab_step.sh <tag> [pytest-k-expression]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/${1:-ab}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
if [ -n "$2" ]; then timeout -s KILL 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "$2" 2>&1 | tail -4 | tee $O/pytest.log; fi
for i in 1 2 3; do
  (cd $R/_base && timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/base_$i.json 2>> $O/base.err)
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
done
python - <<PY | tee $O/ab.txt
import json, glob
print("# same box, 20 steps each, interleaved: _base = previous round's last commit (own libmtp_hip.so) vs this tree; ms per step")
for tag in ("base", "new"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
tail -2 $O/new.err
