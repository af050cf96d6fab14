#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_internimage.py -q --timeout 900 2>&1 | tail -12 | tee $O/pytest_intern.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee $O/smoke.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_errors.json"))
for g, v in d.items():
    if "intern" in g:
        top = sorted(v.items(), key=lambda kv: -kv[1])[:4]
        print(g, [(k, float("%.3g" % x)) for k, x in top])
PY
