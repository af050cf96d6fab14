#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "full_attention" --timeout 600 2>&1 | tail -3
timeout 300 python tools/bench_ops.py attn 2>&1 | grep -i "attn"
(cd $R/_base && timeout 300 python tools/bench_ops.py attn 2>&1 | grep -i "full_attn") 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, os, collections, re
o = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04n/trace/"
rows = list(csv.DictReader(open(o + "t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
print("last step: %d launches, span %.3f ms" % (len(step), span))
agg = collections.OrderedDict()
for r in step:
    n = re.sub(r'^void ', '', r["Kernel_Name"]); n = re.sub(r'\(anonymous namespace\)::', '', n)[:80]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += d
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    print("%8.1f us %4d x %7.1f  %s" % (d, c, d / c, n))
os.remove(o + "t_kernel_trace.csv")
PY
