#!/bin/bash
# HISTORICAL (how profiles/r05_ab_late_adamw_nontemporal.txt was taken): the MTP_ADAMW_NT switch existed only in the experiment's library (nontemporal loads / stores in adamw_kernel) and was removed
if [ "$MTP_RUN_HISTORICAL" != "1" ]; then echo "tools/gpu_r05_late_adamw_nt.sh: historical record of a measurement -- see its header; set MTP_RUN_HISTORICAL=1 to run it anyway" >&2; exit 1; fi
# round 5, late (3): AdamW with streaming (nontemporal) hints on its 28 B per parameter, same library, switch by environment; interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05_p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x --timeout 600 -k "adamw" 2>&1 | tail -3 | tee $O/pytest.log
MTP_ADAMW_NT=1 timeout -s KILL 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x --timeout 600 -k "adamw" 2>&1 | tail -3 | tee -a $O/pytest.log
for i in 1 2 3; do
  MTP_ADAMW_NT=0 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/plain_$i.json 2>> $O/plain.err
  MTP_ADAMW_NT=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/nt_$i.json 2>> $O/nt.err
done
python - <<PY | tee $O/ab.txt
import json, glob
print("# same box, 20 steps each, interleaved; ms per step.  AdamW kernel: plain loads / stores vs nontemporal loads and stores")
for tag in ("plain", "nt"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f" % min(v))
PY
python - <<'PY' | tee -a $O/ab.txt
import os, time, torch
import mtp_amd
from mtp_amd import ops
n = 303_000_000 // 4 * 4
p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
v.abs_()
seg = torch.tensor([0, n // 2 // 4 * 4], dtype=torch.int64, device="cuda"); wd = torch.tensor([0.05, 0.0], device="cuda")
hyper = torch.tensor([6e-5, 0.9, 0.999, 1e-8, 0.1, 0.001], device="cuda")
def run(k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops.adamw_flat(p, g, m, v, seg, wd, hyper)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
run(3)
print("# adamw_flat alone, 303 M parameters, MTP_ADAMW_NT=%s: %.1f us per launch" % (os.environ.get("MTP_ADAMW_NT", "0"), run(20) * 1e3))
PY
MTP_ADAMW_NT=1 python - <<'PY' | tee -a $O/ab.txt
import os, torch
import mtp_amd
from mtp_amd import ops
n = 303_000_000 // 4 * 4
p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
v.abs_()
seg = torch.tensor([0, n // 2 // 4 * 4], dtype=torch.int64, device="cuda"); wd = torch.tensor([0.05, 0.0], device="cuda")
hyper = torch.tensor([6e-5, 0.9, 0.999, 1e-8, 0.1, 0.001], device="cuda")
def run(k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops.adamw_flat(p, g, m, v, seg, wd, hyper)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
run(3)
print("# adamw_flat alone, 303 M parameters, MTP_ADAMW_NT=1: %.1f us per launch" % (run(20) * 1e3))
PY
