#!/bin/bash
# HISTORICAL (how profiles/r05_ab_late_convt_packing.txt was taken): needs tools/_abl/libmtp_hip_prev2.so = the library of commit b475aa6
if [ "$MTP_RUN_HISTORICAL" != "1" ]; then echo "tools/gpu_r05_late_convt_packing.sh: historical record of a measurement -- see its header; set MTP_RUN_HISTORICAL=1 to run it anyway" >&2; exit 1; fi
# round 5, late (4): tiled ConvTranspose2d weight packing / gradient unpacking.  Tests, then the step against the library of commit b475aa6, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05_q; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backbone.py -m gpu -q -x --timeout 600 -k "weight_packing or weight_images or tokens_nchw or vit_l_headline or small_model_forward or vit_b_config1 or patch_size_8 or vitdet" 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2 3 4; do
  MTP_HIP_LIB=$R/tools/_abl/libmtp_hip_prev2.so timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/prev_$i.json 2>> $O/prev.err
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/new_$i.json 2>> $O/new.err
done
python - <<PY | tee $O/ab.txt
import json, glob
print("# same box, 20 steps each, interleaved; ms per step.  prev = library of commit b475aa6; new = + tiled ConvTranspose2d weight packing / gradient unpacking")
for tag in ("prev", "new"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("$O/%s_*.json" % tag))]
    print(tag, v, "min %.3f median %.3f" % (min(v), sorted(v)[len(v) // 2]))
PY
python - <<'PY' | tee -a $O/ab.txt
import torch, mtp_amd
from mtp_amd import ops
w = torch.randn(1024, 1024, 2, 2, device="cuda"); wg = torch.empty(4096, 1024, device="cuda", dtype=torch.bfloat16); wgT = torch.empty(1024, 4096, device="cuda", dtype=torch.bfloat16)
dwg = torch.randn(4096, 1024, device="cuda"); dw = torch.empty_like(w)
def t(fn, k=50):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / k * 1e3
print("# 1024 x 1024 x 2 x 2: convt_pack %.1f us, convt_unpack_grad %.1f us (element-wise kernels of round 4: 31 / 25 us in the step)" % (t(lambda: ops.convt_pack(w, wg, wgT)), t(lambda: ops.convt_unpack_grad(dwg, dw))))
PY
tail -n 2 $O/new.err
