#!/bin/bash
# round 5, call L: full GPU suite; same-box A/B of the other configurations (ViT-B batch 32 with stand-in heads, InternImage-XL 512^2 batch 8) against the round-4 tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05_l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for i in 1 2; do
  (cd $R/_base && timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/vitb_base_$i.json 2>> $O/err.log)
  timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/vitb_new_$i.json 2>> $O/err.log
  (cd $R/_base && timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/ii_base_$i.json 2>> $O/err.log)
  timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only > $O/ii_new_$i.json 2>> $O/err.log
done
python - <<PY | tee $O/ab_other_configs.txt
import json, glob
print("# same box, interleaved: _base = round 4's last commit (own libmtp_hip.so) vs this tree; ms per step / images per second / NT family TF/s")
for tag in ("vitb_base", "vitb_new", "ii_base", "ii_new"):
    rows = []
    for f in sorted(glob.glob("$O/%s_*.json" % tag)):
        d = json.load(open(f))
        fam = (d.get("roofline") or {}).get("families", {})
        rows.append((d["ms_per_step"], d["value"], fam.get("gemm_nt", {}).get("tflops"), d.get("step_mfma_frac")))
    print(tag, rows)
PY
tail -3 $O/err.log
