#!/bin/bash
# round 5, call F: the full GPU suite on the new defaults (grouped TN with read-ahead phases, strip kernel for problems of 40-128 tiles) + step A/B + other configs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for i in 1 2; do
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only > $O/step_new_$i.json 2>> $O/err.log
done
timeout -s KILL 300 python bench.py --model vit_b --batch 32 --heads standin3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vitb_b32_standin3.json 2>> $O/err.log
timeout -s KILL 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_internimage_xl_512_b8.json 2>> $O/err.log
for f in $O/*.json; do echo $f; cut -c1-260 $f; done
tail -5 $O/err.log
