#!/bin/bash
# interleaved same-box A/B of bench.py under two environments:  tools/gpu_ab_env.sh "<env A>" "<env B>" [rounds] [bench args...]
# prints ms/step of every run and the medians (rounds >= 3 recommended: boxes drift by ~0.2 ms)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
A="$1"; B="$2"; N="${3:-3}"; shift 3
for i in $(seq 1 $N); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-gemm-timer "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$tag [$E] $ms"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import statistics
r={'A':[],'B':[]}
for ln in open('/tmp/ab.txt'):
    t=ln.split(); r[t[0]].append(float(t[-1]))
for k,v in r.items(): print(k, "median %.3f ms  (%s)"%(statistics.median(v), " ".join("%.2f"%x for x in v)))
print("B - A = %.3f ms"%(statistics.median(r['B'])-statistics.median(r['A'])))
PY
