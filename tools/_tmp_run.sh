#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for st in 0 1 2 3 4 5 6; do
  MTP_RVSA_STOP=$st MTP_RVSA_SCATTER=dense timeout 120 python tools/ab_rvsa.py 2>/dev/null | head -1 | sed "s/^/stop=$st /"
done
timeout 120 python tools/ab_rvsa.py 2>/dev/null
