#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3wide
mkdir -p $OUT; rm -f $OUT/summary.log
for d in 0 1 2 3; do
  echo "== dbg $d" >> $OUT/summary.log
  NASSEG_PW_DBG=$d KBENCH_WIDE=1 timeout 300 python tools/kbench_pwbwd.py 2>&1 | grep "224, 64\|128, 64" >> $OUT/summary.log
done
cat $OUT/summary.log
