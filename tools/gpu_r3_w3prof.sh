#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3w3
mkdir -p $OUT
for d in 0 6 7; do
  (cd /tmp && NASSEG_W3_DBG=$d KB_ONLY_HEAD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$d -o run -- python $OLDPWD/tools/kbench_wgrad.py > $OUT/p$d.log 2>&1)
  f=$(find $OUT/p$d -name "*kernel_stats.csv" | head -1)
  echo "== dbg $d"; head -4 $f | cut -c1-160
done
