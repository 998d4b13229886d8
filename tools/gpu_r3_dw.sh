#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3dw; mkdir -p $OUT; rm -f $OUT/summary.log
for v in 0 1; do
  echo "== rows form $v" >> $OUT/summary.log
  NASSEG_DW_WGRAD_ROWS=$v timeout 300 python tools/kbench_dwwgrad.py 2>&1 | grep dw_wgrad >> $OUT/summary.log
  (cd /tmp && NASSEG_DW_WGRAD_ROWS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$v -o run -- python $OLDPWD/tools/kbench_dwwgrad.py > $OUT/p$v.log 2>&1)
  f=$(find $OUT/p$v -name "*kernel_stats.csv" | head -1)
  head -8 $f | cut -c1-150 >> $OUT/summary.log
done
cat $OUT/summary.log
