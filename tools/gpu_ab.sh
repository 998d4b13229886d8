#!/bin/bash
# A/B on ONE box: the headline bench of this tree against the round-2 tree (_ab_r2, built in the container:
#   git worktree add _ab_r2 33418a2 && (cd _ab_r2 && python -c 'import __graft_entry__ as g; g.build()');
#   _ab_r2/ is git-ignored and removed again after the measurement - it would travel with every gpurun call)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3ab; mkdir -p $OUT
for i in 1 2; do
  for t in . _ab_r2; do
    (cd $t && python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value'],1), round(d['ms_per_step'],3))") | tee -a $OUT/summary.log
  done
done
