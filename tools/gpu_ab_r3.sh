#!/bin/bash
# A/B on ONE box: this tree against round 3's final tree (_ab_r3, built in the container:
#   git worktree add _ab_r3 5d67da9 && (cd _ab_r3 && python -c 'import __graft_entry__ as g; g.build()');
#   _ab_r3/ is git-ignored and removed again after the measurement)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4ab; mkdir -p $OUT
run() { (cd $1 && shift && python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"); }
for i in 1 2; do
  for t in . _ab_r3; do
    echo "$t headline          $(run $t --steps 12 --warmup 4)" | tee -a $OUT/summary.log
    echo "$t arch1             $(run $t --workload arch1 --steps 8 --warmup 3)" | tee -a $OUT/summary.log
    echo "$t cvpr321 --graph 2 $(run $t --workload cvpr321 --graph 2 --steps 20 --warmup 3)" | tee -a $OUT/summary.log
    echo "$t search713 --graph 2 $(run $t --workload search713 --graph 2 --steps 20 --warmup 3)" | tee -a $OUT/summary.log
    echo "$t task0             $(run $t --workload task0 --steps 20 --warmup 5)" | tee -a $OUT/summary.log
  done
done
