#!/bin/bash
# the GPU suite file by file (-x) + a headline bench line; usage: tools/gpu_quick.sh [tests|bench|all] [outdir]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
WHAT="${1:-all}"
OUT=gpurun_out/${2:-quick}
mkdir -p $OUT; rm -f $OUT/summary.log
if [[ "$WHAT" == "all" || "$WHAT" == "tests" ]]; then
  for f in tests/test_hip_kernels.py tests/test_hip_anchor.py tests/test_hip_golden.py tests/test_hip_engine.py tests/test_hip_bf16.py tests/test_hip_fullsize.py; do
    n=$(basename $f .py)
    timeout 900 python -m pytest $f -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/$n.log 2>&1
    echo "$n exit $?" >> $OUT/summary.log
    tail -2 $OUT/$n.log >> $OUT/summary.log
    grep -E "^FAILED|^ERROR" $OUT/$n.log | head -10 >> $OUT/summary.log
  done
fi
if [[ "$WHAT" == "all" || "$WHAT" == "bench" ]]; then
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 --breakdown --shapes 50 > $OUT/bench.json 2> $OUT/bench.err
  echo "bench exit $?" >> $OUT/summary.log
  python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('headline', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calls', r['nasseg_calls_per_step'], 'tiny', r['tiny_launches_per_step'], round(r['tiny_launch_ms_per_step'],2), 'ms; top', r['kernel'], round(r['frac'],3))" >> $OUT/summary.log 2>&1
fi
cat $OUT/summary.log
