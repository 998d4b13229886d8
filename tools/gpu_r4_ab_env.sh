#!/bin/bash
# A/B of one environment switch on one box: usage tools/gpu_r4_ab_env.sh VAR "v1 v2" outdir [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
VAR=$1; VALS=$2; OUT=gpurun_out/$3; shift 3
mkdir -p $OUT; rm -f $OUT/summary.log
for rep in 1 2; do
for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 "$@" > $OUT/b_$v.json 2> $OUT/b_$v.err
  python -c "import json; d=json.loads(open('$OUT/b_$v.json').read().strip().splitlines()[-1]); r=d['roofline'] or {}; print('$VAR=$v $*:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calls', r.get('nasseg_calls_per_step'), 'tiny', r.get('tiny_launches_per_step'), round(r.get('tiny_launch_ms_per_step') or 0,2), 'ms')" | tee -a $OUT/summary.log
done
done
