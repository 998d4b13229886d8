#!/bin/bash
# the other BASELINE workloads, host-launched (g0) and replayed (g2): one JSON line each
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4wl}
mkdir -p $OUT
run() { # name args...
  n=$1; shift
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 "$@" > $OUT/$n.json 2> $OUT/$n.err
  python -c "import json; d=json.loads(open('$OUT/$n.json').read().strip().splitlines()[-1]); r=d['roofline'] or {}; print('$n', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; calls', r.get('nasseg_calls_per_step'), 'tiny', r.get('tiny_launches_per_step'), round(r.get('tiny_launch_ms_per_step') or 0,2), 'ms; top', r.get('kernel'), round(r.get('frac') or 0,3))" 2>&1 | tee -a $OUT/summary.log
}
rm -f $OUT/summary.log
run arch1_g0 --workload arch1
run cvpr321_g0 --workload cvpr321
run cvpr321_g2 --workload cvpr321 --graph 2
run depth480_bf16_g0 --workload depth480 --dtype bf16
run depth480_bf16_g2 --workload depth480 --dtype bf16 --graph 2
run depth480_g0 --workload depth480
run search713_g0 --workload search713
run search713_g2 --workload search713 --graph 2
run task0_auto --workload task0
run task0_g0 --workload task0 --graph 0
run headline_bf16_g0 --dtype bf16
run headline_bf16_g1 --dtype bf16 --graph 1
run headline_g1 --graph 1
