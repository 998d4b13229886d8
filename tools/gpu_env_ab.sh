#!/bin/bash
# headline under several settings of one environment variable, on one box: tools/gpu_env_ab.sh VAR v1 v2 ...
# (BENCH_ARGS: extra bench.py arguments, e.g. BENCH_ARGS='--dtype bf16')
cd "${GRAFT_REPO_ROOT:-/root/repo}"
var=$1; shift
for v in "$@"; do
  for i in 1 2; do
    env $var=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
