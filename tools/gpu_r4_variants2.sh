#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --no-cpu-baseline --no-roofline --pmc 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f img/s %7.3f ms'%(d['value'],d['ms_per_step']))"; }
for rep in 1 2; do
for lib in base "$@"; do
  if [ $lib = base ]; then unset NASSEG_LIB; else export NASSEG_LIB=$PWD/tools/build/variants/lib_$lib.so; fi
  echo "$lib depth480 bf16 g2 $(run --workload depth480 --dtype bf16 --graph 2 --steps 20 --warmup 3)"
  echo "$lib cvpr321 g2       $(run --workload cvpr321 --graph 2 --steps 20 --warmup 3)"
  echo "$lib task0            $(run --workload task0 --steps 20 --warmup 3)"
  echo "$lib teacher          $(run --workload teacher --steps 8 --warmup 2)"
done; done
