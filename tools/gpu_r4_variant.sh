#!/bin/bash
# A/B of the in-tree library against tools/build/variants/lib_<name>.so on one box: kbench + workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$1
run() { python bench.py --no-cpu-baseline --no-roofline --pmc 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f img/s %7.3f ms'%(d['value'],d['ms_per_step']))"; }
for rep in 1 2; do
for lib in base $V; do
  if [ $lib = base ]; then unset NASSEG_LIB; else export NASSEG_LIB=$PWD/tools/build/variants/lib_$lib.so; fi
  [ $rep = 1 ] && python tools/kbench_dwswz.py 2>&1 | grep "C=" | cut -c1-66
  echo "$lib headline   $(run --steps 8 --warmup 3)"
  echo "$lib arch1      $(run --workload arch1 --steps 8 --warmup 3)"
  echo "$lib cvpr321 g2 $(run --workload cvpr321 --graph 2 --steps 20 --warmup 3)"
done; done
