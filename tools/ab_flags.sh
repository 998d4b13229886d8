#!/bin/bash
# A/B of compile-time variants on the GPU box: tools/ab_flags.sh <source.hip> <kbench.py|-> FLAG1 FLAG2 ...
# (each FLAG is one -D...; "none" = no flag).  Rebuilds the one source per variant, runs the micro-benchmark
# and a graph-replay headline bench; leaves the library built with the LAST variant.
src=$1; kb=$2; shift 2
mkdir -p gpurun_out
for f in "$@"; do
  touch nas-segm-pytorch_amd/csrc/$src
  if [ "$f" = none ]; then NASSEG_EXTRA_FLAGS="" python nas-segm-pytorch_amd/build.py >/dev/null
  else NASSEG_EXTRA_FLAGS="$f" python nas-segm-pytorch_amd/build.py >/dev/null; fi
  echo "=== $f"
  [ "$kb" != "-" ] && python tools/$kb 2>&1 | tail -14
  for i in 1 2; do python bench.py --graph 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'])"; done
done
