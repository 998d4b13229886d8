#!/bin/bash
# per-shape A/B of compile-time variants: tools/ab_breakdown.sh <source.hip> FLAG1 FLAG2 ... -> gpurun_out/ab_<i>.txt
src=$1; shift
mkdir -p gpurun_out
i=0
for f in "$@"; do
  touch nas-segm-pytorch_amd/csrc/$src
  if [ "$f" = none ]; then NASSEG_EXTRA_FLAGS="" python nas-segm-pytorch_amd/build.py >/dev/null
  else NASSEG_EXTRA_FLAGS="$f" python nas-segm-pytorch_amd/build.py >/dev/null; fi
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown --shapes 1000 2> gpurun_out/ab_$i.txt | cut -c80-200
  i=$((i+1))
done
