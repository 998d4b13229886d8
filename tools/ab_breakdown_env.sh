#!/bin/bash
# like ab_breakdown.sh, with environment settings for the bench: tools/ab_breakdown_env.sh "<ENV=..>" <source.hip> FLAG...
envs=$1; src=$2; shift 2
mkdir -p gpurun_out
i=0
for f in "$@"; do
  touch nas-segm-pytorch_amd/csrc/$src
  if [ "$f" = none ]; then NASSEG_EXTRA_FLAGS="" python nas-segm-pytorch_amd/build.py >/dev/null
  else NASSEG_EXTRA_FLAGS="$f" python nas-segm-pytorch_amd/build.py >/dev/null; fi
  env $envs python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown --shapes 1000 2> gpurun_out/ab_$i.txt | cut -c80-200
  i=$((i+1))
done
