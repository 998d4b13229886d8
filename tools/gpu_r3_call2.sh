#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3c2
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_anchor.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/test_anchor.log 2>&1
echo "anchor tests $?" >> $OUT/summary.log; tail -4 $OUT/test_anchor.log >> $OUT/summary.log
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "batch_norm or fused_conv_bn or colred or global or cat_bn" > $OUT/test_bn.log 2>&1
echo "bn kernel tests $?" >> $OUT/summary.log; tail -2 $OUT/test_bn.log >> $OUT/summary.log
B="python bench.py --gpus 2 --same-device --backend gloo --batch 2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --step-times"
for v in "--graph 1" "--graph 1 --sync-each-step" "--graph 0" "--graph 0 --sync-each-step"; do
  n=$(echo $v | tr -d ' -')
  timeout 300 $B $v > $OUT/b2_$n.json 2> $OUT/b2_$n.err; echo "bench 2 ranks [$v] $?" >> $OUT/summary.log
  grep "step ms" $OUT/b2_$n.err >> $OUT/summary.log
  python -c "import json,sys; d=json.loads(open('$OUT/b2_$n.json').read().strip().splitlines()[-1]); print('   value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1))" >> $OUT/summary.log 2>&1
done
cat $OUT/summary.log
