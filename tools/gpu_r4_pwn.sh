#!/bin/bash
# round 4: conv_pwn_kernel parity + per-shape timing + headline A/B over NASSEG_PWN_MODE (+ membench on demand)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4c3}
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "nsplit" > $OUT/pwn_tests.log 2>&1
echo "pwn tests exit $?" | tee $OUT/summary.log
tail -2 $OUT/pwn_tests.log | tee -a $OUT/summary.log
if [[ "${MEMBENCH:-0}" == "1" ]]; then timeout 120 tools/build/membench > $OUT/membench.txt 2>&1; fi
KBENCH_BF16=${KBENCH_BF16:-0} timeout 900 python tools/kbench_pwn.py all > $OUT/kbench_pwn.txt 2>&1
echo "kbench exit $?" | tee -a $OUT/summary.log
for m in ${MODES:-0 1 2}; do
  NASSEG_PWN_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --breakdown --shapes 70 > $OUT/bench_pwn$m.json 2> $OUT/bench_pwn$m.err
  python -c "import json; d=json.loads(open('$OUT/bench_pwn$m.json').read().strip().splitlines()[-1]); r=d['roofline']; print('pwn mode $m: headline', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms; top', r['kernel'], round(r['frac'],3))" | tee -a $OUT/summary.log
done
