#!/bin/bash
# round 4, call 1: the N-split pointwise kernel (parity, then timing against today's dispatch), HBM stream
# ceilings of this box, then the whole GPU suite + a headline line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4c1
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "nsplit" > $OUT/pwn_tests.log 2>&1
echo "pwn tests exit $?" | tee $OUT/summary.log
tail -3 $OUT/pwn_tests.log | tee -a $OUT/summary.log
timeout 120 tools/build/membench > $OUT/membench.txt 2>&1
echo "membench exit $?" | tee -a $OUT/summary.log
timeout 900 python tools/kbench_pwn.py all > $OUT/kbench_pwn.txt 2>&1
echo "kbench exit $?" | tee -a $OUT/summary.log
tools/gpu_quick.sh all r4c1_quick > /dev/null 2>&1
cat gpurun_out/r4c1_quick/summary.log | tee -a $OUT/summary.log
