#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3head
mkdir -p $OUT; rm -f $OUT/summary.log
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dense_conv" > $OUT/test.log 2>&1
echo "dense conv tests $?" >> $OUT/summary.log; tail -3 $OUT/test.log >> $OUT/summary.log
KBENCH_ONLY_3X3=1 timeout 300 python tools/kbench.py conv wgrad dgrad 2>&1 | grep -v amdgpu.ids >> $OUT/summary.log
cat $OUT/summary.log
