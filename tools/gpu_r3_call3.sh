#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3c3
mkdir -p $OUT
B="python bench.py --gpus 2 --same-device --backend gloo --batch 2 --steps 16 --warmup 3 --no-cpu-baseline --no-roofline --step-times --graph 1"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 200 $B > $OUT/b2_$n.json 2> $OUT/b2_$n.err; echo "[$n] rc $?" >> $OUT/summary.log
  grep "step ms" $OUT/b2_$n.err >> $OUT/summary.log
}
run omp1 OMP_NUM_THREADS=1
run hwq1 GPU_MAX_HW_QUEUES=1
run hwq2 GPU_MAX_HW_QUEUES=2
run default FOO=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   tools/repro_dp_graph.py headline 40 2 graph+ar > $OUT/repro_headline40.log 2>&1; echo "repro $?" >> $OUT/summary.log
grep -h "^rank" $OUT/repro_headline40.log >> $OUT/summary.log
cat $OUT/summary.log
