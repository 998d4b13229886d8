#!/bin/bash
# round 3, first GPU call: diagnosis (stray ATen launches, two-rank graph replay), the new anchor tests,
# the self-launching bench with two ranks on the one device, a baseline headline line with --pmc
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3c1
mkdir -p $OUT
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
(rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1; nproc; grep -m1 "model name" /proc/cpuinfo) >> $OUT/env.log
timeout 300 python tools/trace_aten.py headline 2 > $OUT/trace_aten.log 2>&1; echo "trace_aten $?" >> $OUT/summary.log
timeout 600 python -m pytest tests/test_hip_anchor.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/test_anchor.log 2>&1
echo "anchor tests $?" >> $OUT/summary.log; tail -5 $OUT/test_anchor.log >> $OUT/summary.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   tools/repro_dp_graph.py cvpr321 24 > $OUT/repro_cvpr321.log 2>&1; echo "repro cvpr321 $?" >> $OUT/summary.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
   tools/repro_dp_graph.py headline 16 2 > $OUT/repro_headline.log 2>&1; echo "repro headline $?" >> $OUT/summary.log
timeout 300 python bench.py --gpus 2 --same-device --backend gloo --graph 1 --batch 2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline \
   > $OUT/bench_2ranks_graph1.json 2> $OUT/bench_2ranks_graph1.err; echo "bench 2 ranks graph1 $?" >> $OUT/summary.log
timeout 600 python bench.py --steps 10 --warmup 3 --breakdown --shapes 40 > $OUT/bench.json 2> $OUT/bench.err; echo "bench $?" >> $OUT/summary.log
grep -h "^rank" $OUT/repro_*.log >> $OUT/summary.log
cat $OUT/bench_2ranks_graph1.json $OUT/bench.json >> $OUT/summary.log
cat $OUT/summary.log
