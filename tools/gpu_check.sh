#!/bin/bash
# Run on the MI355X box through gpurun: GPU test-suite (one pytest process per
# file so that a faulting kernel cannot take the other files down), smoke(),
# a short bench and a rocprofv3 kernel-trace summary.  Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
WHAT="${1:-all}"
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
(rocm-smi --showuniqueid --showpower 2>/dev/null | grep -i "unique\|power" | head -4) >> $OUT/env.log; nproc >> $OUT/env.log; grep -m1 "model name" /proc/cpuinfo >> $OUT/env.log
if [[ "$WHAT" == "all" || "$WHAT" == "tests" ]]; then
  for f in tests/test_hip_kernels.py tests/test_hip_irdw.py tests/test_hip_anchor.py tests/test_hip_golden.py tests/test_hip_engine.py tests/test_hip_bf16.py tests/test_hip_fullsize.py tests/test_hip_optim.py; do
    n=$(basename $f .py)
    timeout 900 python -m pytest $f -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -s > $OUT/$n.log 2>&1
    echo "$n exit $?" >> $OUT/summary.log
    tail -3 $OUT/$n.log >> $OUT/summary.log
  done
fi
if [[ "$WHAT" == "all" || "$WHAT" == "smoke" ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" >> $OUT/summary.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == "bench" ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 --breakdown --shapes 60 > $OUT/bench.json 2> $OUT/bench.err
  echo "bench exit $?" >> $OUT/summary.log
  cat $OUT/bench.json >> $OUT/summary.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == "graph" ]]; then
  for g in 0 1 2; do
    timeout 300 python bench.py --steps 20 --warmup 5 --graph $g --no-cpu-baseline --no-roofline --secondary 0 > $OUT/bench_graph$g.json 2> $OUT/bench_graph$g.err
    echo "bench --graph $g exit $?" >> $OUT/summary.log; cat $OUT/bench_graph$g.json >> $OUT/summary.log
  done
  for wl in depth480; do
    timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $OUT/bench_${wl}_g0.json 2> $OUT/bench_${wl}_g0.err
    echo "bench $wl exit $?" >> $OUT/summary.log; cat $OUT/bench_${wl}_g0.json >> $OUT/summary.log
  done
  timeout 300 python bench.py --workload depth480 --dtype bf16 --graph 2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $OUT/bench_depth480_bf16_g2.json 2> $OUT/bench_depth480_bf16_g2.err
  echo "bench depth480 bf16 --graph 2 exit $?" >> $OUT/summary.log; cat $OUT/bench_depth480_bf16_g2.json >> $OUT/summary.log
  for wl in headline depth480; do
    timeout 300 python bench.py --workload $wl --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $OUT/bench_${wl}_bf16.json 2> $OUT/bench_${wl}_bf16.err
    echo "bench $wl bf16 exit $?" >> $OUT/summary.log; cat $OUT/bench_${wl}_bf16.json >> $OUT/summary.log
  done
  timeout 300 python bench.py --workload task0 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_task0_auto.json 2> $OUT/bench_task0_auto.err
  echo "bench task0 (auto graph) exit $?" >> $OUT/summary.log; cat $OUT/bench_task0_auto.json >> $OUT/summary.log
  timeout 300 python bench.py --workload task0 --graph 0 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_task0_g0.json 2> $OUT/bench_task0_g0.err
  echo "bench task0 --graph 0 exit $?" >> $OUT/summary.log; cat $OUT/bench_task0_g0.json >> $OUT/summary.log
  timeout 300 python bench.py --workload teacher --steps 8 --warmup 2 > $OUT/bench_teacher.json 2> $OUT/bench_teacher.err
  echo "bench teacher exit $?" >> $OUT/summary.log; cat $OUT/bench_teacher.json >> $OUT/summary.log
  for wl in arch1 cvpr321 search713; do
    for g in 0 2; do
      timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --graph $g --no-cpu-baseline --secondary 0 > $OUT/bench_${wl}_g$g.json 2> $OUT/bench_${wl}_g$g.err
      echo "bench $wl --graph $g exit $?" >> $OUT/summary.log; cat $OUT/bench_${wl}_g$g.json >> $OUT/summary.log
    done
  done
fi
if [[ "$WHAT" == "all" || "$WHAT" == "prof" ]]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o run -- \
     python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --graph 0 --secondary 0 > "$OLDPWD/$OUT/prof.log" 2>&1)
  echo "prof exit $?" >> $OUT/summary.log
  find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/summary.log
fi
cat $OUT/summary.log
