#!/bin/bash
# Everything that runs on the MI355X box through gpurun, as ONE parameterised script (the per-round one-offs
# tools/gpu_r3_*.sh / gpu_r4_*.sh / gpu_ab*.sh / ab_*.sh of rounds 2-4 are folded in here).  Logs go under
# gpurun_out/<outdir>/ (merged back by gpurun); every sub-command appends one-line results to <outdir>/summary.log.
#
#   tools/gpu.sh env      OUT                       device, host cores, host memory
#   tools/gpu.sh tests    OUT [pytest -k expr]      the GPU suite, one pytest process per file
#   tools/gpu.sh smoke    OUT                       __graft_entry__.smoke()
#   tools/gpu.sh bench    OUT [bench args]          one bench.py line (+ --breakdown table in bench.err)
#   tools/gpu.sh ab       OUT VAR "v1 v2" [bench args]   one environment switch off / on, twice, on this box
#   tools/gpu.sh workloads OUT                      the other BASELINE workloads, host-launched and replayed
#   tools/gpu.sh trace    OUT [bench args]          rocprofv3 kernel trace: durations per family, gaps between kernels
#   tools/gpu.sh stats    OUT [bench args]          rocprofv3 --kernel-trace --stats summary (+ per-family table)
#   tools/gpu.sh tree     OUT DIR                   this tree against another BUILT tree (git worktree under the repo)
#   tools/gpu.sh flags    OUT SRC KBENCH|- FLAG...  compile-time variants of one source (-D flags; "none" = plain; FLAGS_ALSO)
#   tools/gpu.sh lib      OUT NAME                  in-tree library against tools/build/variants/lib_NAME.so
#   tools/gpu.sh kbench   OUT SCRIPT [args]         one tools/kbench_*.py table
#   tools/gpu.sh pmc      OUT SCRIPT "C1 C2"...     rocprofv3 --pmc passes (one per quoted counter set) over a tools/ script
# The round's final collection stays tools/gpu_check.sh all; tools/gpu_pmc.sh; tools/gpu_pmc_mfma.sh (collect_profiles.sh).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
CMD="${1:-help}"
if [[ "$CMD" == "help" || $# -lt 2 ]]; then sed -n 2,21p "$0"; exit 0; fi
OUT=gpurun_out/$2
shift 2
mkdir -p $OUT
SUM=$OUT/summary.log

line() {  # name json-file -> one summary line
  python - "$1" "$2" <<'PY'
import json, sys
name, path = sys.argv[1:3]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as e:  # (a failed run: the reason is in the .err file next to it)
    print(name, "FAILED", e)
    sys.exit(0)
r = d.get("roofline") or {}
print(name, round(d["value"], 1), d.get("unit", ""), round(d["ms_per_step"], 3), "ms; calls", r.get("nasseg_calls_per_step"),
      "tiny", r.get("tiny_launches_per_step"), round(r.get("tiny_launch_ms_per_step") or 0, 2), "ms; top", r.get("kernel"),
      round(r.get("frac") or 0, 3))
PY
}
bench() {  # name args... -> $OUT/name.json, one line
  local n=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --pmc 0 --secondary 0 "$@" > $OUT/$n.json 2> $OUT/$n.err
  line "$n" $OUT/$n.json | tee -a $SUM
}

case "$CMD" in
env)
  python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
  (rocm-smi --showuniqueid --showpower 2>/dev/null | grep -i "unique\|power" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo;
   free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null) >> $OUT/env.log
  cat $OUT/env.log | tee -a $SUM ;;
tests)
  for f in tests/test_hip_kernels.py tests/test_hip_irdw.py tests/test_hip_anchor.py tests/test_hip_golden.py tests/test_hip_engine.py \
           tests/test_hip_bf16.py tests/test_hip_fullsize.py tests/test_hip_optim.py; do
    n=$(basename $f .py)
    timeout 1200 python -m pytest $f -m gpu -q --tb=short --timeout 900 -p no:cacheprovider ${1:+-k "$1"} > $OUT/$n.log 2>&1
    echo "$n exit $?" | tee -a $SUM
    tail -2 $OUT/$n.log | tee -a $SUM
    grep -E "^FAILED|^ERROR" $OUT/$n.log | head -12 | tee -a $SUM
  done ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" | tee -a $SUM; tail -2 $OUT/smoke.log | tee -a $SUM ;;
bench)
  bench bench --steps 10 --warmup 3 --breakdown --shapes 60 "$@" ;;
ab)
  VAR=$1; VALS=$2; shift 2
  for rep in 1 2; do for v in $VALS; do
    n="${VAR}_${v}_$(echo "$*" | tr -c 'a-zA-Z0-9' '_')"
    env $VAR=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --secondary 0 "$@" > $OUT/$n.json 2> $OUT/$n.err
    line "$VAR=$v $*" $OUT/$n.json | tee -a $SUM
  done; done ;;
workloads)
  bench arch1_g0 --steps 20 --warmup 5 --workload arch1
  bench cvpr321_g0 --steps 20 --warmup 5 --workload cvpr321
  bench cvpr321_g2 --steps 20 --warmup 5 --workload cvpr321 --graph 2
  bench depth480_bf16_g0 --steps 20 --warmup 5 --workload depth480 --dtype bf16
  bench depth480_bf16_g2 --steps 20 --warmup 5 --workload depth480 --dtype bf16 --graph 2
  bench depth480_g0 --steps 20 --warmup 5 --workload depth480
  bench search713_g0 --steps 20 --warmup 5 --workload search713
  bench search713_g2 --steps 20 --warmup 5 --workload search713 --graph 2
  bench task0_auto --steps 20 --warmup 5 --workload task0
  bench task0_g0 --steps 20 --warmup 5 --workload task0 --graph 0
  bench headline_bf16_g0 --steps 20 --warmup 5 --dtype bf16
  bench headline_g1 --steps 20 --warmup 5 --graph 1
  bench headline_g2 --steps 20 --warmup 5 --graph 2 ;;
trace)
  ABS=$PWD/$OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/prof -o run -- \
     python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --pmc 0 --secondary 0 "$@" > $ABS/bench.json 2> $ABS/bench.err)
  python tools/trace_gaps.py $OUT | tee $OUT/trace.txt | head -60 ;;
stats)
  ABS=$PWD/$OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/prof -o run -- \
     python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --pmc 0 --secondary 0 "$@" > $ABS/prof.log 2>&1)
  echo "stats exit $?" | tee -a $SUM
  python tools/prof_summary.py $(find $OUT/prof -name "*kernel_stats*" | head -1) | tee $OUT/kernel_families.txt | head -40 ;;
tree)
  OTHER=$1
  run() { (cd $1 && shift && python bench.py --no-cpu-baseline --no-roofline --pmc 0 --secondary 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"); }
  for i in 1 2; do for t in . $OTHER; do
    echo "$t headline            $(run $t --steps 12 --warmup 4)" | tee -a $SUM
    echo "$t arch1               $(run $t --workload arch1 --steps 8 --warmup 3)" | tee -a $SUM
    echo "$t cvpr321 --graph 2   $(run $t --workload cvpr321 --graph 2 --steps 20 --warmup 3)" | tee -a $SUM
    echo "$t search713 --graph 2 $(run $t --workload search713 --graph 2 --steps 20 --warmup 3)" | tee -a $SUM
    echo "$t depth480 bf16 -g 2  $(run $t --workload depth480 --dtype bf16 --graph 2 --steps 20 --warmup 3)" | tee -a $SUM
    echo "$t task0               $(run $t --workload task0 --steps 20 --warmup 5)" | tee -a $SUM
  done; done ;;
flags)
  SRC=$1; KB=$2; shift 2
  for f in "$@"; do
    touch nas-segm-pytorch_amd/csrc/$SRC
    if [ "$f" = none ]; then NASSEG_EXTRA_FLAGS="" python nas-segm-pytorch_amd/build.py > /dev/null
    else NASSEG_EXTRA_FLAGS="$f" python nas-segm-pytorch_amd/build.py > /dev/null; fi
    echo "=== $f" | tee -a $SUM
    [ "$KB" != "-" ] && python tools/$KB 2>&1 | tail -20 | tee -a $SUM
    for i in 1 2; do
      bench "flag_$(echo "$f" | tr -c 'a-zA-Z0-9' '_')_$i" --steps 20 --warmup 5 --no-roofline
      # FLAGS_ALSO="--workload cvpr321 --graph 2;--workload task0": further bench lines per variant
      IFS=';' read -ra more <<< "${FLAGS_ALSO:-}"
      for m in "${more[@]}"; do [ -n "$m" ] && bench "flag_$(echo "$f $m" | tr -c 'a-zA-Z0-9' '_')_$i" --steps 20 --warmup 5 --no-roofline $m; done
    done
  done ;;
lib)
  V=$1
  for rep in 1 2; do for lib in base $V; do
    if [ $lib = base ]; then unset NASSEG_LIB; else export NASSEG_LIB=$PWD/tools/build/variants/lib_$lib.so; fi
    bench "${lib}_headline_$rep" --steps 8 --warmup 3 --no-roofline
    bench "${lib}_arch1_$rep" --workload arch1 --steps 8 --warmup 3 --no-roofline
    bench "${lib}_cvpr321_g2_$rep" --workload cvpr321 --graph 2 --steps 20 --warmup 3 --no-roofline
  done; done ;;
kbench)
  S=$1; shift
  timeout 900 python tools/$S "$@" 2>&1 | tee $OUT/$(basename $S .py).txt | tail -60 ;;
pmc)
  S=$1; shift; i=0
  ABS=$PWD/$OUT
  for set in "$@"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ABS/p$i -o run -- python $OLDPWD/tools/$S > $ABS/p$i.log 2>&1)
    echo "pass $i ($set) exit $?" | tee -a $SUM
  done
  python tools/pmc_table.py $OUT | tee $OUT/pmc.txt | head -80 ;;
*)
  echo "unknown sub-command $CMD"; sed -n 2,21p "$0"; exit 2 ;;
esac
