#!/bin/bash
# nasseg_optim_step: tests, then torch foreach / torch fused / native per workload on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/optim; mkdir -p $OUT
python -m pytest tests/test_hip_optim.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v Warn | tail -25
python -m pytest tests/test_hip_engine.py -m gpu -q --tb=short -p no:cacheprovider -k "graphed_step" 2>&1 | tail -15
run() { python bench.py --no-cpu-baseline --no-roofline --pmc 0 "$@" 2>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f img/s %7.3f ms'%(d['value'],d['ms_per_step']))" || tail -5 $OUT/err.txt; }
for rep in 1 2; do
for mode in foreach fused native; do
  case $mode in
    foreach) export NASSEG_NATIVE_OPTIM=0; F=0;;
    fused) export NASSEG_NATIVE_OPTIM=0; F=1;;
    native) export NASSEG_NATIVE_OPTIM=1; F=0;;
  esac
  echo "$mode headline        $(run --steps 8 --warmup 3 --fused-optim $F)"
  echo "$mode cvpr321 g2      $(run --workload cvpr321 --graph 2 --steps 20 --warmup 3 --fused-optim $F)"
  echo "$mode cvpr321 g1      $(run --workload cvpr321 --graph 1 --steps 20 --warmup 3 --fused-optim $F)"
  echo "$mode search713 g2    $(run --workload search713 --graph 2 --steps 20 --warmup 3 --fused-optim $F)"
  echo "$mode arch1           $(run --workload arch1 --steps 8 --warmup 3 --fused-optim $F)"
  echo "$mode task0           $(run --workload task0 --steps 20 --warmup 3 --fused-optim $F)"
done; done | tee $OUT/ab.txt
