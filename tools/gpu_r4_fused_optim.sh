#!/bin/bash
# torch.optim foreach (default) against fused=True, per workload, on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/fusedopt; mkdir -p $OUT
run() { python bench.py --no-cpu-baseline --no-roofline --pmc 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f img/s %7.3f ms'%(d['value'],d['ms_per_step']))"; }
for rep in 1 2; do
for f in 0 1; do
  echo "fused=$f headline        $(run --steps 8 --warmup 3 --fused-optim $f)"
  echo "fused=$f cvpr321 g2      $(run --workload cvpr321 --graph 2 --steps 20 --warmup 3 --fused-optim $f)"
  echo "fused=$f search713 g2    $(run --workload search713 --graph 2 --steps 20 --warmup 3 --fused-optim $f)"
  echo "fused=$f arch1           $(run --workload arch1 --steps 8 --warmup 3 --fused-optim $f)"
done; done | tee $OUT/ab.txt
