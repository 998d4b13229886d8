#!/bin/bash
# Copy the evidence of the last `tools/gpu_check.sh all; tools/gpu_pmc.sh; tools/gpu_pmc_mfma.sh` run
# (merged into gpurun_out/ by gpurun) into profiles/ under this round's names.  Run in the build
# container after the gpurun call:  bash tools/collect_profiles.sh r02
set -eu
cd "$(dirname "$0")/.."
R="${1:-r02}"
G=gpurun_out
P=profiles
cp $G/bench.json $P/${R}_bench_headline.json
grep -v "amdgpu.ids" $G/bench.err > $P/${R}_bench_headline_breakdown.txt || true
cp $G/env.log $P/${R}_env.txt
for n in graph0 graph1 graph2 headline_bf16 arch1_g0 arch1_g2 cvpr321_g0 cvpr321_g2 search713_g0 search713_g2 depth480_g0 \
         depth480_bf16 depth480_bf16_g2 task0_auto task0_g0 teacher; do
  [ -f $G/bench_$n.json ] && cp $G/bench_$n.json $P/${R}_bench_$n.json
done
cp $G/prof/run_kernel_stats.csv $P/${R}_kernel_stats.csv
python tools/prof_summary.py $G/prof/run_kernel_stats.csv > $P/${R}_kernel_families.txt
cp $G/pmc_summary.txt $P/pmc_fetch_write_latest.txt
cp $G/pmc_summary.txt $P/${R}_pmc_fetch_write.txt
cp $G/pmc_mfma_summary.txt $P/${R}_pmc_mfma.txt
grep -E "exit|passed|failed" $G/summary.log > $P/${R}_gpu_tests.txt || true
ls -la $P | tail -n +1 | wc -l
