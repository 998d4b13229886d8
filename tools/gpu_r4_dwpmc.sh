#!/bin/bash
# counter passes over tools/kbench_dw5_one.py: where the cycles of the dilated 5x5 depthwise kernel go
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dwpmc
rm -rf $OUT; mkdir -p $OUT
ARGS="${*:-}"
i=0
SETS=${SETS:-all}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  if [ "$SETS" = mem ] && [ $i -lt 3 ]; then continue; fi
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- python $OLDPWD/tools/kbench_dw5_one.py $ARGS > $OUT/p$i.log 2>&1)
  echo "pass $i ($set) exit $?"
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/dwpmc/p*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dw_fwd_strip" not in r["Kernel_Name"]:
            continue
        t = tot[r["Counter_Name"]]
        t[0] += 1
        t[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(tot.items()):
    print("%-34s per launch %.4g" % (k, v / max(n, 1)))
PY
