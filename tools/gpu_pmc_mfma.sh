#!/bin/bash
# Matrix-core activity of the bench workload (north_star: "rocprof HBM GB/s and MFMA-busy"):
# one rocprofv3 --pmc pass with the SQ MFMA counters (SQ has 8 slots; TCC counters are collected
# separately by tools/gpu_pmc.sh) + a kernel trace for the launch durations, joined per dispatch
# and summarised per kernel family -> gpurun_out/pmc_mfma_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="${*:---steps 2 --warmup 1}"
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $OUT/pmc_mfma -o run -- \
   python $OLDPWD/bench.py $ARGS --no-cpu-baseline --no-roofline --graph 0 --secondary 0 > $OUT/pmc_mfma.log 2>&1)
echo "pmc_mfma exit $?"
python - <<'PY'
import collections, csv, glob, re
cnt = collections.defaultdict(dict)   # dispatch id -> {counter: value}
name = {}
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = r["Dispatch_Id"]
        cnt[d][r["Counter_Name"]] = cnt[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        name[d] = r["Kernel_Name"]
dur = {}
for f in glob.glob("gpurun_out/pmc_mfma/**/*kernel_trace*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in cnt.items():
    n = re.sub(r"\(anonymous namespace\)::", "", name[d])
    n = re.split(r"[<(]", re.sub(r"^void ", "", n))[0]
    fam[n]["launches"] += 1
    fam[n]["ns"] += dur.get(d, 0.0)
    for k, v in c.items():
        fam[n][k] += v
GHZ, SIMDS = 2.4, 1024.0  # (256 CUs x 4 SIMDs; nominal clock)
rows = sorted(fam.items(), key=lambda kv: -kv[1]["ns"])
with open("gpurun_out/pmc_mfma_summary.txt", "w") as fo:
    fo.write("# per kernel family: launches, total us (under the profiler), MFMA ops (f32, x512 flop each MOP unit raw), "
             "SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, mfma_busy = MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs), "
             "mfma_busy_sq = MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES)\n")
    fo.write("family, launches, total_us, MOPS_F32, MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, INSTS_VALU, mfma_busy, mfma_busy_sq\n")
    for n, c in rows[:40]:
        cyc = c["ns"] * GHZ
        fo.write("%s, %d, %.1f, %.4g, %.4g, %.4g, %.4g, %.4f, %.4f\n" % (
            n, c["launches"], c["ns"] / 1e3, c["SQ_INSTS_VALU_MFMA_MOPS_F32"], c["SQ_VALU_MFMA_BUSY_CYCLES"],
            c["SQ_BUSY_CYCLES"], c["SQ_INSTS_VALU"],
            c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS + 1e-9),
            c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CYCLES"] + 1e-9)))
print(open("gpurun_out/pmc_mfma_summary.txt").read())
PY
