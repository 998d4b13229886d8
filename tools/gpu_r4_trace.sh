#!/bin/bash
# rocprofv3 kernel trace of a short headline run: pure kernel durations, and the gaps between consecutive kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r4trace}
shift || true
EXTRA="$*"   # e.g. --workload cvpr321 --graph 2
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --pmc 0 $EXTRA > $OUT/bench.json 2> $OUT/bench.err
cd $OLDPWD
python - <<'PY' $OUT
import csv, glob, sys, re, collections
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: the last 4/6 of the run by kernel count (2 warm-up + 4 timed of identical launches)
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
# take the last third of the trace (two steps)
lo = n - n // 3
busy = sum(ends[i] - starts[i] for i in range(lo, n))
span = ends[-1] - starts[lo]
gaps = [starts[i + 1] - ends[i] for i in range(lo, n - 1)]
pos = [g for g in gaps if g > 0]
print("kernels in window", n - lo, "span %.3f ms" % (span / 1e6), "busy %.3f ms (%.1f %%)" % (busy / 1e6, 100.0 * busy / span))
print("gaps > 0: n=%d sum %.3f ms, mean %.2f us, median %.2f us; overlaps (negative gaps): %d" % (
    len(pos), sum(pos) / 1e6, sum(pos) / max(len(pos), 1) / 1e3, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0, sum(1 for g in gaps if g < 0)))
big = sorted(((g, names[lo + i], names[lo + i + 1]) for i, g in enumerate(gaps)), reverse=True)[:12]
for g, a, b in big:
    print("  gap %.1f us after %s before %s" % (g / 1e3, re.sub(r"\(anonymous namespace\)::", "", a)[:50], re.sub(r"\(anonymous namespace\)::", "", b)[:50]))
fam = collections.defaultdict(lambda: [0, 0])
for i in range(lo, n):
    k = re.split(r"[<(]", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", names[i])))[0]
    fam[k][0] += 1; fam[k][1] += ends[i] - starts[i]
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-40s n=%4d %8.3f ms %6.1f us avg" % (k[:40], c, t / 1e6, t / c / 1e3))
PY
