"""rocprofv3 kernel trace of a bench run (tools/gpu.sh trace OUT ...): pure kernel durations per kernel family and the
gaps between consecutive kernels, over the last third of the trace (the timed steps).
    python tools/trace_gaps.py gpurun_out/<outdir>"""
import collections
import csv
import glob
import re
import sys


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", name)


def main(out):
    files = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)
    if not files:
        print("no kernel trace under", out)
        return
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    n = len(rows)
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    lo = n - n // 3  # (2 warm-up + 4 timed steps of identical launches: the last two steps)
    busy = sum(ends[i] - starts[i] for i in range(lo, n))
    span = ends[-1] - starts[lo]
    gaps = [starts[i + 1] - ends[i] for i in range(lo, n - 1)]
    pos = [g for g in gaps if g > 0]
    print("kernels in window", n - lo, "span %.3f ms" % (span / 1e6), "busy %.3f ms (%.1f %%)" % (busy / 1e6, 100.0 * busy / span))
    print("gaps > 0: n=%d sum %.3f ms, mean %.2f us, median %.2f us; overlaps (negative gaps): %d" % (
        len(pos), sum(pos) / 1e6, sum(pos) / max(len(pos), 1) / 1e3, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0,
        sum(1 for g in gaps if g < 0)))
    for g, a, b in sorted(((g, names[lo + i], names[lo + i + 1]) for i, g in enumerate(gaps)), reverse=True)[:12]:
        print("  gap %.1f us after %s before %s" % (g / 1e3, short(a)[:50], short(b)[:50]))
    fam = collections.defaultdict(lambda: [0, 0])
    for i in range(lo, n):
        k = re.split(r"[<(]", re.sub(r"^void ", "", short(names[i])))[0]
        fam[k][0] += 1
        fam[k][1] += ends[i] - starts[i]
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-40s n=%4d %8.3f ms %6.1f us avg" % (k[:40], c, t / 1e6, t / c / 1e3))
    aten = sum(c for k, (c, _) in fam.items() if k.startswith("at::") or "rocclr" in k)
    print("ATen / runtime copy kernels in window:", aten)


if __name__ == "__main__":
    main(sys.argv[1])
