"""rocprofv3 kernel trace of a replayed bench run (tools/gpu.sh trace OUT ...): how many kernels are in flight.

Over the last `steps` steps of the trace (a step ends with its last optim_apply_kernel; without optimiser kernels the
last third of the trace): share of the time with 0 / 1 / >= 2 / >= 3 kernels running, the gaps between steps, and the
sum of kernel durations against the span (> 1 = kernels overlapped).
    python tools/trace_overlap.py gpurun_out/<outdir> [steps]"""
import csv
import glob
import re
import sys


def main(out, steps=3):
    files = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)
    if not files:
        print("no kernel trace under", out)
        return
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    n = len(rows)
    # step boundaries: the end of the last optim_apply_kernel of each step (consecutive ones belong to one step)
    marks = [i for i in range(n) if "optim_apply_kernel" in names[i] and (i + 1 == n or "optim_" not in names[i + 1])]
    if len(marks) > steps:
        lo = marks[-steps - 1] + 1
        hi = marks[-1] + 1
        bounds = marks[-steps - 1:]
    else:
        lo, hi, bounds = n - n // 3, n, []
    t0, t1 = min(starts[lo:hi]), max(ends[lo:hi])
    ev = []
    for i in range(lo, hi):
        ev.append((starts[i], 1))
        ev.append((ends[i], -1))
    ev.sort()
    level, last, hist = 0, t0, {}
    for t, d in ev:
        hist[level] = hist.get(level, 0) + (t - last)
        last = t
        level += d
    span = t1 - t0
    busy = sum(ends[i] - starts[i] for i in range(lo, hi))
    print("window: {} kernels, {} steps, span {:.3f} ms ({:.3f} ms per step), kernel time {:.3f} ms = {:.2f} x span".format(
        hi - lo, max(1, len(bounds) - 1), span / 1e6, span / 1e6 / max(1, len(bounds) - 1), busy / 1e6, busy / span))
    for k in sorted(hist):
        print("  {} in flight: {:6.2f} %".format(k, 100.0 * hist[k] / span))
    print("  >= 2 in flight: {:.1f} %   >= 3: {:.1f} %".format(
        100.0 * sum(v for k, v in hist.items() if k >= 2) / span, 100.0 * sum(v for k, v in hist.items() if k >= 3) / span))
    for b in bounds[:-1]:
        nxt = min(starts[b + 1:b + 40]) if b + 1 < n else None
        if nxt is not None:
            print("  gap between steps: {:.1f} us (after {} before {})".format(
                (nxt - ends[b]) / 1e3, re.sub(r"\(.*", "", names[b])[:30], re.sub(r"\(.*", "", names[b + 1])[:40]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
