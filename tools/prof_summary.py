"""Aggregate a rocprofv3 --kernel-trace --stats CSV by kernel family (template
instantiations of one __global__ function summed): python tools/prof_summary.py <csv>"""
import collections
import csv
import re
import sys


def families(path):
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"^void ", "", name)
        base = re.split(r"[<(]", name)[0]
        fam[base][0] += int(r["Calls"])
        fam[base][1] += float(r["TotalDurationNs"])
    return sorted(fam.items(), key=lambda kv: -kv[1][1])


if __name__ == "__main__":
    rows = families(sys.argv[1])
    total = sum(t for _, (_, t) in rows)
    print("{:34s} {:>7s} {:>11s} {:>9s} {:>6s}".format("kernel family", "calls", "total ms", "avg us", "%"))
    for k, (n, t) in rows:
        print("{:34s} {:7d} {:11.3f} {:9.1f} {:6.1f}".format(k[:34], n, t / 1e6, t / n / 1e3, 100 * t / total))
