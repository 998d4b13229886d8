"""per-launch averages of the counters collected by `tools/gpu.sh pmc OUT SCRIPT "set1" "set2" ...`
    python tools/pmc_table.py gpurun_out/<outdir> [kernel-name substring]"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/p*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:60]
        if only and only not in name:
            continue
        t = tot[name][r["Counter_Name"]]
        t[0] += 1
        t[1] += float(r["Counter_Value"])
for name, counters in sorted(tot.items()):
    print(name)
    for k, (n, v) in sorted(counters.items()):
        print("   %-34s per launch %.4g  (%d launches)" % (k, v / max(n, 1), n))
