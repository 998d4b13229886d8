#!/bin/bash
# HBM traffic counters for the bench workload: two separate rocprofv3 --pmc passes
# (FETCH_SIZE uses 3 of the 4 TCC slots, WRITE_SIZE 2 - they do not fit one pass).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o run -- \
     python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --graph 0 --secondary 0 > $OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
  find $OUT/pmc_$c -name "*.csv" | head
done
python - <<'PY'
import csv, glob, re, collections, os
out = os.environ.get("OUT", "gpurun_out")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s/**/*counter_collection*.csv" % c, recursive=True)
    fam = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            base = re.split(r"[<(]", name)[0]
            fam[base][0] += 1
            fam[base][1] += float(r["Counter_Value"])
    res[c] = fam
keys = sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]), key=lambda k: -(res["FETCH_SIZE"].get(k, [0, 0])[1]))
import hashlib
lib_hash = hashlib.sha256(open("nas-segm-pytorch_amd/libnasseg_hip.so", "rb").read()).hexdigest()[:16]
STEPS = 3  # (bench.py --steps 2 --warmup 1)
# HBM bytes of one step: every kernel of the run (2 x FETCH_SIZE: gfx950 counts half of wide reads) / steps
step_bytes = sum(2.0 * res["FETCH_SIZE"].get(k, [0, 0.0])[1] + res["WRITE_SIZE"].get(k, [0, 0.0])[1]
                 for k in keys) * 1024.0 / STEPS
with open("gpurun_out/pmc_summary.txt", "w") as fo:
    fo.write("# lib %s steps %d step_bytes %.0f  (bench.py uses this file only while the hash matches its "
             "libnasseg_hip.so)\n" % (lib_hash, STEPS, step_bytes))
    fo.write("kernel family, launches, FETCH_SIZE KB/launch (raw), WRITE_SIZE KB/launch (raw)\n")
    for k in keys[:40]:
        nf, vf = res["FETCH_SIZE"].get(k, [0, 0.0]); nw, vw = res["WRITE_SIZE"].get(k, [0, 0.0])
        fo.write("%s, %d, %.1f, %.1f\n" % (k, max(nf, nw), vf / max(nf, 1), vw / max(nw, 1)))
print(open("gpurun_out/pmc_summary.txt").read())
PY
