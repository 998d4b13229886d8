"""Which launches run a nearly empty last round: reads a rocprofv3 --kernel-trace CSV and, per (kernel, grid), works out how
many workgroups a CU holds at once (LDS, registers, 32 waves) and how many rounds of 256 CUs x that the grid makes.

  python tools/rounds_report.py gpurun_out/prof/run_kernel_trace.csv [min_us]

A grid of 528 workgroups where 512 run at once takes two rounds for 3 % more work than one (conv3x3_lds_kernel on the
16 x 81 x 81 maps of the CVPR cells, round 6): printed are the launches whose last round is less than a third full and
that run at most four rounds, largest total time first."""
import csv
import re
import sys
from collections import defaultdict

CUS, LDS, WAVES = 256, 160 << 10, 32


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    rows = defaultdict(list)
    for r in csv.DictReader(open(path)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        nwg = grid // max(wg, 1)
        waves = (wg + 63) // 64
        regs = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
        per_simd = max(1, min(8, 512 // max(regs, 1)))
        occ = min(WAVES // waves, (per_simd * 4) // waves if waves <= per_simd * 4 else 0) or 1
        lds = int(r["LDS_Block_Size"])
        if lds:
            occ = max(1, min(occ, LDS // lds))
        name = re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = name[:name.index("(")] if "(" in name else name
        rows[(name, nwg, wg, occ)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = []
    for (name, nwg, wg, occ), us in rows.items():
        cap = CUS * occ
        rounds = nwg / cap
        last = rounds - int(rounds)
        avg = sum(us) / len(us)
        if avg >= min_us and 1 <= int(rounds) <= 4 and 0 < last < 0.34:
            # what the launch would take if the last round were folded into the others
            saved = avg * (1 - int(rounds) / (int(rounds) + 1)) * len(us)
            out.append((saved, name, nwg, wg, occ, rounds, avg, len(us)))
    out.sort(reverse=True)
    print("{:60s} {:>7s} {:>5s} {:>4s} {:>7s} {:>8s} {:>5s} {:>10s}".format("kernel", "wgs", "wg", "occ", "rounds", "avg us", "n", "at stake us"))
    for saved, name, nwg, wg, occ, rounds, avg, n in out[:40]:
        print("{:60s} {:7d} {:5d} {:4d} {:7.2f} {:8.1f} {:5d} {:10.0f}".format(name[:60], nwg, wg, occ, rounds, avg, n, saved))


if __name__ == "__main__":
    main()
