"""What serialises a recorded step: reads the dump engine/graph_dag.py writes under NASSEG_GRAPH_DUMP=<file>.

  python tools/dag_report.py gpurun_out/.../dag.json [--sync US] [--lanes "1 2 3 4 8 64"]

Prints the stage planner's model for several lane counts in both orders, the one-graph scheduler's span, the critical
path (unlimited lanes, sync 0), the barriers, and - for the units on the critical path - which address range made each dependency (so that a
false dependency, e.g. two writers of disjoint parts of one storage, shows up by name).
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd.engine import graph_dag as G  # noqa: E402


def main():
    path = sys.argv[1]
    sync = float(sys.argv[sys.argv.index("--sync") + 1]) if "--sync" in sys.argv else G.SYNC_US
    lanes = [int(v) for v in (sys.argv[sys.argv.index("--lanes") + 1] if "--lanes" in sys.argv else "1 2 3 4 8 64").split()]
    d = json.load(open(path))
    units = [G.Unit(u["name"], u["first"], u["last"], [tuple(r) for r in u["reads"]], [tuple(w) for w in u["writes"]],
                    u["barrier"], u["why"]) for u in d["units"]]
    deps = G.dependencies(units)
    n = units[-1].last
    print("units", len(units), "nodes", n, "line", round(sum(u.us for u in units)), "us (model)")
    us_list = [u["us"] for u in d["units"]]  # (the durations the layout was made with: measured where available)
    for oname, order in (("recorded", None), ("asap", G.asap_order(units, deps, us_list))):
        for L in lanes:
            if L > 8:
                continue
            stages, model = G.plan_stages(units, deps, us_list, lanes=L, order=order)
            stage_of, lane = G.assign_lanes(units, deps, us_list, stages, lanes=L, order=order)
            G.verify_stages(units, deps, stage_of, lane)
            multi = sum(1 for s_ in range(len(stages)) if len(set(lane[u] for u in range(len(units)) if stage_of[u] == s_)) > 1)
            print("stages, {:8s} order, {} lanes: model {:8.0f} us of {:8.0f}; {} stages ({} with side lanes), {} units in side "
                  "lanes".format(oname, L, model, sum(us_list), len(stages), multi, sum(1 for v in lane if v)))
    for L in lanes:  # (the one-graph form, NASSEG_GRAPH_MODE=rewire)
        for s in (0.0, sync):
            lane, edges, us = G.schedule(units, deps, lanes=L, sync_us=s)
            print("rewire: lanes {:3d} sync {:4.1f}: span {:8.0f} us, per lane {}".format(L, s, us, [lane.count(l) for l in range(max(lane) + 1)]))
    # critical path
    fin = [0.0] * len(units)
    prev = [-1] * len(units)
    for u, unit in enumerate(units):
        best = -1
        for x in deps[u]:
            if best < 0 or fin[x] > fin[best]:
                best = x
        fin[u] = (fin[best] if best >= 0 else 0.0) + unit.us
        prev[u] = best
    end = max(range(len(units)), key=lambda u: fin[u])
    path_units = []
    while end >= 0:
        path_units.append(end)
        end = prev[end]
    path_units.reverse()
    print("critical path: {} units, {:.0f} us".format(len(path_units), fin[path_units[-1]]))
    print("barriers:", [(i, u.name, u.first, u.last) for i, u in enumerate(units) if u.barrier])
    from collections import Counter
    print("names on the critical path:", Counter(units[u].name for u in path_units).most_common(12))
    if "--path" in sys.argv:
        for a, b in zip(path_units, path_units[1:]):
            ua, ub = units[a], units[b]
            why = []
            for lo, hi in ub.reads:
                for wl, wh in ua.writes:
                    if lo < wh and wl < hi:
                        why.append("RAW {}B".format(min(hi, wh) - max(lo, wl)))
            for lo, hi in ub.writes:
                for wl, wh in ua.writes:
                    if lo < wh and wl < hi:
                        why.append("WAW {}B".format(min(hi, wh) - max(lo, wl)))
                for rl, rh in ua.reads:
                    if lo < rh and rl < hi:
                        why.append("WAR {}B".format(min(hi, rh) - max(lo, rl)))
            print("{:4d} {:34s} -> {:4d} {:34s} {:7.1f} us  {}".format(a, ua.name, b, ub.name, ub.us, " ".join(why[:3]) or ("barrier" if ua.barrier or ub.barrier else "?")))


if __name__ == "__main__":
    main()
