"""Host logic of engine/graph_dag.py (no GPU): dependencies from address ranges, the stage / lane planner, the
checks that guard a layout, and the header contract they rest on (const = read, non-const = written).

Reference: the structure being exploited is src/nn/micro_decoders.py:54-139 - a ContextualCell's ops read one
input, a MergeCell's two cells share nothing until their sum."""
import os
import re
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd._lib import HEADER_PATH, NassegError, pointer_access  # noqa: E402
from nas_segm_amd.engine import graph_dag as G  # noqa: E402

U = G.Unit


def cell(n0, x, outs, tag):
    """five two-kernel ops reading x, each writing its own map; then a sum of the five"""
    units, n = [], n0
    for o in outs:
        units.append(U("op" + tag, n, n + 2, [x], [o], False))
        n += 2
    return units, n


def two_cells():
    a, b = (1000, 2000), (3000, 4000)              # adapt outputs
    oa = [(10000 + 100 * i, 10000 + 100 * i + 64) for i in range(5)]
    ob = [(20000 + 100 * i, 20000 + 100 * i + 64) for i in range(5)]
    sa, sb, out = (30000, 30100), (31000, 31100), (40000, 40100)
    units = [U("adapt_a", 0, 1, [(1, 2)], [a], False), U("adapt_b", 1, 2, [(1, 2)], [b], False)]
    n = 2
    ca, n = cell(n, a, oa, "a")
    units += ca
    units.append(U("sum_a", n, n + 1, oa, [sa], False)); n += 1
    cb, n = cell(n, b, ob, "b")
    units += cb
    units.append(U("sum_b", n, n + 1, ob, [sb], False)); n += 1
    units.append(U("merge", n, n + 1, [sa, sb], [out], False)); n += 1
    return units, n


def test_dependencies_are_the_hazards_on_address_ranges():
    A = (100, 200)
    units = [U("w", 0, 1, [], [A], False),               # 0 writes A
             U("r1", 1, 2, [A], [(300, 310)], False),     # 1 reads A
             U("r2", 2, 3, [(150, 160)], [(400, 410)], False),  # 2 reads a PART of A
             U("w2", 3, 4, [], [(190, 260)], False),      # 3 overwrites the end of A and beyond: after 0, 1 - not 2
             U("r3", 4, 5, [(250, 255)], [], False),      # 4 reads what only 3 wrote
             U("r4", 5, 6, [(100, 110)], [], False)]      # 5 reads the part of A that 3 left alone
    deps = G.dependencies(units)
    assert deps == [set(), {0}, {0}, {0, 1}, {3}, {0}]


def test_a_barrier_follows_everything_and_everything_follows_it():
    units = [U("a", 0, 1, [], [(0, 8)], False), U("b", 1, 2, [], [(8, 16)], False),
             U("aten", 2, 3, [], [], True, "not a nasseg call"), U("c", 3, 4, [], [(16, 24)], False)]
    deps = G.dependencies(units)
    assert deps[2] == {0, 1} and deps[3] == {2}
    stages, _ = G.plan_stages(units, deps, [10.0] * 4, lanes=3, fork_us=1.0)
    assert (2, 3) in stages  # (a stage of its own)


def test_unknown_nodes_become_barriers():
    units = G.fill_gaps([U("a", 1, 2, [], [(0, 8)], False), U("b", 4, 5, [], [(0, 8)], False)], 6)
    assert [(u.first, u.last, u.barrier) for u in units] == [(0, 1, True), (1, 2, False), (2, 4, True), (4, 5, False),
                                                             (5, 6, True)]
    with pytest.raises(NassegError):
        G.fill_gaps([U("a", 0, 3, [], [], False), U("b", 2, 4, [], [], False)], 4)


def test_two_cells_become_two_lanes_and_the_merge_a_later_stage():
    units, n = two_cells()
    deps = G.dependencies(units)
    us = [50.0 * (u.last - u.first) for u in units]
    stages, cost = G.plan_stages(units, deps, us, lanes=2, fork_us=20.0)
    stage_of, lane = G.assign_lanes(units, deps, us, stages, lanes=2)
    G.verify_stages(units, deps, stage_of, lane)
    by_name = dict((u.name, (stage_of[i], lane[i])) for i, u in enumerate(units))
    # the cells sit in one stage on different lanes; the merge comes in a later stage
    assert by_name["opa"][0] == by_name["opb"][0] and by_name["opa"][1] != by_name["opb"][1]
    assert by_name["sum_a"] == by_name["opa"] and by_name["sum_b"] == by_name["opb"]
    assert by_name["merge"][0] > by_name["sum_a"][0] and by_name["merge"][1] == 0
    assert cost < sum(us)
    # one lane: the line
    stages1, cost1 = G.plan_stages(units, deps, us, lanes=1)
    assert cost1 == sum(us)
    # a fork that costs more than it saves is not taken
    stages_x, cost_x = G.plan_stages(units, deps, us, lanes=2, fork_us=1e6)
    so, ln = G.assign_lanes(units, deps, us, stages_x, lanes=2)
    assert not any(ln) and cost_x == sum(us)


def test_five_ops_off_one_input_split_behind_their_producer():
    """the planner cuts right behind a small shared producer: its consumers are independent in the NEXT stage"""
    x = (0, 64)
    outs = [(1000 * (i + 1), 1000 * (i + 1) + 64) for i in range(5)]
    units = [U("producer", 0, 1, [(5000, 5010)], [x], False)]
    ops, n = cell(1, x, outs, "")
    units += ops
    deps = G.dependencies(units)
    us = [5.0] + [100.0] * 5
    stages, cost = G.plan_stages(units, deps, us, lanes=4, fork_us=10.0)
    stage_of, lane = G.assign_lanes(units, deps, us, stages, lanes=4)
    G.verify_stages(units, deps, stage_of, lane)
    assert stage_of[0] < stage_of[1] and len(set(lane[1:])) == 4
    assert cost == pytest.approx(5.0 + max(100.0, 500.0 / 4) + 10.0 + (0.0 if len(set(stage_of[1:])) == 1 else 1e9), abs=101.0)


def test_verify_stages_rejects_a_layout_that_breaks_a_dependency():
    units, n = two_cells()
    deps = G.dependencies(units)
    stage_of = [0] * len(units)
    lane = [0] * len(units)
    G.verify_stages(units, deps, stage_of, lane)  # (the line)
    lane[-1] = 1                                  # the merge beside what it reads
    with pytest.raises(NassegError):
        G.verify_stages(units, deps, stage_of, lane)


def test_rewire_schedule_orders_every_dependency():
    units, n = two_cells()
    deps = G.dependencies(units)
    for lanes in (1, 2, 3):
        lane, edges, span = G.schedule(units, deps, lanes=lanes, sync_us=3.0)
        G.verify(units, deps, lane, edges, n)
    lane, edges, _ = G.schedule(units, deps, lanes=2, sync_us=3.0)
    with pytest.raises(NassegError):
        G.verify(units, deps, lane, [e for e in edges if e[1] != units[-1].first], n)  # the merge's inputs cut off


def test_measured_durations_follow_the_entry_points():
    units = [U("nasseg_a", 0, 1, [], [], False), U("nasseg_b", 1, 2, [], [], False), U("nasseg_a", 2, 3, [], [], False),
             U("(recorded outside lib.call)", 3, 4, [], [], True, "not a nasseg call"),
             U("nasseg_many", 4, 9, [], [], False)]
    measured = [("nasseg_a", 11.0), ("nasseg_zero_nodes", 3.0), ("nasseg_b", 22.0), ("nasseg_many", 30.0),
                ("nasseg_a", 33.0), ("nasseg_many", 50.0)]
    us = G.durations_for(units, measured)
    assert us[0] == 11.0 and us[1] == 22.0 and us[2] == 33.0
    assert us[3] == units[3].us            # nothing measured: the byte model
    assert us[4] == 80.0                   # two measured launches, one recorded unit: their total
    assert G.durations_for(units, None) == [u.us for u in units]


def test_header_const_is_the_read_write_contract():
    """every pointer of every prototype is classified; the two entry points that use their input buffer as scratch
    declare it non-const; outputs are never const"""
    acc = pointer_access()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER_PATH).read(), flags=re.S)
    assert "nasseg_graph_split" in acc and "nasseg_conv_fwd" in acc
    assert dict(acc["nasseg_bn_finalize"])[0] == "w" and dict(acc["nasseg_rows_sum"])[0] == "w"
    assert dict(acc["nasseg_conv_fwd"])[3] == "w" and dict(acc["nasseg_conv_fwd"])[0] == "r"
    assert [k for _, k in acc["nasseg_pack_weights"]][:2] == ["t", "t"]
    for name, args in acc.items():
        proto = re.search(r"\b" + name + r"\s*\(([^)]*)\)", text).group(1)
        n_ptr = sum(1 for a in proto.split(",") if "*" in a and a[a.rindex("*") + 1:].strip() != "stream")
        assert n_ptr == len(args), name


class _FakeTensor(object):
    """what Recorder.note needs of a tensor: an address inside a storage"""

    class _Storage(object):
        def __init__(self, lo, n):
            self._lo, self._n = lo, n

        def data_ptr(self):
            return self._lo

        def nbytes(self):
            return self._n

    def __init__(self, lo, n, offset=0):
        self._st, self._addr = self._Storage(lo, n), lo + offset

    def data_ptr(self):
        return self._addr

    def untyped_storage(self):
        return self._st


def test_recorder_classifies_what_a_call_was_handed(monkeypatch):
    """What the Recorder makes of a call (no GPU: the node counter is faked): pointers that `ptr()` handed out for
    THIS call become read (const in include/nasseg.h) or written ranges - the whole storage behind a view -; an
    address from an earlier call, a pointer table nobody annotated or an entry point the header does not know make
    the call a barrier; an annotation replaces the header's view; a call that recorded no node leaves no unit."""
    rec = G.Recorder()
    count = [0]
    monkeypatch.setattr(rec, "_nodes", lambda: count[0])

    def launch(n_nodes):
        def fn(*args):
            count[0] += n_nodes
            return 0
        return fn

    x, y, sc = _FakeTensor(1000, 400), _FakeTensor(5000, 800, offset=64), _FakeTensor(9000, 64)
    # nasseg_affine_act(const x, const scale, const shift, const res, y, n, C, act, stream)
    args = (rec.note(x), rec.note(sc), None, None, rec.note(y), 100, 4, 0, 0)
    rec.call("nasseg_affine_act", args, launch(1))
    u = rec.units[-1]
    assert not u.barrier and (u.first, u.last) == (0, 1)
    assert sorted(u.reads) == [(1000, 1400), (9000, 9064)] and u.writes == [(5000, 5800)]  # (the view's whole storage)
    # the same addresses again WITHOUT ptr(): what they point to may have been freed and handed out again since
    rec.call("nasseg_affine_act", args, launch(1))
    assert rec.units[-1].barrier and "ptr() did not hand out" in rec.units[-1].why
    # a pointer table (nasseg_pack_weights: const float* const* w) is a barrier ...
    rec.call("nasseg_pack_weights", (2, object(), object(), object(), 0), launch(2))
    assert rec.units[-1].barrier and (rec.units[-1].first, rec.units[-1].last) == (2, 4)
    # ... unless the caller says what the launches behind it touch
    rec.annotate(reads=[x], writes=[y, (7000, 16)])
    rec.call("nasseg_pack_weights", (2, object(), object(), object(), 0), launch(1))
    u = rec.units[-1]
    assert not u.barrier and u.reads == [(1000, 1400)] and u.writes == [(5000, 5800), (7000, 7016)]
    assert rec.annotation is None
    # an entry point the header does not declare: a barrier
    rec.call("nasseg_not_in_the_header", (rec.note(x),), launch(1))
    assert rec.units[-1].barrier
    # a call that recorded nothing (a query, an empty launch) leaves no unit - and consumes the noted addresses
    n_units = len(rec.units)
    rec.call("nasseg_affine_act", (rec.note(x), None, None, None, rec.note(y), 0, 4, 0, 0), launch(0))
    assert len(rec.units) == n_units and rec.fresh == {}
    # a failing call reports its status and records nothing
    assert rec.call("nasseg_affine_act", args, lambda *a: -1) == -1 and len(rec.units) == n_units
    # every noted storage stays referenced until the capture ends (no block is handed out twice while recording)
    assert len(rec.keep) >= 5
    rec.release()
    assert rec.keep == []
