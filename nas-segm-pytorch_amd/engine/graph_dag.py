"""Real dependencies for a step recorded into a hipGraph.

``engine/graphed.py`` records forward + loss + backward (+ optimisers) of a candidate from ONE stream, so the graph
it gets is a line: every kernel waits for the one recorded before it.  The network is not a line - the five ops of a
ContextualCell read one input, the two cells of a MergeCell share nothing until their sum, the aux heads hang off
finished maps (reference src/nn/micro_decoders.py:54-139,198-251), and no weight gradient is read before the
optimiser - and on the 11x11 ... 81x81 maps of the CVPR cells a dependent kernel costs >= 4.6 us whatever it does
(~900 of them per 321x321 step, tools/graph_branches.hip measures the forms).  While the step is recorded a
``Recorder`` sits behind ``lib.call`` / ``ptr``: for every entry point it notes which recorded nodes the call added
and the address ranges behind its pointer arguments - read (``const`` in include/nasseg.h) or written.  From those
``schedule`` derives every read-after-write / write-after-read / write-after-write pair ON ADDRESSES (memory the
caching allocator handed out twice during the capture is ordered like any other hazard), places the calls in a few
LANES - lines of the graph that the runtime maps to streams - and ``nasseg_graph_rewire`` replaces the line's edges.

What is not known is ordered conservatively: nodes recorded outside ``lib.call`` (ATen kernels, copies), calls with
pointer tables that nobody annotated (``lib.annotate``), and pointers that did not come from ``ptr()`` right before
the call make their call a BARRIER - it waits for everything recorded before it and everything after waits for it.

Why lanes and not the bare DAG: a dependency between two streams costs this runtime about as much as a small
kernel (tools/graph_branches.hip: groups of one kernel feeding five parallel ones replay SLOWER than the line), so
a call changes lane only when the model below says it pays: list scheduling in recording order, a call goes to
the lane where it can start first, a dependency on another lane costs ``SYNC_US``.  Same kernels, same arguments,
same memory, no float atomics anywhere: results are bit-identical to the line
(tests/test_hip_engine.py::test_lanes_replay_equals_the_line).
"""
import bisect
import ctypes
import logging
import os

from .._lib import NassegError, current_stream, lib, pointer_access

logger = logging.getLogger(__name__)

# NASSEG_GRAPH_LANES: lanes a recorded step is laid out in (1 = the line as recorded)
LANES = int(os.environ.get("NASSEG_GRAPH_LANES", "4"))
# NASSEG_GRAPH_MODE: "stages" (default: stages of independent lanes, every (stage, lane) a line graph, ordered with
# events) | "rewire" (ONE graph whose edges are the real dependencies laid out in lanes - kept for A/B: on this
# runtime a graph with branches replays from per-node commands and pays ~10 us per crossing, tools/graph_branches.hip)
MODE = os.environ.get("NASSEG_GRAPH_MODE", "stages")
# cost model of the list scheduler, in microseconds: a dependency that crosses lanes, the floor of any launch
SYNC_US = float(os.environ.get("NASSEG_GRAPH_SYNC_US", "8"))
MIN_US = 4.7
_BYTES_PER_US = 2.5e6  # (2.5 TB/s: what mid-sized launches of this library reach)


class Unit(object):
    """the nodes [first, last) one call added, what it read and wrote"""
    __slots__ = ("name", "first", "last", "reads", "writes", "barrier", "why", "us")

    def __init__(self, name, first, last, reads, writes, barrier, why=None):
        self.name, self.first, self.last = name, first, last
        self.reads, self.writes, self.barrier, self.why = reads, writes, barrier, why
        nbytes = sum(hi - lo for lo, hi in reads) + sum(hi - lo for lo, hi in writes)
        self.us = max(MIN_US, nbytes / _BYTES_PER_US) * max(1, last - first)


class Recorder(object):
    """installed as ``lib.recorder`` while a stream is capturing (``with Recorder() as rec: ...`` inside the capture)"""

    def __init__(self):
        self.access = pointer_access()
        self.fresh = {}
        self.units = []
        self.annotation = None
        # Every storage a recorded launch touches stays allocated until the capture ends: a block the caching
        # allocator hands out twice during the capture ties its second user to its first (write-after-read /
        # write-after-write on the same addresses) - in recording order, i.e. it would put the line back.  Storages,
        # not tensors: autograd adopts a gradient tensor as param.grad only while nobody else references the TENSOR.
        # The price is the pool: it holds the sum of the step's temporaries instead of their peak.
        self.keep = []
        self._count = lib._fn["nasseg_graph_capture_nodes"]

    def __enter__(self):
        lib.load()
        if lib.recorder is not None:
            raise NassegError("graph_dag: a step is already being recorded")
        lib.recorder = self
        return self

    def __exit__(self, exc_type, exc, tb):
        lib.recorder = None
        if exc_type is not None:
            self.keep = []
        return False

    def release(self):
        """after the capture has ended: the storages go back to the graph's pool"""
        self.keep = []

    # -- hooks (called from _lib.ptr / _lib._Library.call) -----------------------------------------
    def note(self, t):
        """ptr(t): the address, and - for the call that follows - the range of the storage behind it"""
        addr = t.data_ptr()
        st = t.untyped_storage()
        lo = st.data_ptr()
        self.fresh[addr] = (lo, lo + st.nbytes())
        self.keep.append(st)
        return addr

    def annotate(self, reads=(), writes=()):
        """the next call's pointer tables: the tensors its kernels read / write behind them"""
        self.annotation = ([self._range(t) for t in reads if t is not None],
                           [self._range(t) for t in writes if t is not None])

    @staticmethod
    def _range(t):
        if isinstance(t, tuple):  # (address, bytes)
            return (int(t[0]), int(t[0]) + int(t[1]))
        st = t.untyped_storage()
        lo = st.data_ptr()
        return (lo, lo + st.nbytes())

    def nodes(self):
        """nodes the capture has recorded so far"""
        return self._nodes()

    def _nodes(self):
        n = self._count(current_stream())
        if n < 0:
            raise NassegError("graph_dag: the stream stopped capturing ({})".format(lib.last_error()))
        return n

    def call(self, name, args, fn):
        first = self._nodes()
        rc = fn(*args)
        if rc < 0:
            return rc
        last = self._nodes()
        fresh, self.fresh = self.fresh, {}
        note, self.annotation = self.annotation, None
        reads, writes, why = [], [], None
        if note is not None:
            reads, writes = note
        else:
            for i, kind in self.access.get(name, ((0, "t"),)):
                a = args[i]
                if a is None or (isinstance(a, int) and a == 0):
                    continue
                rng = fresh.get(a) if isinstance(a, int) else None
                if kind == "t" or rng is None:
                    why = "argument {} ({})".format(i, "a pointer table" if kind == "t" or not isinstance(a, int)
                                                    else "an address ptr() did not hand out for this call")
                    break
                (reads if kind == "r" else writes).append(rng)
        if last > first:
            self.units.append(Unit(name, first, last, reads, writes, why is not None, why))
        return rc


def annotate(reads=(), writes=()):
    """functional.py's pointer-table calls (nasseg_*_many, nasseg_wgrad_finalize_many, nasseg_pack_weights) say what
    their launches touch; a no-op unless a step is being recorded"""
    rec = lib.recorder
    if rec is not None:
        rec.annotate(reads, writes)


class _Ranges(object):
    """address ranges -> (last writer, readers since): a sorted list of boundaries, segments split on demand"""

    def __init__(self):
        self.bounds = [0, 1 << 62]
        self.state = [[-1, []]]  # state[i] covers [bounds[i], bounds[i+1])

    def _split(self, x):
        i = bisect.bisect_right(self.bounds, x) - 1
        if self.bounds[i] != x:
            self.bounds.insert(i + 1, x)
            w, r = self.state[i]
            self.state.insert(i + 1, [w, list(r)])
            i += 1
        return i

    def segments(self, lo, hi):
        a = self._split(lo)
        b = self._split(hi)
        return self.state[a:b]


def dependencies(units):
    """deps[u] = the earlier units u must follow (read-after-write, write-after-read, write-after-write on address
    ranges; a barrier follows everything since the barrier before it, and everything after follows it)"""
    ranges = _Ranges()
    deps = []
    last_barrier = -1
    for u, unit in enumerate(units):
        d = set()
        if unit.barrier:
            d.update(range(last_barrier + 1, u))  # (transitively: everything recorded so far)
            if last_barrier >= 0:
                d.add(last_barrier)
            last_barrier = u
            ranges = _Ranges()  # (every later hazard with an earlier unit is covered by the barrier)
        else:
            if last_barrier >= 0:
                d.add(last_barrier)
            for lo, hi in unit.reads:
                for seg in ranges.segments(lo, hi):
                    if seg[0] >= 0:
                        d.add(seg[0])
            for lo, hi in unit.writes:
                for seg in ranges.segments(lo, hi):
                    if seg[0] >= 0:
                        d.add(seg[0])
                    d.update(seg[1])
            # (a unit that reads and writes one range: record the write last)
            for lo, hi in unit.reads:
                for seg in ranges.segments(lo, hi):
                    if not seg[1] or seg[1][-1] != u:
                        seg[1].append(u)
            for lo, hi in unit.writes:
                for seg in ranges.segments(lo, hi):
                    seg[0] = u
                    seg[1] = []
            d.discard(u)
        deps.append(d)
    return deps


def fill_gaps(units, n_nodes):
    """units covering every node: the nodes no call accounts for (ATen kernels, copies recorded between the calls)
    become barriers"""
    out, at = [], 0
    for unit in sorted(units, key=lambda x: x.first):
        if unit.first < at:
            raise NassegError("graph_dag: calls overlap in the recorded graph")
        if unit.first > at:
            out.append(Unit("(recorded outside lib.call)", at, unit.first, [], [], True, "not a nasseg call"))
        out.append(unit)
        at = unit.last
    if at < n_nodes:
        out.append(Unit("(recorded outside lib.call)", at, n_nodes, [], [], True, "not a nasseg call"))
    return out


def schedule(units, deps, lanes=None, sync_us=None, durations=None):
    """lane[u] for every unit and the edges (node indices) of the laid-out graph.

    List scheduling in recording order with a cost for crossing lanes: unit u goes to the lane where it can start
    first - ready(l) = max over its dependencies d of finish[d] (+ sync_us if d sits in another lane and lane l has
    not waited for it already), start = max(ready(l), when lane l is free).  Ties go to the lane of the dependency
    that finishes last (a chain stays in its lane), then to the lowest lane.  Within a lane units keep their
    recording order, so every lane is a line and an edge only ever points forward."""
    L = max(1, int(LANES if lanes is None else lanes))
    sync = SYNC_US if sync_us is None else float(sync_us)
    n = len(units)
    us = durations if durations is not None else [x.us for x in units]
    lane = [0] * n
    finish = [0.0] * n
    free = [0.0] * L
    tail = [-1] * L                       # last unit of each lane
    seen = [[-1] * L for _ in range(L)]   # seen[l][m]: last unit of lane m that lane l has (transitively) waited for
    seen_at = [None] * n                  # snapshot of seen[lane[u]] when u was placed
    edges = []
    for u in range(n):
        unit = units[u]
        d = sorted(deps[u])
        if unit.barrier or L == 1:
            best = 0
        else:
            best, best_key = 0, None
            last_dep = max(d, key=lambda x: finish[x]) if d else -1
            for l in range(L):
                ready = 0.0
                for x in d:
                    cross = lane[x] != l and seen[l][lane[x]] < x
                    ready = max(ready, finish[x] + (sync if cross else 0.0))
                key = (max(ready, free[l]), 0 if (last_dep >= 0 and lane[last_dep] == l) else 1, l)
                if best_key is None or key < best_key:
                    best, best_key = l, key
        l = lane[u] = best
        ready = free[l]
        if tail[l] >= 0:
            edges.append((units[tail[l]].last - 1, unit.first))
        for x in reversed(d):  # (latest first: one edge per lane covers the earlier ones)
            m = lane[x]
            if m == l or seen[l][m] >= x:
                ready = max(ready, finish[x])
                continue
            edges.append((units[x].last - 1, unit.first))
            ready = max(ready, finish[x] + sync)
            seen[l][m] = x
            for k in range(L):
                if seen_at[x][k] > seen[l][k]:
                    seen[l][k] = seen_at[x][k]
        seen[l][l] = u
        seen_at[u] = list(seen[l])
        for k in range(unit.first, unit.last - 1):
            edges.append((k, k + 1))
        finish[u] = ready + us[u]
        free[l] = finish[u]
        tail[l] = u
    return lane, edges, max(free) if n else 0.0


def verify(units, deps, lane, edges, n_nodes):
    """every dependency is implied by the edges (reachability over the laid-out graph) - a check of ``schedule``
    against ``dependencies``, cheap enough to run at every capture"""
    succ = [[] for _ in range(n_nodes)]
    for a, b in edges:
        if not 0 <= a < b < n_nodes:
            raise NassegError("graph_dag: edge {} -> {} of {} nodes".format(a, b, n_nodes))
        succ[a].append(b)
    # reach[v] = bitset of nodes v can reach (node indices grow along every edge: one pass from the end)
    reach = [0] * n_nodes
    for v in range(n_nodes - 1, -1, -1):
        r = 0
        for w in succ[v]:
            r |= reach[w] | (1 << w)
        reach[v] = r
    for u, d in enumerate(deps):
        for x in d:
            if not (reach[units[x].last - 1] >> units[u].first) & 1:
                raise NassegError("graph_dag: {} (unit {}) is not ordered after {} (unit {})".format(
                    units[u].name, u, units[x].name, x))


def lay_out(recorder, raw_graph, n_nodes, lanes=None, durations=None):
    """rewire the recorded graph (hipGraph_t handle ``raw_graph``, ``n_nodes`` nodes).  Returns a summary dict."""
    units = fill_gaps(recorder.units, n_nodes)
    deps = dependencies(units)
    lane, edges, model_us = schedule(units, deps, lanes=lanes, durations=durations_for(units, durations))
    verify(units, deps, lane, edges, n_nodes)
    flat = (ctypes.c_int * (2 * max(1, len(edges))))()
    for i, (a, b) in enumerate(edges):
        flat[2 * i], flat[2 * i + 1] = a, b
    lib.call("nasseg_graph_rewire", raw_graph, n_nodes, len(edges), flat)
    L = max(lane) + 1 if lane else 1
    barriers = [x for x in units if x.barrier]
    info = {"nodes": n_nodes, "units": len(units), "edges": len(edges), "lanes": L,
            "per_lane": [sum(1 for v in lane if v == l) for l in range(L)],
            "cross_edges": len(edges) - sum(max(0, x.last - x.first - 1) for x in units)
            - sum(max(0, c - 1) for c in [sum(1 for v in lane if v == l) for l in range(L)]),
            "barriers": len(barriers), "model_us": model_us,
            "line_us": sum(x.us for x in units),
            "barrier_names": sorted(set("{}: {}".format(x.name, x.why) for x in barriers))}
    logger.info("graph_dag: %s", info)
    _dump(info, units, deps, lane, edges)
    return info


# ---------------------------------------------------------------------------------------------------------------
# stages and lanes: every (stage, lane) a line graph of its own
# ---------------------------------------------------------------------------------------------------------------
# NASSEG_GRAPH_TRIALS=0: take the cost model's layout for LANES lanes unmeasured (default: time the candidates)
TRIALS = os.environ.get("NASSEG_GRAPH_TRIALS", "1") != "0"
# the order stages are cut in when layouts are not timed: "asap" | "recorded" (asap_order)
ORDER = os.environ.get("NASSEG_GRAPH_ORDER", "asap")
# cost of a stage with more than one lane (event record + waits on both sides), microseconds
FORK_US = float(os.environ.get("NASSEG_GRAPH_FORK_US", "30"))
_STAGE_WINDOW = 512  # longest stage the planner looks at, in units


def asap_order(units, deps, us):
    """the units sorted by the time they could start at the earliest (unlimited lanes, no cost for crossing them) -
    a topological order like the recording order, but one in which work that CAN overlap sits side by side: autograd
    runs the backward of the most expensive cell first and the encoder's backward last, although the deep half of the
    encoder's backward only waits for the small cells - in recording order no contiguous stage holds both."""
    start = [0.0] * len(units)
    end = [0.0] * len(units)
    for u in range(len(units)):
        t = 0.0
        for d in deps[u]:
            if end[d] > t:
                t = end[d]
        start[u] = t
        end[u] = t + us[u]
    return sorted(range(len(units)), key=lambda u: (start[u], u))


def plan_stages(units, deps, us, lanes=None, fork_us=None, order=None):
    """Cut the units into stages of connected components.

    A stage is a range [i, j) of ``order`` (a topological order of the units: the recording order, or asap_order);
    inside it, units tied by a dependency form a component, and components share nothing - they may run side by
    side, a component per lane at a time, with no synchronisation until the stage ends.  Its cost: the sum of its
    units when it is one component, else max(largest component, sum / lanes) + fork_us.  Dynamic programme over the
    cut points (best[j] = min over i of best[i] + cost(i, j)); a barrier is a stage of its own.  Cutting right
    behind a small shared producer (the 1x1 adapt conv four cells read) is what makes its consumers independent
    components of the NEXT stage - the programme finds that by itself.
    Returns ([(i, j)] covering range(len(units)) as POSITIONS in ``order``, modelled microseconds)."""
    L = max(1, int(LANES if lanes is None else lanes))
    fork = FORK_US if fork_us is None else float(fork_us)
    n = len(units)
    order = list(range(n)) if order is None else order
    pos_of = [0] * n
    for k, u in enumerate(order):
        pos_of[u] = k
    inf = float("inf")
    best = [inf] * (n + 1)
    cut = [0] * (n + 1)
    best[0] = 0.0
    dep_pos = [sorted(pos_of[d] for d in deps[order[k]]) for k in range(n)]
    for i in range(n):
        if best[i] == inf:
            continue
        base = best[i]
        if units[order[i]].barrier:
            c = base + us[order[i]]
            if c < best[i + 1]:
                best[i + 1], cut[i + 1] = c, i
            continue
        parent = {}
        weight = {}
        total = 0.0
        biggest = 0.0
        comps = 0
        for j in range(i, min(n, i + _STAGE_WINDOW)):
            if units[order[j]].barrier:
                break
            parent[j] = j
            w = us[order[j]]
            root = j
            weight[j] = w
            comps += 1
            for d in dep_pos[j]:
                if d < i:
                    continue
                r = d
                while parent[r] != r:
                    parent[r] = parent[parent[r]]
                    r = parent[r]
                if r != root:
                    parent[r] = root
                    weight[root] += weight[r]
                    comps -= 1
            total += w
            if weight[root] > biggest:
                biggest = weight[root]
            c = base + (total if comps == 1 or L == 1 else max(biggest, total / L) + fork)
            if c < best[j + 1]:
                best[j + 1], cut[j + 1] = c, i
    out = []
    j = n
    while j > 0:
        out.append((cut[j], j))
        j = cut[j]
    out.reverse()
    return out, best[n]


def assign_lanes(units, deps, us, stages, lanes=None, order=None):
    """lane[u] inside its stage: components, longest first, each to the lane with the least work so far (the longest
    lands in lane 0, the stream the step runs on); stage_of[u].  Both indexed by unit."""
    L = max(1, int(LANES if lanes is None else lanes))
    n = len(units)
    order = list(range(n)) if order is None else order
    lane = [0] * n
    stage_of = [0] * n
    for s, (i, j) in enumerate(stages):
        members = order[i:j]
        local = dict((u, k) for k, u in enumerate(members))
        parent = list(range(j - i))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for u in members:
            stage_of[u] = s
            for d in deps[u]:
                if d in local:
                    a, b = find(local[u]), find(local[d])
                    if a != b:
                        parent[a] = b
        comps = {}
        for u in members:
            comps.setdefault(find(local[u]), []).append(u)
        if len(comps) == 1 or L == 1:
            continue
        load = [0.0] * L
        for group in sorted(comps.values(), key=lambda m: (-sum(us[u] for u in m), min(m))):
            l = min(range(L), key=lambda k: (load[k], k))
            load[l] += sum(us[u] for u in group)
            for u in group:
                lane[u] = l
    return stage_of, lane


def verify_stages(units, deps, stage_of, lane):
    """every dependency stays inside a lane of its stage or points to an earlier stage"""
    for u, d in enumerate(deps):
        for x in d:
            # (inside a (stage, lane) line the nodes run in recording order: x < u holds for every dependency)
            if not (stage_of[x] < stage_of[u] or (stage_of[x] == stage_of[u] and lane[x] == lane[u] and x < u)):
                raise NassegError("graph_dag: {} (unit {}, stage {} lane {}) is not ordered after {} (unit {}, stage {} "
                                  "lane {})".format(units[u].name, u, stage_of[u], lane[u], units[x].name, x,
                                                    stage_of[x], lane[x]))


_SIDE_STREAMS = {}  # device index -> raw handles of this process's lane streams (created once, never destroyed)
_CANDIDATES = 10    # streams tried per device


def _overlap_probe(device):
    """-> overlaps(a, b): do launches on the torch streams a and b run side by side?  HIP maps streams to a few
    hardware queues (4 by default) and two streams on one queue execute one after the other - which streams share a
    queue is not exposed, so it is measured: two recorded lines of 150 tiny kernels each, replayed on a and b at the
    same time, take ~0.7 of the time of the two lines in turn on two queues (measured: 0.345 against 0.49 ms) and all of
    it on one."""
    import time

    import torch

    buf = torch.zeros(2, 256, device=device, dtype=torch.float32)
    graphs = []
    for k in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(150):
                lib.call("nasseg_fill", buf[k].data_ptr(), 256, 1.0, current_stream())
        graphs.append(g)

    def both(a, b):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            graphs[0].replay()
        with torch.cuda.stream(b):
            graphs[1].replay()
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    def overlaps(a, b):
        both(a, b)
        pair = min(both(a, b) for _ in range(3))
        alone = min(both(a, a) for _ in range(3))  # (one stream: the two lines in turn)
        PROBE_LOG.append(round(pair / alone, 3))
        return pair < 0.85 * alone

    return overlaps


PROBE_LOG = []  # pair / alone of every probe of this process (diagnostics)


def side_streams(count):
    """up to ``count`` non-blocking streams for lanes 1 ... (lane 0 is the stream the step runs on) that run side by
    side with the current stream and with each other (_overlap_probe) - fewer when the device's hardware queues do
    not allow more.  Found once per device and process."""
    import torch

    device = torch.cuda.current_device()
    have = _SIDE_STREAMS.get(device)
    if have is None:
        have = _SIDE_STREAMS[device] = []
        if os.environ.get("NASSEG_GRAPH_PROBE", "1") == "0":  # (A/B: the first streams created, unprobed)
            for _ in range(3):
                out = (ctypes.c_void_p * 1)()
                lib.call("nasseg_lane_stream_create", out)
                have.append(int(out[0]))
        else:
            overlaps = _overlap_probe(torch.device("cuda", device))
            main = torch.cuda.current_stream(device)
            chosen = []
            for _ in range(_CANDIDATES):
                if len(chosen) == 3:
                    break
                out = (ctypes.c_void_p * 1)()
                lib.call("nasseg_lane_stream_create", out)
                cand = torch.cuda.ExternalStream(int(out[0]), device=device)
                if overlaps(main, cand) and all(overlaps(c, cand) for c in chosen):
                    chosen.append(cand)
                    have.append(int(out[0]))
            logger.info("graph_dag: %d lane streams on queues of their own", len(have))
    return have[:count]


class Plan(object):
    """a recorded step cut into line graphs, and the launch list that replays it (nasseg_graph_run)"""

    def __init__(self, raw_graph, n_nodes, units, stage_of, lane):
        # parts: consecutive stages that use lane 0 only are ONE line graph; a stage with side lanes has one per lane
        part_of = {}
        sequence = []   # [(lane -> part)] per launch group
        current = None
        n_stage = max(stage_of) + 1 if stage_of else 0
        lanes_of = [set() for _ in range(n_stage)]
        for u in range(len(units)):
            lanes_of[stage_of[u]].add(lane[u])
        for s in range(n_stage):
            if lanes_of[s] == {0}:
                if current is None:
                    current = {0: len(part_of)}
                    part_of[(s, 0)] = current[0]
                    sequence.append(current)
                else:
                    part_of[(s, 0)] = current[0]
            else:
                current = None
                group = {}
                for l in sorted(lanes_of[s]):
                    group[l] = part_of[(s, l)] = max(part_of.values(), default=-1) + 1
                sequence.append(group)
        n_parts = max(part_of.values(), default=-1) + 1
        node_part = (ctypes.c_int * n_nodes)()
        for u, unit in enumerate(units):
            p = part_of[(stage_of[u], lane[u])]
            for k in range(unit.first, unit.last):
                node_part[k] = p
        execs = (ctypes.c_void_p * n_parts)()
        lib.call("nasseg_graph_split", raw_graph, n_nodes, node_part, n_parts, execs)
        self.execs = [int(e or 0) for e in execs]
        self.events = []
        n_side = max((max(g) for g in sequence), default=0)
        side = side_streams(n_side)
        ops = []
        for group in sequence:
            if list(group) == [0]:
                ops.append((0, self.execs[group[0]], 0))
                continue
            fork = self._event()
            ops.append((1, fork, 0))
            joins = []
            for l in sorted(group):
                if l == 0:
                    continue
                st = side[l - 1]
                done = self._event()
                ops += [(2, fork, st), (0, self.execs[group[l]], st), (1, done, st)]
                joins.append(done)
            if 0 in group:
                ops.append((0, self.execs[group[0]], 0))
            ops += [(2, done, 0) for done in joins]
        self.n_ops = len(ops)
        self.ops = (ctypes.c_int64 * (3 * max(1, len(ops))))()
        for i, op in enumerate(ops):
            self.ops[3 * i], self.ops[3 * i + 1], self.ops[3 * i + 2] = op
        self.n_parts = n_parts
        self.n_groups = len(sequence)
        self.n_forks = sum(1 for g in sequence if list(g) != [0])

    def _event(self):
        out = (ctypes.c_void_p * 1)()
        lib.call("nasseg_lane_event_create", out)
        self.events.append(int(out[0]))
        return self.events[-1]

    def run(self):
        lib.call("nasseg_graph_run", self.n_ops, self.ops, current_stream())

    def close(self):
        execs, self.execs = self.execs, []
        events, self.events = self.events, []
        for e in execs:
            if e:
                lib.call("nasseg_graph_exec_destroy", e)
        for ev in events:
            lib.call("nasseg_lane_destroy", None, ev)

    def __del__(self):
        try:
            self.close()
        except Exception:  # (interpreter shutdown)
            pass


def lay_out_stages(recorder, raw_graph, n_nodes, lanes=None, durations=None, trial=None):
    """-> (Plan | None, summary): the recorded step as stages of independent lanes (None: the line as recorded is
    best, or the graph holds nodes that cannot be re-created - the caller replays the line).

    trial(run) -> seconds per replay (the caller's clock around a few replays; it puts back whatever they change):
    the cost model knows nothing of kernels that fill the GPU by themselves, of the runtime's queues, of what a
    fork costs today - so the few layouts worth trying (lane counts, two prices for a fork) are built and TIMED on
    the recorded step itself, the line included, and the fastest one is kept.  Without ``trial`` the model's choice
    for ``lanes`` is taken unmeasured."""
    units = fill_gaps(recorder.units, n_nodes)
    deps = dependencies(units)
    us = durations_for(units, durations)
    lanes = max(1, int(LANES if lanes is None else lanes))
    lanes = min(lanes, 1 + len(side_streams(lanes - 1)))
    info = {"mode": "stages", "probe": list(PROBE_LOG), "nodes": n_nodes, "units": len(units), "line_us": round(sum(us), 1),
            "barrier_units": sum(1 for x in units if x.barrier),
            "barriers": sorted(set("{}: {}".format(x.name, x.why) for x in units if x.barrier)),
            "measured_durations": durations is not None}
    orders = {"recorded": None, "asap": asap_order(units, deps, us)}
    if trial is None:
        candidates = [(lanes, FORK_US, ORDER)]
    else:
        candidates = [(L, f, o) for o in ("asap", "recorded") for L in range(2, lanes + 1) for f in (FORK_US, 4 * FORK_US)]
    best = None  # (seconds or None, plan, description)
    tried = []
    seen = set()
    for L, fork, oname in candidates:
        stages, model_us = plan_stages(units, deps, us, lanes=L, fork_us=fork, order=orders[oname])
        stage_of, lane = assign_lanes(units, deps, us, stages, lanes=L, order=orders[oname])
        verify_stages(units, deps, stage_of, lane)
        if not any(lane) or (tuple(stage_of), tuple(lane)) in seen:
            continue
        seen.add((tuple(stage_of), tuple(lane)))
        try:
            plan = Plan(raw_graph, n_nodes, units, stage_of, lane)
        except NassegError as e:
            if "neither a kernel nor a memset" not in str(e):
                raise
            info["unsupported"] = str(e)
            break
        desc = {"lanes": L, "fork_us": fork, "order": oname, "model_us": round(model_us, 1), "parts": plan.n_parts,
                "launches": plan.n_groups, "forks": plan.n_forks, "side_units": sum(1 for v in lane if v)}
        seconds = trial(plan.run) if trial is not None else None
        if seconds is not None:
            desc["ms"] = round(1e3 * seconds, 3)
        tried.append(desc)
        if best is None or (seconds is not None and seconds < best[0]):
            if best is not None:
                best[1].close()
            best = (seconds, plan, desc)
            _dump(dict(info, **desc), units, deps, lane, [], stage_of, us)
        else:
            plan.close()
    info["tried"] = tried
    if best is None:
        return None, info
    info.update(best[2])
    logger.info("graph_dag: %s", info)
    return best[1], info


def durations_for(units, measured):
    """microseconds per unit: measured per call (``measured`` = [(name, us)] of every lib.call of a warm-up pass) -
    the k-th unit of an entry point takes the k-th measurement of that entry point (the warm-up may launch the
    grouped weight gradients at other moments and in other group sizes than the recorded pass: an entry point with
    another number of calls gets its measured TOTAL spread over its units) - else the byte model"""
    if not measured:
        return [x.us for x in units]
    by_name = {}
    for name, us in measured:
        by_name.setdefault(name, []).append(us)
    count = {}
    for x in units:
        count[x.name] = count.get(x.name, 0) + 1
    seen = {}
    out = []
    for x in units:
        got = by_name.get(x.name)
        if not got:
            out.append(x.us)
            continue
        k = seen.get(x.name, 0)
        seen[x.name] = k + 1
        us = got[k] if len(got) == count[x.name] else sum(got) / count[x.name]
        out.append(max(MIN_US, us))
    return out


def _dump(info, units, deps, lane, edges, stage_of=None, us=None):
    dump = os.environ.get("NASSEG_GRAPH_DUMP")
    if dump:  # (tools/dag_report.py reads it: what serialises a recorded step)
        import json

        with open(dump, "w") as f:
            json.dump({"info": info, "lane": lane, "edges": edges, "stage_of": stage_of,
                       "units": [{"name": x.name, "first": x.first, "last": x.last, "reads": x.reads,
                                  "writes": x.writes, "barrier": x.barrier, "why": x.why,
                                  "us": x.us if us is None else us[i], "deps": sorted(d)}
                                 for i, (x, d) in enumerate(zip(units, deps))]}, f)
