"""How much of a replayed step runs with two or more lanes busy - measured WITHOUT a tracer (rocprofv3 --kernel-trace keeps
the kernels of different streams from overlapping at all: profiles/r06_kernel_trace_cvpr321_replayed.txt).

The step's plan (engine/graph_dag.Plan: line graphs launched on the main stream and on side streams, ordered by events)
is replayed op by op with a HIP event recorded on the op's stream before and after every line graph; the intervals
[start, end] of the line graphs give, per replay: the span, the share of it with >= 1 / >= 2 / >= 3 lanes busy, and the
sum of the intervals over the span (lanes busy on average).  A line graph is a chain of back-to-back kernels (96-97 % of a
line's span is kernel time: profiles/r06_replay_probe.txt), so "lane busy" is "a kernel of that lane in flight" to that
precision.  The extra events cost a replay ~2 % (printed).

    python tools/lane_timeline.py [workload] [replays]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cvpr321"
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    import nas_segm_amd  # noqa: F401
    from nas_segm_amd import functional as F
    from nas_segm_amd.engine.graphed import GraphedSegmenterStep

    device = torch.device("cuda", 0)
    wl = bench.WORKLOADS[workload]
    segmenter, net = bench.build_model(device, workload)
    segmenter.train()
    optim_enc = torch.optim.SGD(net.encoder.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    optim_dec = torch.optim.Adam(net.decoder.parameters(), lr=3e-3, weight_decay=1e-5)
    image, mask = bench.synthetic_batch(wl[3], wl[4], wl[5], 0, device, wl[2])
    g = GraphedSegmenterStep(segmenter, image, mask, optim_enc, optim_dec, 255, 3.0, 3.0, -1, capture_optimisers=True)
    plan = g.plan
    lay = dict((k, v) for k, v in (g.layout or {}).items() if k not in ("tried", "probe"))
    print("workload", workload, "layout:", lay)
    if plan is None:
        print("the step replays as the line it was recorded as: nothing to overlap")
        return
    main_stream = torch.cuda.current_stream()
    ops = [(int(plan.ops[3 * i]), int(plan.ops[3 * i + 1]), int(plan.ops[3 * i + 2])) for i in range(plan.n_ops)]
    streams = {0: main_stream}
    for _, _, b in ops:
        if b and b not in streams:
            streams[b] = torch.cuda.ExternalStream(b, device=device)
    one = (ctypes.c_int64 * 3)()

    def run_op(op):
        one[0], one[1], one[2] = op
        F.lib.call("nasseg_graph_run", 1, one, F.current_stream())

    def plain():
        plan.run()

    def timed_ms(fn, n=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def instrumented():
        marks = []
        for op in ops:
            if op[0] == 0:
                st = streams[op[2]]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                run_op(op)
                b.record(st)
                marks.append((op[2], a, b))
            else:
                run_op(op)
        return marks

    print("replay: {:.3f} ms as the engine launches it, {:.3f} ms with the events of this tool".format(
        timed_ms(plain), timed_ms(lambda: instrumented())))
    results = []
    for _ in range(replays):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record(main_stream)
        marks = instrumented()
        torch.cuda.synchronize()
        iv = sorted((t0.elapsed_time(a), t0.elapsed_time(b), lane) for lane, a, b in marks)
        begin, end = iv[0][0], max(e for _, e, _ in iv)
        points = sorted([(s, 1) for s, _, _ in iv] + [(e, -1) for _, e, _ in iv])
        busy = [0.0, 0.0, 0.0, 0.0, 0.0]
        level, last = 0, begin
        for t, d in points:
            busy[min(level, 4)] += t - last
            level, last = level + d, t
        span = end - begin
        results.append((span, busy, sum(e - s for s, e, _ in iv), len(iv), len(set(l for _, _, l in iv))))
    results.sort(key=lambda r: r[0])
    span, busy, total, n_graphs, n_lanes = results[len(results) // 2]  # (the median replay)
    ge = lambda k: sum(busy[k:]) / span  # noqa: E731
    print("median of {} replays: span {:.3f} ms, {} line graphs on {} streams".format(replays, span, n_graphs, n_lanes))
    print("  lanes busy:  0: {:5.1f} %   >= 1: {:5.1f} %   >= 2: {:5.1f} %   >= 3: {:5.1f} %   >= 4: {:5.1f} %".format(
        100 * busy[0] / span, 100 * ge(1), 100 * ge(2), 100 * ge(3), 100 * ge(4)))
    print("  sum of the line graphs' intervals {:.3f} ms = {:.2f} lanes busy on average".format(total, total / span))


if __name__ == "__main__":
    main()
