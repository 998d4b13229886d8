"""Micro-benchmark of the per-channel reductions (nasseg_bn_bwd_reduce, nasseg_bn_stats) at the shapes
of the headline step.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
DEV = "cuda:0"
CASES = [(131072, 32), (131072, 64), (131072, 192), (524288, 32), (524288, 64), (524288, 144), (2097152, 32),
         (2097152, 96), (32768, 64), (8192, 64)]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, C in CASES:
    g, z = torch.randn(M, C, device=DEV), torch.randn(M, C, device=DEV)
    v = lambda: torch.rand(C, device=DEV) + 0.5  # noqa: E731
    scale, shift, mean, invstd = v(), v(), v(), v()
    sums = torch.empty(2 * C, device=DEV)
    ws = torch.empty(lib.query("nasseg_colred_workspace", 1, M, C), device=DEV)
    s = stream()
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)

    def bwd():
        lib.call("nasseg_bn_bwd_reduce", ptr(g), C, ptr(z), C, M, C, ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 1, ptr(sums), ptr(ws), s)

    def stats():
        lib.call("nasseg_bn_stats", ptr(z), C, M, C, 1e-5, 0.1, ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 ptr(scale), ptr(shift), ptr(rm), ptr(rv), None, ptr(ws), s)

    tb, ts = timeit(bwd), timeit(stats)
    print("{:8d} x {:3d}: bn_bwd_reduce {:6.1f} us ({:5.0f} GB/s)  bn_stats {:6.1f} us ({:5.0f} GB/s)".format(
        M, C, tb, 8e-3 * M * C / tb, ts, 4e-3 * M * C / ts))
