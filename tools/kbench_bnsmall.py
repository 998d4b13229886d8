"""nasseg_bn_bwd_small against nasseg_bn_bwd_reduce + nasseg_bn_bwd_apply at the small maps of the CVPR cells"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
DEV = "cuda:0"


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, C, H, W in [(16, 64, 11, 11), (16, 64, 21, 21), (16, 64, 6, 6), (16, 21, 81, 81)[:0] or (16, 32, 21, 21), (8, 48, 23, 23)]:
    M = B * H * W
    dy, x, dx = (torch.randn(B, H, W, C, device=DEV) for _ in range(3))
    v = [torch.rand(C, device=DEV) + 0.5 for _ in range(4)]
    sums = torch.empty(2 * C, device=DEV)
    ws = torch.empty(lib.query("nasseg_colred_workspace", 1, M, C), device=DEV)
    s = stream()

    def three():
        lib.call("nasseg_bn_bwd_reduce", ptr(dy), C, ptr(x), C, M, C, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), 1,
                 ptr(sums), ptr(ws), s)
        lib.call("nasseg_bn_bwd_apply", ptr(dy), ptr(x), ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), ptr(sums), M, C, 1,
                 1, ptr(dx), s)

    def one():
        lib.call("nasseg_bn_bwd_small", ptr(dy), C, ptr(x), C, M, C, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), 1, 1,
                 ptr(sums), ptr(dx), C, s)

    def one_sums():
        lib.call("nasseg_bn_bwd_small", ptr(dy), C, ptr(x), C, M, C, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), 1, 1,
                 ptr(sums), None, C, s)

    print("M={:5d} C={:3d}: reduce+finalize+apply {:6.1f} us | small {:6.1f} us | small (sums only) {:6.1f} us".format(
        M, C, timeit(three), timeit(one), timeit(one_sums)), flush=True)
