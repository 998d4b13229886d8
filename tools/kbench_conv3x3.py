"""3x3 stride-1 convs of the CVPR cells: forward with statistics and backward-data (nasseg_conv_fwd), timing"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
lib.load()
DEV = "cuda:0"


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


for B, K, H, W, N, pad, dil in [(16, 64, 81, 81, 64, 1, 1), (16, 64, 81, 81, 64, 3, 3), (16, 64, 81, 81, 21, 1, 1),
                                (16, 64, 41, 41, 64, 1, 1), (16, 64, 21, 21, 64, 1, 1), (16, 64, 11, 11, 64, 3, 3),
                                (4, 64, 256, 512, 19, 1, 1), (4, 32, 128, 256, 32, 1, 1), (8, 48, 120, 160, 48, 1, 1),
                                (8, 64, 179, 179, 64, 3, 3), (8, 64, 30, 40, 64, 1, 1), (8, 64, 60, 80, 64, 1, 1),
                                (8, 64, 60, 80, 64, 3, 3), (8, 48, 90, 90, 48, 1, 1)]:
    x = torch.randn(B, H, W, K, device=DEV)
    w = torch.randn(N, K, 3, 3, device=DEV) * 0.05
    wp = torch.empty(9 * N * K, device=DEV)
    s = stream()
    lib.call("nasseg_conv_pack_weight", ptr(w), ptr(wp), N, K, 3, 3, 0, s)
    y = torch.empty(B, H, W, N, device=DEV)
    out = []
    if N % 4 == 0:
        rows = lib.query("nasseg_conv_fwd_stats_rows", B, H, W, N, K, 3, 3, 1, pad, dil)
        part = torch.empty((rows + 64) * 2 * N, device=DEV)
        out.append("with statistics ({} rows) {:7.1f} us".format(rows, timeit(lambda: lib.call(
            "nasseg_conv_fwd", ptr(x), K, ptr(wp), ptr(y), N, None, None, 0, None, None, 0, None, 0, B, H, W, K, H, W, N,
            3, 3, 1, pad, dil, 0, ptr(part), s))))
    out.append("plain {:7.1f} us".format(timeit(lambda: lib.call(
        "nasseg_conv_fwd", ptr(x), K, ptr(wp), ptr(y), N, None, None, 0, None, None, 0, None, 0, B, H, W, K, H, W, N, 3, 3,
        1, pad, dil, 0, None, s))))
    fl = 2.0 * B * H * W * N * K * 9
    print("{}x{}x{} {}->{} d{}: {}  ({:.1f} GFLOP)".format(B, H, W, K, N, dil, " | ".join(out), fl / 1e9), flush=True)
