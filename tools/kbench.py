"""Micro-benchmark of single C-ABI entry points at headline shapes (GPU box).
usage: python tools/kbench.py [conv|wgrad|dw|dwwgrad|red] ...   """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cl(*shape):
    return torch.randn(*shape, device=DEV).contiguous(memory_format=torch.channels_last)


CONV = [  # B, K, H, W, N, k, stride, pad, dil
    (4, 224, 256, 512, 64, 1, 1, 0, 1), (4, 64, 256, 512, 224, 1, 1, 0, 1), (4, 128, 256, 512, 64, 1, 1, 0, 1),
    (4, 24, 256, 512, 144, 1, 1, 0, 1), (4, 144, 256, 512, 24, 1, 1, 0, 1), (4, 16, 512, 1024, 96, 1, 1, 0, 1),
    (4, 96, 512, 1024, 16, 1, 1, 0, 1), (4, 32, 128, 256, 32, 1, 1, 0, 1), (4, 64, 32, 64, 64, 1, 1, 0, 1),
    (4, 32, 512, 1024, 32, 1, 1, 0, 1), (4, 64, 256, 512, 19, 3, 1, 1, 1), (4, 3, 1024, 2048, 32, 3, 2, 1, 1),
    (4, 32, 128, 256, 192, 1, 1, 0, 1), (4, 192, 128, 256, 32, 1, 1, 0, 1),
    (4, 64, 256, 512, 20, 3, 1, 1, 1), (4, 64, 256, 512, 32, 3, 1, 1, 1), (4, 64, 256, 512, 16, 3, 1, 1, 1),
]
DW = [  # B, C, H, W, K, stride, pad, dil
    (4, 32, 128, 256, 5, 1, 2, 1), (4, 24, 256, 512, 5, 1, 2, 1), (4, 32, 128, 256, 5, 1, 12, 6),
    (4, 64, 128, 256, 5, 1, 2, 1), (4, 96, 512, 1024, 3, 2, 1, 1), (4, 144, 256, 512, 3, 1, 1, 1),
    (4, 32, 512, 1024, 3, 1, 1, 1), (4, 64, 32, 64, 5, 1, 2, 1), (4, 32, 256, 512, 3, 1, 1, 1),
]


def bench_conv(which):
    for (B, K, H, W, N, k, s, p, d) in CONV:
        x = cl(B, K, H, W).requires_grad_(True)
        w = (torch.randn(N, K, k, k, device=DEV) * 0.1).requires_grad_(True)
        y = F.conv2d(x, w, None, s, p, d)
        g = torch.randn_like(y)
        Ho, Wo = y.shape[2], y.shape[3]
        by = 4 * (B * K * H * W + B * N * Ho * Wo)
        if which in ("conv", "all"):
            with torch.no_grad():
                t = timeit(lambda: F.conv2d(x, w, None, s, p, d))
            print("conv_fwd   K{:4d} N{:4d} {:4d}x{:4d} k{} s{}: {:8.1f} us {:8.1f} GB/s".format(
                K, N, H, W, k, s, t * 1e3, by / t / 1e6))
        if which in ("wgrad", "all"):
            t = timeit(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
            print("conv_wgrad K{:4d} N{:4d} {:4d}x{:4d} k{} s{}: {:8.1f} us {:8.1f} GB/s".format(
                K, N, H, W, k, s, t * 1e3, by / t / 1e6))
        if which in ("dgrad", "all") and K > 3:
            t = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
            print("conv_dgrad K{:4d} N{:4d} {:4d}x{:4d} k{} s{}: {:8.1f} us {:8.1f} GB/s".format(
                K, N, H, W, k, s, t * 1e3, by / t / 1e6))


def bench_dw(which):
    for (B, C, H, W, K, s, p, d) in DW:
        x = cl(B, C, H, W).requires_grad_(True)
        w = (torch.randn(C, 1, K, K, device=DEV) * 0.1).requires_grad_(True)
        y = F.depthwise_conv2d(x, w, s, p, d)
        g = torch.randn_like(y)
        by = 4 * (x.numel() + y.numel())
        if which in ("dw", "all"):
            with torch.no_grad():
                t = timeit(lambda: F.depthwise_conv2d(x, w, s, p, d))
            print("dw_fwd    C{:4d} {:4d}x{:4d} k{} s{} d{}: {:8.1f} us {:8.1f} GB/s".format(C, H, W, K, s, d, t * 1e3, by / t / 1e6))
            t = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
            print("dw_dgrad  C{:4d} {:4d}x{:4d} k{} s{} d{}: {:8.1f} us {:8.1f} GB/s".format(C, H, W, K, s, d, t * 1e3, by / t / 1e6))
        if which in ("dwwgrad", "all"):
            t = timeit(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
            print("dw_wgrad  C{:4d} {:4d}x{:4d} k{} s{} d{}: {:8.1f} us {:8.1f} GB/s".format(C, H, W, K, s, d, t * 1e3, by / t / 1e6))


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    if os.environ.get("KBENCH_ONLY_3X3"):
        CONV[:] = [c for c in CONV if c[5] == 3 and c[1] == 64]
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("NASSEG_")})
    for w in what:
        if w in ("conv", "wgrad", "dgrad", "all"):
            bench_conv(w)
        if w in ("dw", "dwwgrad", "all"):
            bench_dw(w)
