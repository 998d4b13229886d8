"""Micro-benchmark of the one-kernel pointwise backward (nasseg_conv_pw_bwd_bn) against the two
kernels it replaces (nasseg_conv_wgrad_bn + nasseg_conv_fwd as backward-data).  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
DEV = "cuda:0"
CASES = [(4, 512, 1024, 16, 96), (4, 256, 512, 24, 144), (4, 128, 256, 32, 192), (4, 512, 1024, 32, 32),
         (4, 256, 512, 24, 64), (4, 128, 256, 64, 64), (4, 128, 256, 32, 32), (4, 32, 64, 64, 64), (4, 256, 512, 224, 64), (16, 81, 81, 192, 64),
         (16, 11, 11, 320, 64)]


if os.environ.get("KBENCH_CASES"):  # "B,H,W,K,N;..."
    CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["KBENCH_CASES"].split(";")]
if os.environ.get("KBENCH_WIDE"):
    CASES = [c for c in CASES if c[3] > 64] + [(4, 256, 512, 128, 64), (8, 179, 179, 192, 48)]


def timeit(fn, reps=20):
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


for B, H, W, K, N in CASES:
    mk = lambda C: torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, g, z = mk(K), mk(N), mk(N)
    w = torch.randn(N, K, 1, 1, device=DEV)
    wb = torch.empty(N * K, device=DEV)
    lib.call("nasseg_conv_pack_weight", ptr(w), ptr(wb), N, K, 1, 1, 1, stream())
    v = lambda n: torch.rand(n, device=DEV) + 0.5  # noqa: E731
    scale, shift, mean, invstd, sums = v(N), v(N), v(N), v(N), v(2 * N)
    dz, dx, dw = torch.empty_like(z), torch.empty_like(x), torch.empty_like(w)
    ws = torch.empty(lib.query("nasseg_conv_wgrad_workspace", B, H, W, N, K, 1, 1), device=DEV)
    nsl = lib.query("nasseg_conv_pw_bwd_slabs", B, H, W, K, N)
    ws2 = torch.empty(max(nsl, 1) * N * K, device=DEV)
    s = stream()

    def two():
        lib.call("nasseg_conv_wgrad_bn", ptr(x), K, ptr(g), N, ptr(z), N, ptr(dz), N, None, ptr(ws), None, None, 0,
                 ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), 1, 1, B, H, W, K, N, s)
        lib.call("nasseg_conv_fwd", ptr(dz), N, ptr(wb), ptr(dx), K, None, None, 0, None, None, 0, None, 0, B, H, W,
                 N, H, W, K, 1, 1, 1, 0, 1, 1, None, s)

    def one():
        lib.call("nasseg_conv_pw_bwd_bn", ptr(x), ptr(g), ptr(z), ptr(wb), ptr(dx), None, ptr(ws2), None, None, 0,
                 0, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), 1, 1, B, H, W, K, N, None, None, None, None, s)

    def one_dxs():  # (the conv sits behind a BatchNorm of the chain: the first half of ITS backward in the epilogue)
        lib.call("nasseg_conv_pw_bwd_bn", ptr(x), ptr(g), ptr(z), ptr(wb), ptr(dx), None, ptr(ws2), ptr(psc), ptr(psh), 2,
                 0, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), 1, 1, B, H, W, K, N, ptr(imean), ptr(iis),
                 ptr(dxs), None, s)

    if os.environ.get("KBENCH_DXS") and nsl and K <= 64:
        psc, psh, imean, iis = v(K), v(K), v(K), v(K)
        dxs = torch.empty((nsl + 64) * 2 * K, device=DEV)
        print("{}: one kernel with the sums of the BatchNorm in front {:7.1f} us, slabs {}".format(
            (B, H, W, K, N), timeit(one_dxs), nsl))
        continue
    t2, t1 = timeit(two), (timeit(one) if nsl else float("nan"))
    mb = 4e-6 * B * H * W * (2 * K + 2 * N)
    print("{}: two kernels {:7.1f} us, one kernel {:7.1f} us ({:5.0f} GB/s of x+g+z+dx), slabs {}".format(
        (B, H, W, K, N), t2, t1, mb / t1 * 1e3, nsl))
