"""Micro-benchmark of the one-kernel backward of a 3x3 depthwise conv between two BatchNorms
(nasseg_dwconv_bwd_bn) against nasseg_dwconv_wgrad_bn + nasseg_dwconv_bwd_data_bn.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
DEV = "cuda:0"
CASES = [(4, 96, 512, 1024, 2), (4, 144, 256, 512, 1), (4, 32, 512, 1024, 1), (4, 144, 256, 512, 2),
         (4, 192, 128, 256, 1)]


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


for B, C, H, W, stride in CASES:
    Ho, Wo = F.conv_out_size(H, 3, stride, 1, 1), F.conv_out_size(W, 3, stride, 1, 1)
    mk = lambda h, w: torch.randn(B, C, h, w, device=DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    xz, g, z = mk(H, W), mk(Ho, Wo), mk(Ho, Wo)
    w = torch.randn(C, 1, 3, 3, device=DEV)
    wt, wtf = torch.empty(9 * C, device=DEV), torch.empty(9 * C, device=DEV)
    lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wt), C, 3, 0, stream())
    lib.call("nasseg_dw_pack_weight", ptr(w), ptr(wtf), C, 3, 1, stream())
    v = lambda n: torch.rand(n, device=DEV) + 0.5  # noqa: E731
    isc, ish, imu, iis, scale, shift, mean, invstd, sums = v(C), v(C), v(C), v(C), v(C), v(C), v(C), v(C), v(2 * C)
    dz, ge = torch.empty_like(z), torch.empty_like(xz)
    ws = torch.empty(lib.query("nasseg_dwconv_wgrad_workspace", B, C, Ho, Wo, 3), device=DEV)
    geom = (B, Ho, Wo, C, H, W, 3, 1, 1, 1, 0) if stride == 1 else (B, Ho, Wo, C, H, W, 3, stride, 1, 1, 1)
    wb = wtf if stride == 1 else wt
    nb = lib.query("nasseg_dwconv_bwd_data_bn_blocks", B, C, H, W, 3, geom[7], geom[8], 1, geom[10])
    part = torch.empty((nb + 64) * 2 * C, device=DEV)
    rows = lib.query("nasseg_dwconv_bwd_bn_rows", B, C, H, W, 3, stride, 1, 1)
    ws2 = torch.empty(rows * 9 * C, device=DEV)
    part2 = torch.empty((rows + 64) * 2 * C, device=DEV)
    s = stream()

    def two():
        lib.call("nasseg_dwconv_wgrad_bn", ptr(xz), ptr(g), ptr(z), ptr(dz), None, ptr(ws), ptr(isc), ptr(ish), 2,
                 ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums), 1, 0, B, H, W, C, Ho, Wo, 3, stride, 1, 1, s)
        lib.call("nasseg_dwconv_bwd_data_bn", ptr(dz), ptr(wb), ptr(ge), ptr(xz), ptr(isc), ptr(ish), ptr(imu),
                 ptr(iis), 2, *geom, ptr(part), s)

    def one():
        lib.call("nasseg_dwconv_bwd_bn", ptr(xz), ptr(g), ptr(z), ptr(wb), int(stride == 1), ptr(ge), None, ptr(ws2),
                 ptr(isc), ptr(ish), ptr(imu), ptr(iis), 2, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums),
                 1, 0, B, H, W, C, Ho, Wo, 3, stride, 1, 1, ptr(part2), s)

    t2, t1 = timeit(two), timeit(one)
    mb = 4e-6 * B * C * (2 * H * W + 2 * Ho * Wo)
    print("{}: two kernels {:7.1f} us, one kernel {:7.1f} us ({:5.0f} GB/s of xz+g+z+ge), rows {}".format(
        (B, C, H, W, stride), t2, t1, mb / t1 * 1e3, rows))
