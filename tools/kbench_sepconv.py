"""Micro-benchmark of one SepConv stage: the fused kernel (nasseg_sepconv_fwd) against the two
separate launches (nasseg_dwconv + nasseg_conv_fwd with the statistics epilogue), training form
(depthwise output stored, statistics rows) and inference form.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

lib, ptr, stream = F.lib, F.ptr, F.current_stream
DEV = "cuda:0"
CASES = [
    # B, C, N, H, W, k, stride, pad, dil
    (4, 32, 32, 128, 256, 5, 1, 2, 1),
    (4, 32, 32, 128, 256, 5, 1, 12, 6),
    (4, 64, 64, 128, 256, 5, 1, 2, 1),
    (4, 24, 64, 256, 512, 5, 1, 2, 1),
    (4, 64, 64, 256, 512, 5, 1, 12, 6),
    (4, 32, 32, 256, 512, 3, 1, 1, 1),
    (4, 64, 64, 32, 64, 5, 1, 2, 1),
    (8, 48, 48, 179, 179, 5, 1, 2, 1),
    (4, 48, 48, 128, 256, 5, 1, 2, 1),
    (4, 24, 24, 256, 512, 3, 1, 1, 1),
    (4, 64, 64, 64, 128, 5, 1, 2, 1),
    (4, 32, 32, 128, 256, 3, 2, 1, 1),
    (16, 64, 64, 81, 81, 5, 1, 12, 6),
    (16, 64, 64, 21, 21, 3, 1, 1, 1),
]


def timeit(fn, reps=30):
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


for case in CASES:
    B, C, N, H, W, k, stride, pad, dil = case
    Ho, Wo = F.conv_out_size(H, k, stride, pad, dil), F.conv_out_size(W, k, stride, pad, dil)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    wdw = torch.randn(C, 1, k, k, device=DEV)
    wpw = torch.randn(N, C, 1, 1, device=DEV)
    wt = torch.empty(k * k * C, device=DEV)
    lib.call("nasseg_dw_pack_weight", ptr(wdw), ptr(wt), C, k, 0, stream())
    z = torch.empty((B, C, Ho, Wo), device=DEV).contiguous(memory_format=torch.channels_last)
    y = torch.empty((B, N, Ho, Wo), device=DEV).contiguous(memory_format=torch.channels_last)
    nb1 = lib.query("nasseg_conv_fwd_stats_blocks", B, Ho, Wo, N, C, 1)
    nb2 = lib.query("nasseg_sepconv_blocks", B, C, Ho, Wo, N, k, stride, dil)
    part = torch.empty((max(nb1, nb2) + 64) * 2 * N, device=DEV)
    s = stream()

    def separate(stats=True):
        lib.call("nasseg_dwconv", ptr(x), ptr(wt), ptr(z), None, None, 0, None, None, 0, B, H, W, C, Ho, Wo, k,
                 stride, pad, dil, 0, None, s)
        lib.call("nasseg_conv_fwd", ptr(z), C, ptr(wpw), ptr(y), N, None, None, 0, None, None, 0, None, 0, B, Ho,
                 Wo, C, Ho, Wo, N, 1, 1, 1, 0, 1, 0, ptr(part) if stats else None, s)

    def fused(train=True):
        lib.call("nasseg_sepconv_fwd", ptr(x), ptr(wt), ptr(wpw), ptr(z) if train else None, ptr(y), None, None, 0,
                 None, None, 0, B, H, W, C, Ho, Wo, N, k, stride, pad, dil, ptr(part) if train else None, s)

    t_sep, t_fus = timeit(separate), timeit(fused)
    t_sep_i, t_fus_i = timeit(lambda: separate(False)), timeit(lambda: fused(False))
    mb = 4e-6 * B * (C * H * W + (C + N) * Ho * Wo)
    print("{}: train separate {:7.1f} us fused {:7.1f} us ({:5.0f} GB/s of x+zdw+y) | inference separate {:7.1f} "
          "fused {:7.1f} us".format(case, t_sep, t_fus, mb / t_fus * 1e3, t_sep_i, t_fus_i))
