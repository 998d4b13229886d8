"""Direct (no autograd) timing of nasseg_dwconv_wgrad at the decoder's and encoder's depthwise shapes of the
headline step.  usage: python tools/kbench_dwwgrad.py   (NASSEG_DW_WGRAD_ROWS=0: the strip kernel everywhere)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as F  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # B, C, H, W, K, stride, pad, dil
    (4, 32, 128, 256, 5, 1, 2, 1), (4, 64, 128, 256, 5, 1, 2, 1), (4, 32, 128, 256, 5, 1, 12, 6),
    (4, 24, 256, 512, 5, 1, 2, 1), (4, 64, 32, 64, 5, 1, 2, 1), (4, 32, 256, 512, 3, 1, 1, 1),
    (4, 32, 64, 128, 3, 1, 1, 1), (4, 64, 64, 128, 5, 1, 2, 1), (4, 144, 256, 512, 3, 1, 1, 1),
    (4, 96, 512, 1024, 3, 2, 1, 1), (4, 64, 256, 512, 5, 1, 12, 6), (4, 24, 256, 512, 5, 1, 12, 6),
    (4, 48, 128, 256, 5, 1, 2, 1), (16, 64, 11, 11, 5, 1, 2, 1), (16, 64, 41, 41, 5, 1, 12, 6),
]


def main():
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("NASSEG_")})
    s = F.current_stream()
    for (B, C, H, W, K, st, pad, dil) in SHAPES:
        Ho, Wo = F.conv_out_size(H, K, st, pad, dil), F.conv_out_size(W, K, st, pad, dil)
        x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, C, Ho, Wo, device=DEV).contiguous(memory_format=torch.channels_last)
        dw = torch.empty(C, 1, K, K, device=DEV)
        ws = torch.empty(F.lib.query("nasseg_dwconv_wgrad_workspace", B, C, Ho, Wo, K), device=DEV)

        def run():
            F.lib.call("nasseg_dwconv_wgrad", F.ptr(x), F.ptr(dy), F.ptr(dw), F.ptr(ws), None, None, 0,
                       B, H, W, C, Ho, Wo, K, st, pad, dil, s)
        times = []
        for mode in ((0, 1) if K == 5 else (1,)):
            F.lib.call("nasseg_dw_wgrad_lds", mode)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 40
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / n)
        t = times[-1]
        ref = torch.nn.grad.conv2d_weight(x.cpu(), (C, 1, K, K), dy.cpu(), stride=st, padding=pad, dilation=dil,
                                          groups=C) if H * W <= 128 * 256 else None
        err = float((dw.cpu() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        by = 4 * (x.numel() + dy.numel())
        print("dw_wgrad C{:4d} B{:2d} {:4d}x{:4d} k{} s{} d{}: {} us (strip, LDS) {:8.1f} GB/s  rel err {:.1e}".format(
            C, B, H, W, K, st, dil, " ".join("{:7.1f}".format(v * 1e3) for v in times), by / t / 1e6, err))


if __name__ == "__main__":
    main()
