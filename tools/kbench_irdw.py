"""InvertedResidual's expansion stored against rebuilt (csrc/irdw.hip), at the encoder's geometries of the headline
step (4x3x1024x2048): microseconds per launch (HIP events, median of 20), forward and backward.
    python tools/kbench_irdw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nas_segm_amd  # noqa: E402,F401
from nas_segm_amd import functional as Fm  # noqa: E402

DEV = "cuda:0"
lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dtype = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
    B = 4
    for K, C, H, W, stride, pro in ((16, 96, 512, 1024, 2, 1), (24, 144, 256, 512, 1, 0), (24, 144, 256, 512, 2, 0),
                                    (32, 192, 128, 256, 1, 0)):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        g = torch.Generator().manual_seed(0)
        cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last).to(dtype)  # noqa: E731
        x = cl(torch.randn(B, K, H, W, generator=g))
        w1 = (torch.randn(C, K, 1, 1, generator=g) / K ** 0.5).to(DEV)
        wd = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(DEV)
        vec = lambda n, base=0.0: (torch.randn(n, generator=g) * 0.2 + base).to(DEV)  # noqa: E731
        isc, ish = (vec(K, 1.0), vec(K)) if pro else (None, None)
        sc1, sh1, mu1, is1 = vec(C, 1.0), vec(C, 0.5), vec(C), vec(C, 1.0).abs() + 0.3
        sc2, sh2, mu2, is2 = vec(C, 1.0), vec(C), vec(C), vec(C, 1.0).abs() + 0.3
        sums = vec(2 * C)
        wt, wtf = torch.empty(9 * C, device=DEV), torch.empty(9 * C, device=DEV)
        lib.call("nasseg_dw_pack_weight", ptr(wd), ptr(wt), C, 3, 0, stream())
        lib.call("nasseg_dw_pack_weight", ptr(wd), ptr(wtf), C, 3, 1, stream())
        z1 = torch.empty((B, C, H, W), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
        z2 = torch.empty((B, C, Ho, Wo), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
        gq = cl(torch.randn(B, C, Ho, Wo, generator=g))
        ge = torch.empty_like(z1)
        dw = torch.empty_like(wd)
        nb1 = lib.query("nasseg_conv_fwd_stats_blocks", B, H, W, C, K, 1)
        rows1 = torch.empty((nb1 + 64) * 2 * C, device=DEV)
        nb2 = lib.query("nasseg_dwconv_stats_blocks", B, C, Ho, Wo, 3, stride, 1)
        rows2 = torch.empty((nb2 + 64) * 2 * C, device=DEV)
        rf = lib.query("nasseg_irdw_rows", B, H, W, K, C, stride, 0)
        rb = lib.query("nasseg_irdw_rows", B, H, W, K, C, stride, 1)
        rows3 = torch.empty((max(rf, rb) + 64) * 2 * C, device=DEV)
        rows_old = lib.query("nasseg_dwconv_bwd_bn_rows", B, C, H, W, 3, stride, 1, 1)
        ws_old = torch.empty(rows_old * 9 * C, device=DEV)
        part_old = torch.empty((rows_old + 64) * 2 * C, device=DEV)
        ws_new = torch.empty(rb * 9 * C, device=DEV)
        wb, flipped = (wtf, 1) if stride == 1 else (wt, 0)
        pw_args = (C, ptr(isc), ptr(ish), 0, None, None, 0, None, 0, B, H, W, K, H, W, C, 1, 1, 1, 0, 1, 0)

        def pw1_store():
            lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w1), ptr(z1), *pw_args, ptr(rows1), stream())

        def pw1_stats():
            lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w1), None, *pw_args, ptr(rows1), stream())

        def dw_old():
            lib.call(Fm._k("nasseg_dwconv", z1), ptr(z1), ptr(wt), ptr(z2), ptr(sc1), ptr(sh1), 2, None, None, 0, B, H, W,
                     C, Ho, Wo, 3, stride, 1, 1, 0, ptr(rows2), stream())

        def dw_new():
            lib.call(Fm._k("nasseg_irdw_fwd", x), ptr(x), ptr(w1), ptr(wt), ptr(z2), ptr(isc), ptr(ish), 0, ptr(sc1),
                     ptr(sh1), 2, B, H, W, K, C, Ho, Wo, stride, ptr(rows3), stream())

        def bwd_old():
            lib.call(Fm._k("nasseg_dwconv_bwd_bn", z1), ptr(z1), ptr(gq), ptr(z2), ptr(wb), flipped, ptr(ge), None,
                     ptr(ws_old), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), 2, ptr(sc2), ptr(sh2), ptr(mu2), ptr(is2),
                     ptr(sums), 1, 2, B, H, W, C, Ho, Wo, 3, stride, 1, 1, ptr(part_old), stream())

        def bwd_new():
            lib.call(Fm._k("nasseg_irdw_bwd", x), ptr(x), ptr(w1), ptr(gq), ptr(z2), ptr(wb), flipped, ptr(ge), None,
                     ptr(ws_new), ptr(isc), ptr(ish), 0, ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), 2, ptr(sc2), ptr(sh2),
                     ptr(mu2), ptr(is2), ptr(sums), 1, 2, B, H, W, K, C, Ho, Wo, stride, ptr(rows3), stream())

        wsm = torch.empty(lib.query("nasseg_irdw_stats_workspace", K), device=DEV)
        st = [torch.empty(C, device=DEV) for _ in range(4)]

        def moments():
            lib.call(Fm._k("nasseg_irdw_stats", x), ptr(x), ptr(w1), ptr(isc), ptr(ish), 0, B, H, W, K, C, 1e-5, 0.1, None,
                     None, ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), None, None, None, ptr(wsm), stream())

        def finalize_old():
            lib.call("nasseg_bn_finalize", ptr(rows1), nb1, B * H * W, C, 1e-5, 0.1, None, None, ptr(st[0]), ptr(st[1]),
                     ptr(st[2]), ptr(st[3]), None, None, None, stream())

        pw1_store()
        t = [timed(f) for f in (pw1_store, pw1_stats, dw_old, dw_new, bwd_old, bwd_new)]
        tm, tf = timed(moments), timed(finalize_old)
        print("   statistics of the expansion: stored conv + finaliser {:.1f} + {:.1f} | from the input's moments (3 launches) "
              "{:.1f}  => forward of expansion + depthwise {:.1f} -> {:.1f}".format(t[0], tf, tm, t[0] + tf + t[2], tm + t[3]))
        print("{:2d} -> {:3d} at {}x{}x{} stride {} {}: expansion stored {:6.1f} | statistics only {:6.1f}; depthwise fwd "
              "{:6.1f} | rebuilt {:6.1f}  => fwd {:6.1f} -> {:6.1f}; depthwise bwd {:6.1f} | rebuilt {:6.1f}  (rows fwd {} bwd {})"
              .format(K, C, B, H, W, stride, "bf16" if dtype == torch.bfloat16 else "fp32", t[0], t[1], t[2], t[3],
                      t[0] + t[2], t[1] + t[3], t[4], t[5], rf, rb))


if __name__ == "__main__":
    main()
