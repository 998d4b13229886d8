"""nasseg_irdw_fwd / nasseg_irdw_bwd (csrc/irdw.hip) against the kernels they replace.

MobileNetV2's InvertedResidual (src/nn/layer_factory.py:125-158): the 1x1 expansion's output z1 = W1 x is rebuilt on
the matrix cores inside the 3x3 depthwise kernels instead of being stored and read.  The rebuilt z1 has the bits the
pointwise forward kernels produce (same operand mapping, same accumulation order), so: the depthwise output and the
input gradient are BIT-IDENTICAL to the stored-z1 path; statistics rows and weight gradients agree to the rounding of
their (differently partitioned) sums.  Geometries: the three expansions of the encoder, both strides, ragged maps
(widths that are not multiples of the 14 / 7 / 16 columns of a strip, odd sizes), with and without a prologue on x,
fp32 and bf16 storage."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def F():
    from nas_segm_amd import functional

    return functional


def dev(t):
    t = t.to(DEV)
    return t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CASES = [
    # B, K, C, H, W, stride
    (2, 16, 96, 18, 22, 2), (1, 16, 96, 37, 45, 2), (2, 24, 144, 13, 17, 1), (1, 24, 144, 30, 61, 1),
    (2, 24, 144, 17, 23, 2), (1, 32, 192, 19, 33, 1), (3, 32, 192, 8, 9, 1), (1, 16, 96, 5, 3, 1), (1, 24, 144, 2, 2, 2),
    (1, 8, 48, 21, 40, 1), (2, 4, 16, 9, 29, 2), (1, 16, 96, 70, 15, 1),
]


def _setup(case, dtype, pro):
    Fm = F()
    lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream
    B, K, C, H, W, stride = case
    Ho, Wo = Fm.conv_out_size(H, 3, stride, 1, 1), Fm.conv_out_size(W, 3, stride, 1, 1)
    x = dev(rnd(B, K, H, W, seed=1)).to(dtype)
    w1 = dev(rnd(C, K, 1, 1, seed=2) * (1.0 / K ** 0.5))
    wd = dev(rnd(C, 1, 3, 3, seed=3) * 0.3)
    v = lambda n, seed, base=0.0, sc=0.2: (rnd(n, seed=seed) * sc + base).to(DEV)  # noqa: E731
    if pro:
        isc, ish, iact = v(K, 4, 1.0), v(K, 5), pro - 1  # pro 1: affine only (linear bottleneck), 2: affine + ReLU
    else:
        isc = ish = None
        iact = 0
    sc1, sh1 = v(C, 6, 1.0), v(C, 7, 0.5)
    wt = torch.empty(9 * C, device=DEV)
    wtf = torch.empty(9 * C, device=DEV)
    lib.call("nasseg_dw_pack_weight", ptr(wd), ptr(wt), C, 3, 0, stream())
    lib.call("nasseg_dw_pack_weight", ptr(wd), ptr(wtf), C, 3, 1, stream())
    # the stored expansion: z1 = W1 pro(x) by the general forward kernel
    z1 = torch.empty((B, C, H, W), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
    lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w1), ptr(z1), C, ptr(isc), ptr(ish), iact, None, None, 0, None,
             0, B, H, W, K, H, W, C, 1, 1, 1, 0, 1, 0, None, stream())
    return Fm, (B, K, C, H, W, stride, Ho, Wo), (x, w1, wd, wt, wtf, isc, ish, iact, sc1, sh1, z1)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}_{}to{}_{}x{}_s{}".format(*c))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_depthwise_forward_with_the_expansion_rebuilt(case, dtype, pro):
    Fm, geom, t = _setup(case, dtype, pro)
    lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream
    B, K, C, H, W, stride, Ho, Wo = geom
    x, w1, wd, wt, wtf, isc, ish, iact, sc1, sh1, z1 = t
    act1 = 2
    # the kernels it replaces: depthwise forward over the stored z1 with BatchNorm + ReLU6 applied on load
    nb = lib.query("nasseg_dwconv_stats_blocks", B, C, Ho, Wo, 3, stride, 1)
    z2_ref = torch.empty((B, C, Ho, Wo), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
    part_ref = torch.zeros((nb + 64) * 2 * C, device=DEV)
    lib.call(Fm._k("nasseg_dwconv", z1), ptr(z1), ptr(wt), ptr(z2_ref), ptr(sc1), ptr(sh1), act1, None, None, 0,
             B, H, W, C, Ho, Wo, 3, stride, 1, 1, 0, ptr(part_ref), stream())
    rows = lib.query("nasseg_irdw_rows", B, H, W, K, C, stride, 0)
    assert rows > 0
    z2 = torch.full_like(z2_ref, float("nan"))
    part = torch.full(((rows + 64) * 2 * C,), float("nan"), device=DEV)
    lib.call(Fm._k("nasseg_irdw_fwd", x), ptr(x), ptr(w1), ptr(wt), ptr(z2), ptr(isc), ptr(ish), iact, ptr(sc1),
             ptr(sh1), act1, B, H, W, K, C, Ho, Wo, stride, ptr(part), stream())
    assert torch.equal(z2, z2_ref), float((z2.float() - z2_ref.float()).abs().max())
    s_ref = part_ref[:nb * 2 * C].view(nb, 2 * C).double().sum(0)
    s_got = part[:rows * 2 * C].view(rows, 2 * C).double().sum(0)
    assert_close(s_got, s_ref, 2e-5 * float(s_ref.abs().max()) + 1e-6, 1e-4, "statistics rows")


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}_{}to{}_{}x{}_s{}".format(*c))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pro,bact,train", [(0, 0, True), (1, 2, True), (2, 1, False)])
def test_depthwise_backward_with_the_expansion_rebuilt(case, dtype, pro, bact, train):
    Fm, geom, t = _setup(case, dtype, pro)
    lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream
    B, K, C, H, W, stride, Ho, Wo = geom
    x, w1, wd, wt, wtf, isc, ish, iact, sc1, sh1, z1 = t
    act1 = 2
    v = lambda seed, base=0.0, sc=0.2: (rnd(C, seed=seed) * sc + base).to(DEV)  # noqa: E731
    mu1, is1 = v(8), v(9, 1.0).abs() + 0.3
    sc2, sh2, mu2, is2 = v(10, 1.0), v(11), v(12), v(13, 1.0).abs() + 0.3
    sums = (rnd(2 * C, seed=14) * 3).to(DEV)
    g = dev(rnd(B, C, Ho, Wo, seed=15)).to(dtype)
    z2 = dev(rnd(B, C, Ho, Wo, seed=16)).to(dtype)
    wb, flipped = (wtf, 1) if stride == 1 else (wt, 0)
    rows_ref = lib.query("nasseg_dwconv_bwd_bn_rows", B, C, H, W, 3, stride, 1, 1)
    assert rows_ref > 0
    ge_ref = torch.empty_like(z1)
    dw_ref = torch.empty_like(wd)
    ws_ref = torch.empty(rows_ref * 9 * C, device=DEV)
    part_ref = torch.zeros((rows_ref + 64) * 2 * C, device=DEV)
    lib.call(Fm._k("nasseg_dwconv_bwd_bn", z1), ptr(z1), ptr(g), ptr(z2), ptr(wb), flipped, ptr(ge_ref), ptr(dw_ref),
             ptr(ws_ref), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), act1, ptr(sc2), ptr(sh2), ptr(mu2), ptr(is2),
             ptr(sums), int(train), bact, B, H, W, C, Ho, Wo, 3, stride, 1, 1, ptr(part_ref), stream())
    rows = lib.query("nasseg_irdw_rows", B, H, W, K, C, stride, 1)
    assert rows > 0
    ge = torch.full_like(z1, float("nan"))
    dw = torch.full_like(wd, float("nan"))
    ws = torch.full((rows * 9 * C,), float("nan"), device=DEV)
    part = torch.full(((rows + 64) * 2 * C,), float("nan"), device=DEV)
    lib.call(Fm._k("nasseg_irdw_bwd", x), ptr(x), ptr(w1), ptr(g), ptr(z2), ptr(wb), flipped, ptr(ge), ptr(dw), ptr(ws),
             ptr(isc), ptr(ish), iact, ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), act1, ptr(sc2), ptr(sh2), ptr(mu2),
             ptr(is2), ptr(sums), int(train), bact, B, H, W, K, C, Ho, Wo, stride, ptr(part), stream())
    assert torch.equal(ge, ge_ref), float((ge.float() - ge_ref.float()).abs().max())
    M = B * Ho * Wo
    assert_close(dw, dw_ref, 5e-5 * float(dw_ref.abs().max()) * max(1.0, (M / 4096.0) ** 0.5) + 1e-6, 1e-4, "dw")
    s_ref = part_ref[:rows_ref * 2 * C].view(rows_ref, 2 * C).double().sum(0)
    s_got = part[:rows * 2 * C].view(rows, 2 * C).double().sum(0)
    tol = 2e-5 * float(s_ref.abs().max()) * max(1.0, (B * H * W / 4096.0) ** 0.5) + 1e-6
    assert_close(s_got, s_ref, tol, 1e-4, "statistics rows")


@pytest.mark.parametrize("case", [(1, 40, 52, 16, 96), (2, 33, 47, 24, 144), (1, 37, 41, 32, 192)],
                         ids=lambda c: "B{}_{}x{}_{}to{}".format(*c))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_statistics_only_pass_of_the_pointwise_kernel(case, dtype):
    """nasseg_conv_fwd with y == NULL: the statistics rows of the call that stores, bit for bit, and nothing written;
    where the N-split persistent kernel does not serve, an error - not a silent no-op"""
    Fm = F()
    lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream
    B, H, W, K, N = case
    x = dev(rnd(B, K, H, W, seed=1)).to(dtype)
    w = dev(rnd(N, K, 1, 1, seed=2) * 0.2)
    sc, sh = (rnd(K, seed=3) * 0.2 + 1).to(DEV), (rnd(K, seed=4) * 0.2).to(DEV)
    prev = lib.query("nasseg_conv_pwn_mode", 2)  # (every call the kernel supports, whatever the size)
    lib._memo.clear()
    try:
        assert lib.query("nasseg_conv_pointwise_kernel", B, H, W, N, K, 1) == 2
        nb = lib.query("nasseg_conv_fwd_stats_blocks", B, H, W, N, K, 1)
        y = torch.empty((B, N, H, W), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
        rows_ref = torch.zeros((nb + 64) * 2 * N, device=DEV)
        rows = torch.zeros((nb + 64) * 2 * N, device=DEV)
        args = (N, ptr(sc), ptr(sh), 2, None, None, 0, None, 0, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, 0)
        lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w), ptr(y), *args, ptr(rows_ref), stream())
        lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w), None, *args, ptr(rows), stream())
        assert torch.equal(rows[:nb * 2 * N], rows_ref[:nb * 2 * N])
        with pytest.raises(RuntimeError):
            lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w), None, *args, None, stream())
    finally:
        lib.query("nasseg_conv_pwn_mode", prev)
        lib._memo.clear()
    with pytest.raises(RuntimeError):  # a 3x3 conv has no such pass
        w3 = dev(rnd(N, K, 3, 3, seed=5) * 0.1)
        lib.call(Fm._k("nasseg_conv_fwd", x), ptr(x), K, ptr(w3), None, N, None, None, 0, None, None, 0, None, 0, B, H, W,
                 K, H, W, N, 3, 3, 1, 1, 1, 0, ptr(rows), stream())


@pytest.mark.parametrize("case", CASES + [(4, 16, 96, 64, 128, 2), (2, 32, 192, 40, 60, 1)],
                         ids=lambda c: "B{}_{}to{}_{}x{}_s{}".format(*c))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_statistics_of_the_expansion_from_the_moments_of_its_input(case, dtype, pro):
    """nasseg_irdw_stats: mean / invstd / scale / shift / running statistics / num_batches_tracked of z1 = W1 pro(x) from
    the K x K moments of pro(x) - against nasseg_bn_stats over the stored z1 (fp32: to 2e-5 of the statistics' scale;
    bf16 storage: the stored map is rounded, its statistics move by the rounding)"""
    Fm, geom, t = _setup(case, dtype, pro)
    lib, ptr, stream = Fm.lib, Fm.ptr, Fm.current_stream
    B, K, C, H, W, stride, Ho, Wo = geom
    x, w1, wd, wt, wtf, isc, ish, iact, sc1, sh1, z1 = t
    M = B * H * W
    if M < 2:
        pytest.skip("one value per channel")
    gamma, beta = (rnd(C, seed=21) * 0.2 + 1).to(DEV), (rnd(C, seed=22) * 0.2).to(DEV)
    eps, mom = 1e-5, 0.1

    def buffers():
        return [torch.empty(C, device=DEV) for _ in range(4)] + [torch.full((C,), 0.25, device=DEV),
                                                                 torch.full((C,), 1.5, device=DEV),
                                                                 torch.full((), 7, device=DEV, dtype=torch.int64)]

    ref = buffers()
    ws = torch.empty(lib.query("nasseg_colred_workspace", 1, M, C), device=DEV)
    lib.call(Fm._k("nasseg_bn_stats", z1), ptr(z1), C, M, C, eps, mom, ptr(gamma), ptr(beta), *[ptr(b) for b in ref],
             ptr(ws), stream())
    got = buffers()
    ws2 = torch.full((lib.query("nasseg_irdw_stats_workspace", K),), float("nan"), device=DEV)
    lib.call(Fm._k("nasseg_irdw_stats", x), ptr(x), ptr(w1), ptr(isc), ptr(ish), iact, B, H, W, K, C, eps, mom,
             ptr(gamma), ptr(beta), *[ptr(b) for b in got], ptr(ws2), stream())
    assert int(got[6]) == 8 == int(ref[6])
    rel = 2e-5 if dtype == torch.float32 else 2e-2
    std = 1.0 / ref[1]  # (per channel)
    assert float(((got[0] - ref[0]).abs() / std).max()) < rel, "mean"
    assert float(((got[1] - ref[1]).abs() / ref[1]).max()) < 10 * rel, "invstd"
    assert_close(got[2], ref[2], 10 * rel * float(ref[2].abs().max()), 10 * rel, "scale")
    assert float(((got[3] - ref[3]).abs()).max()) < 10 * rel * (1.0 + float((ref[0].abs() / std).max())), "shift"
    assert_close(got[4], ref[4], rel * float(std.max()) + 1e-6, rel, "running_mean")
    assert_close(got[5], ref[5], 10 * rel * float(ref[5].abs().max()), 10 * rel, "running_var")
