"""conv_ws.hip -- the weights-stationary persistent convolution kernels -- against stock PyTorch-CPU fp32 convolutions of the
same rounded inputs (fp32 1e-3 relative, bf16 2e-2: the bounds of tests/test_kernels_gpu.py), through the C ABI
(dsn_conv2d_fwd / dsn_conv2d_dgrad / dsn_conv2d_fwd_bnacc pick the kernel; the library's labelled profiler confirms which one ran).
Cases: several tiles per persistent block, ragged pixel and channel tiles, partial 128-byte slabs (32 / 40 / 96 channels),
channel slices of wider buffers, bias + activation epilogues, BatchNorm partial sums."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import TOL, q, rnd, to_dev
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from desenet_amd import hip_ops
    return hip_ops


def _ran(ops, fn, family):
    ops.profile_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        prof = ops.profile_collect()
    finally:
        ops.profile_enable(False)
    assert any(k.startswith(family) for k in prof), (family, list(prof))
    return out


# n, ci, h, w, co
CASES_1X1 = [
    (8, 128, 80, 80, 64),      # config 3's C3 cv1: 800 pixel tiles over <= 512 blocks (several tiles per block, two slabs)
    (2, 64, 47, 45, 64),       # ragged last pixel tile, one slab
    (2, 32, 64, 64, 32),       # half a slab (32 bf16 channels), 128 x 32 tile
    (2, 96, 40, 56, 40),       # one and a half slabs, ragged channel tile (40 of 64)
    (1, 256, 48, 48, 200),     # four slabs (32 x 64 tiles; eight in fp32: the long-K form), ragged last channel tile
    (2, 160, 40, 40, 96),      # five fp32 slabs / two and a half bf16 slabs
    (3, 40, 31, 33, 72),       # nothing aligned to a tile
    (4, 128, 40, 40, 128),
    (8, 512, 20, 20, 256),     # eight slabs (bf16 only): 64 KB of resident weights, 2-stage ring, one block per CU
    (2, 448, 40, 40, 72),      # seven slabs, ragged channel tile
    (8, 256, 20, 20, 512),     # the data gradient's K = 512
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES_1X1)
@pytest.mark.parametrize("act", ["silu", "none"])
def test_conv1x1_ws_forward(ops, case, dtype, act):
    n, ci, h, w, co = case
    if dtype == torch.float32 and (ci > 256 or (ci > 128 and n * h * w < 2048)):
        pytest.skip("more than eight fp32 slabs (or a long-K launch on a tiny map): not a weights-stationary launch")
    x, wt, b = rnd((n, ci, h, w), 11), rnd((co, ci, 1, 1), 12, -0.3, 0.3), rnd((co,), 13)
    ref = F.conv2d(q(x, dtype), q(wt, dtype), b)
    if act == "silu":
        ref = F.silu(ref)
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    p = ops.conv_params(1, act=ops.ACT_SILU if act == "silu" else ops.ACT_NONE)
    _ran(ops, lambda: ops.conv2d_fwd(xd, wp, b.cuda(), None, y, p), "conv1x1_ws_kernel")
    assert_close(y.float().cpu(), ref, TOL[dtype], f"ws 1x1 fwd {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES_1X1)
def test_conv1x1_ws_dgrad(ops, case, dtype):
    n, ci, h, w, co = case
    if dtype == torch.float32 and (co > 256 or (co > 128 and n * h * w < 2048)):
        pytest.skip("more than eight fp32 slabs (or a long-K launch on a tiny map): not a weights-stationary launch")
    if ci > 256 and co < 64:
        pytest.skip("fewer than 64 source channels with a long destination: fine, but not a case worth its time")
    dy, wt = rnd((n, co, h, w), 21), rnd((co, ci, 1, 1), 22, -0.3, 0.3)
    ref = F.conv_transpose2d(q(dy, dtype), q(wt, dtype))
    dyd = to_dev(ops, dy, dtype)
    wd = ops.pack_weight_dgrad(wt.cuda(), dtype)
    dx = ops.new_act(n, ci, h, w, dtype, "cuda")
    _ran(ops, lambda: ops.conv2d_dgrad(dyd, wd, dx, ops.conv_params(1)), "conv1x1_ws_kernel")
    assert_close(dx.float().cpu(), ref, TOL[dtype], f"ws 1x1 dgrad {case}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1x1_ws_channel_slices(ops, dtype):
    """Reads a channel slice of a wider buffer and writes a slice of a concat buffer (pixel strides larger than the channel counts);
    nothing outside the destination slice is touched."""
    n, h, w = 2, 40, 48
    big = rnd((n, 192, h, w), 31)
    wt = rnd((96, 64, 1, 1), 32, -0.3, 0.3)
    xin = to_dev(ops, big, dtype)
    ref = F.conv2d(q(big[:, 64:128], dtype), q(wt, dtype))
    cat = ops.new_act(n, 160, h, w, dtype, "cuda", zero=True)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    _ran(ops, lambda: ops.conv2d_fwd(xin[:, 64:128], wp, None, None, cat[:, 32:128], ops.conv_params(1)), "conv1x1_ws_kernel")
    assert_close(cat[:, 32:128].float().cpu(), ref, TOL[dtype], "slice write")
    assert float(cat[:, :32].abs().max()) == 0 and float(cat[:, 128:].abs().max()) == 0, "wrote outside the slice"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(8, 128, 80, 80, 64), (2, 96, 40, 56, 40), (3, 64, 31, 33, 72)])
def test_conv1x1_ws_batchnorm_sums(ops, case, dtype):
    """Training forward: y = conv(x) and the per-channel fp64 sums of y's fp32 accumulators (dsn_conv2d_fwd_bnacc), folded by
    dsn_bn_act_fwd_acc -- mean / rstd / output against F.batch_norm of the reference convolution."""
    n, ci, h, w, co = case
    x, wt = rnd((n, ci, h, w), 41), rnd((co, ci, 1, 1), 42, -0.3, 0.3)
    g, b = rnd((co,), 43, 0.5, 1.5), rnd((co,), 44, -0.2, 0.2)
    yref = F.conv2d(q(x, dtype), q(wt, dtype))
    zref = F.silu(F.batch_norm(yref, None, None, g, b, True, 0.03, 1e-3))
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    z = ops.new_act(n, co, h, w, dtype, "cuda")
    rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
    stats = _ran(ops, lambda: ops.conv2d_fwd_bnstats(xd, wp, y, ops.conv_params(1), g.cuda(), b.cuda(), rm, rv, 0.03, 1e-3,
                                                     ops.ACT_SILU, None, z), "conv1x1_ws_kernel")
    mean = yref.mean((0, 2, 3))
    var = yref.var((0, 2, 3), unbiased=False)
    assert_close(stats[2].cpu(), mean, 2e-3 if dtype == torch.float32 else 2e-2, "batch mean")
    assert_close(stats[3].cpu(), 1.0 / torch.sqrt(var + 1e-3), 2e-3 if dtype == torch.float32 else 2e-2, "rstd")
    assert_close(y.float().cpu(), yref, TOL[dtype], "conv output")
    assert_close(z.float().cpu(), zref, 2 * TOL[dtype], "BN + SiLU output")


# n, ci, h, w, co, dil
CASES_3X3 = [
    (8, 64, 80, 80, 64, 1),      # config 3's Bottleneck 3x3 @80: 800 patches of 8 x 8 over 256 persistent blocks
    (2, 64, 40, 48, 64, 2),      # RFB2 branch, dilation 2 (12 x 12 halo)
    (2, 64, 40, 40, 64, 3),      # dilation 3 (14 x 14 halo)
    (2, 64, 70, 67, 96, 1),      # ragged patches at both borders, ragged channel tile
    (2, 32, 64, 64, 32, 1),      # half a slab, 8 x 16 patches x 32 channels
    (4, 128, 40, 40, 128, 1),    # two slabs per patch (the halo of slab 1 is prefetched while slab 0 is multiplied)
    (2, 96, 33, 47, 40, 1),      # one and a half slabs, ragged everything
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES_3X3)
def test_conv3x3_ws_forward_and_dgrad(ops, case, dtype):
    n, ci, h, w, co, d = case
    if dtype == torch.float32 and (ci > 64 or (ci > 32 and d > 1)):
        pytest.skip("more than two fp32 slabs (or two with dilation): not a weights-stationary launch")
    x, wt, b = rnd((n, ci, h, w), 51), rnd((co, ci, 3, 3), 52, -0.2, 0.2), rnd((co,), 53)
    ref = F.silu(F.conv2d(q(x, dtype), q(wt, dtype), b, 1, d, d))
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    _ran(ops, lambda: ops.conv2d_fwd(xd, wp, b.cuda(), None, y, ops.conv_params(3, 1, d, d, act=ops.ACT_SILU)), "conv3x3_ws_kernel")
    assert_close(y.float().cpu(), ref, TOL[dtype], f"ws 3x3 fwd {case}")
    if dtype == torch.float32 and (co > 64 or (co > 32 and d > 1)):
        return
    dy = rnd((n, co, h, w), 54)
    dref = F.conv_transpose2d(q(dy, dtype), q(wt, dtype), None, 1, d, 0, 1, d)
    dyd = to_dev(ops, dy, dtype)
    wd = ops.pack_weight_dgrad(wt.cuda(), dtype)
    dx = ops.new_act(n, ci, h, w, dtype, "cuda")
    _ran(ops, lambda: ops.conv2d_dgrad(dyd, wd, dx, ops.conv_params(3, 1, d, d)), "conv3x3_ws_kernel")
    assert_close(dx.float().cpu(), dref, TOL[dtype], f"ws 3x3 dgrad {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(8, 64, 80, 80, 64, 1), (2, 64, 70, 67, 96, 1), (4, 128, 40, 40, 128, 1)])
def test_conv3x3_ws_batchnorm_sums(ops, case, dtype):
    n, ci, h, w, co, d = case
    if dtype == torch.float32 and ci > 64:
        pytest.skip("more than two fp32 slabs")
    x, wt = rnd((n, ci, h, w), 61), rnd((co, ci, 3, 3), 62, -0.2, 0.2)
    g, b = rnd((co,), 63, 0.5, 1.5), rnd((co,), 64, -0.2, 0.2)
    yref = F.conv2d(q(x, dtype), q(wt, dtype), None, 1, d, d)
    zref = F.silu(F.batch_norm(yref, None, None, g, b, True, 0.03, 1e-3))
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    z = ops.new_act(n, co, h, w, dtype, "cuda")
    rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
    stats = _ran(ops, lambda: ops.conv2d_fwd_bnstats(xd, wp, y, ops.conv_params(3, 1, d, d), g.cuda(), b.cuda(), rm, rv, 0.03, 1e-3,
                                                     ops.ACT_SILU, None, z), "conv3x3_ws_kernel")
    assert_close(stats[2].cpu(), yref.mean((0, 2, 3)), 2e-3 if dtype == torch.float32 else 2e-2, "batch mean")
    assert_close(y.float().cpu(), yref, TOL[dtype], "conv output")
    assert_close(z.float().cpu(), zref, 2 * TOL[dtype], "BN + SiLU output")


@pytest.mark.parametrize("case", [(8, 16, 320, 320, 32), (2, 16, 70, 67, 24), (1, 16, 48, 64, 32)])
@pytest.mark.parametrize("mode", ["act", "stats"])
def test_conv3x3_thin_input_focus_kernel(ops, case, mode):
    """Focus' convolution shape: 16 bf16 channels per pixel (12 real + 4 zero lanes), 3x3 / stride 1 -- the three taps of a kernel row
    as one contiguous 96-byte run (conv3x3_thin_ws_kernel), with the fused bias + SiLU epilogue and with BatchNorm sums."""
    n, ci, h, w, co = case
    dtype = torch.bfloat16
    x = rnd((n, ci, h, w), 71)
    x[:, 12:] = 0.0                                   # the padding lanes of the space-to-depth tensor
    wt, b = rnd((co, ci, 3, 3), 72, -0.2, 0.2), rnd((co,), 73)
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    if mode == "act":
        ref = F.silu(F.conv2d(q(x, dtype), q(wt, dtype), b, 1, 1))
        _ran(ops, lambda: ops.conv2d_fwd(xd, wp, b.cuda(), None, y, ops.conv_params(3, act=ops.ACT_SILU)), "conv3x3_thin_ws_kernel")
        assert_close(y.float().cpu(), ref, TOL[dtype], f"thin 3x3 {case}")
    else:
        g_, b_ = rnd((co,), 74, 0.5, 1.5), rnd((co,), 75, -0.2, 0.2)
        yref = F.conv2d(q(x, dtype), q(wt, dtype), None, 1, 1)
        zref = F.silu(F.batch_norm(yref, None, None, g_, b_, True, 0.03, 1e-3))
        z = ops.new_act(n, co, h, w, dtype, "cuda")
        rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
        stats = _ran(ops, lambda: ops.conv2d_fwd_bnstats(xd, wp, y, ops.conv_params(3), g_.cuda(), b_.cuda(), rm, rv, 0.03, 1e-3,
                                                         ops.ACT_SILU, None, z), "conv3x3_thin_ws_kernel")
        assert_close(stats[2].cpu(), yref.mean((0, 2, 3)), 2e-2, "batch mean")
        assert_close(y.float().cpu(), yref, TOL[dtype], "conv output")
        assert_close(z.float().cpu(), zref, 2 * TOL[dtype], "BN + SiLU output")


@pytest.mark.parametrize("case", [(16, 12, 320, 320, 32), (2, 12, 70, 67, 24), (1, 12, 48, 64, 48)])
@pytest.mark.parametrize("mode", ["act", "stats"])
def test_conv3x3_thin_fp32_focus_kernel(ops, case, mode):
    """The fp32 Focus shape: 12 channels per pixel (48 bytes), 3x3 / stride 1 -- a kernel row's three taps as one run of 36 floats
    (conv3x3_thin_f32_ws_kernel: 9 k-steps instead of 18), fused bias + SiLU epilogue and BatchNorm sums."""
    n, ci, h, w, co = case
    dtype = torch.float32
    x = rnd((n, ci, h, w), 81)
    wt, b = rnd((co, ci, 3, 3), 82, -0.2, 0.2), rnd((co,), 83)
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, h, w, dtype, "cuda")
    if mode == "act":
        ref = F.silu(F.conv2d(x, wt, b, 1, 1))
        _ran(ops, lambda: ops.conv2d_fwd(xd, wp, b.cuda(), None, y, ops.conv_params(3, act=ops.ACT_SILU)), "conv3x3_thin_f32_ws_kernel")
        assert_close(y.float().cpu(), ref, TOL[dtype], f"thin fp32 3x3 {case}")
    else:
        g_, b_ = rnd((co,), 84, 0.5, 1.5), rnd((co,), 85, -0.2, 0.2)
        yref = F.conv2d(x, wt, None, 1, 1)
        zref = F.silu(F.batch_norm(yref, None, None, g_, b_, True, 0.03, 1e-3))
        z = ops.new_act(n, co, h, w, dtype, "cuda")
        rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
        stats = _ran(ops, lambda: ops.conv2d_fwd_bnstats(xd, wp, y, ops.conv_params(3), g_.cuda(), b_.cuda(), rm, rv, 0.03, 1e-3,
                                                         ops.ACT_SILU, None, z), "conv3x3_thin_f32_ws_kernel")
        assert_close(stats[2].cpu(), yref.mean((0, 2, 3)), 1e-3, "batch mean")
        assert_close(y.float().cpu(), yref, TOL[dtype], "conv output")
        assert_close(z.float().cpu(), zref, 2 * TOL[dtype], "BN + SiLU output")


def _layers_of(ops, fn):
    ops.profile_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        _, layers = ops.profile_collect(by_layer=True)
    finally:
        ops.profile_enable(False)
    return out, layers


# n, ci, h, w, co  (3x3 / stride 2 / pad 1; >= 16384 output pixels so that the gather form takes the launch)
CASES_S2 = [
    (8, 32, 320, 320, 64),     # DeSeNet-s layer 1: K = 288 = 4.5 slabs, 2 blocks per CU
    (4, 32, 150, 134, 64),     # odd output map (75 x 67), ragged last tile
    (8, 64, 160, 160, 128),    # layer 3: K = 576 = 9 slabs, two output-channel tiles
    (3, 64, 166, 150, 72),     # odd sizes, ragged channel tile (72 of 128)
]


@pytest.mark.parametrize("case", CASES_S2)
def test_stride2_forward_gathered_k(ops, case):
    """3x3 / stride-2 forward through the weights-stationary kernel's gather form (K = 9 taps x Ci fetched by LDS-DMA from per-tap source
    pixels; borders = out-of-range lanes): conv output and the BatchNorm sums of its epilogue against ATen."""
    n, ci, h, w, co = case
    dtype = torch.bfloat16
    x = rnd((n, ci, h, w), 111)
    wt = rnd((co, ci, 3, 3), 112, -0.2, 0.2)
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
    y = ops.new_act(n, co, ho, wo, dtype, "cuda")
    yref = F.conv2d(q(x, dtype), q(wt, dtype), None, 2, 1)
    _, layers = _layers_of(ops, lambda: ops.conv2d_fwd(xd, wp, None, None, y, ops.conv_params(3, 2, 1, 1)))
    assert any(k[0].startswith("conv1x1_ws_kernel") and k[1].startswith("k3s2") for k in layers), list(layers)
    assert_close(y.float().cpu(), yref, TOL[dtype], f"s2 fwd {case}")
    g_, b_ = rnd((co,), 113, 0.5, 1.5), rnd((co,), 114, -0.2, 0.2)
    z = ops.new_act(n, co, ho, wo, dtype, "cuda")
    y2 = ops.new_act(n, co, ho, wo, dtype, "cuda")
    rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
    stats, layers = _layers_of(ops, lambda: ops.conv2d_fwd_bnstats(xd, wp, y2, ops.conv_params(3, 2, 1, 1), g_.cuda(), b_.cuda(), rm, rv,
                                                                    0.03, 1e-3, ops.ACT_SILU, None, z))
    assert any(k[0].startswith("conv1x1_ws_kernel") and k[1].startswith("k3s2") for k in layers), list(layers)
    assert torch.equal(y2, y)
    assert_close(stats[2].cpu(), yref.mean((0, 2, 3)), 2e-2, "batch mean")
    zref = F.silu(F.batch_norm(yref, None, None, g_, b_, True, 0.03, 1e-3))
    assert_close(z.float().cpu(), zref, 2 * TOL[dtype], "BN + SiLU output")


# n, ci, h, w, co: dx is [n, ci, h, w], dy [n, co, ho, wo]; 4 * co must be whole slabs and <= 4 of them
CASES_S2_DGRAD = [
    (8, 32, 320, 320, 64),     # layer 1's data gradient: K = 256, 128 GEMM columns -> 4 sub-pixel classes x 32 channels
    (4, 32, 149, 135, 64),     # odd destination map: the last row / column of classes is cut
    (4, 24, 160, 128, 64),     # a class boundary inside a 64-column tile that is not a multiple of 32
    (8, 64, 160, 160, 128),    # layer 3's data gradient: K = 512 = 8 slabs, weights + 3-stage ring fill the LDS exactly
]


@pytest.mark.parametrize("case", CASES_S2_DGRAD)
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bnred"])
def test_stride2_dgrad_gathered_k_depth_to_space(ops, case, mode):
    """The stride-2 data gradient (2x2 form, [4 Ci][2][2][Co] weights) through the gather form with the depth-to-space store, plain /
    accumulating / with the BatchNorm backward sums of the block it completes -- against the implicit-GEMM launch (bit-equal
    stores, the same sums)."""
    import os
    n, ci, h, w, co = case
    dtype = torch.bfloat16
    conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(rnd((co, ci, 3, 3), 120, -0.2, 0.2))
    bank = ops.WeightBank([conv], [ci], dtype, "cuda")
    bank.pack()
    s2 = bank.dgrad_s2[0]
    ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
    gy = rnd((n, co, ho, wo), 121)
    gyd = to_dev(ops, gy, dtype)
    p = ops.conv_params(3, 2, 1, 1, accumulate=(mode == "accumulate"))
    base = rnd((n, ci, h, w), 122)
    red = red_ref = None
    if mode == "bnred":
        yb = to_dev(ops, rnd((n, ci, h, w), 123), dtype)
        st = torch.stack([torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5,
                          torch.randn(ci, device="cuda") * 0.1, torch.rand(ci, device="cuda") + 0.5])
        acc, _ = ops.bn_acc(ci, "cuda")
        acc_ref, _ = ops.bn_acc(ci, "cuda")
        red = ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ops.ACT_SILU, acc, ci, 0)])
        red_ref = ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ops.ACT_SILU, acc_ref, ci, 0)])
    dx = to_dev(ops, base, dtype)
    ops._lib.lib().dsn_pp_mode(0)                       # (the big-tile kernel of conv_pp.hip is tried first: tests/test_pp_gpu.py)
    try:
        _, layers = _layers_of(ops, lambda: ops.conv2d_dgrad_s2(gyd, s2, dx, p, red=red))
    finally:
        ops._lib.lib().dsn_pp_mode(1)
    assert any(k[0].startswith("conv1x1_ws_kernel") and k[1].startswith("k2s2") for k in layers), list(layers)
    ops._lib.lib().dsn_ws_mode(0, -1)                   # reference: the implicit-GEMM launch
    ops._lib.lib().dsn_pp_mode(0)
    try:
        dx_ref = to_dev(ops, base, dtype)
        _, layers = _layers_of(ops, lambda: ops.conv2d_dgrad_s2(gyd, s2, dx_ref, p, red=red_ref))
        assert any(k[0].startswith("igemm_kernel") for k in layers), list(layers)
    finally:
        ops._lib.lib().dsn_ws_mode(1, -1)
        ops._lib.lib().dsn_pp_mode(1)
    xr = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xr, q(conv.weight.detach().cpu(), dtype), None, 2, 1).backward(q(gy, dtype))
    want = xr.grad + (q(base, dtype) if mode == "accumulate" else 0)
    assert_close(dx.float().cpu(), want, 2 * TOL[dtype], f"dgrad_s2 gather {case} {mode}")
    assert_close(dx.float().cpu(), dx_ref.float().cpu(), 1e-2, "gather form vs implicit GEMM")
    if mode == "bnred":
        def fold(a):
            return a.view(torch.float64)[:8 * 2 * ci].view(8, 2, ci).sum(0).float()
        got, ref = fold(acc), fold(acc_ref)
        scale = float(ref.abs().max())
        assert scale > 0 and float((got - ref).abs().max()) <= 3e-3 * scale, float((got - ref).abs().max()) / scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 32, 160, 160, 64), (4, 32, 150, 134, 40)])
def test_stride2_forward_gathered_k_fused_epilogue(ops, case, dtype):
    """The inference form of the same launch: bias + SiLU in the epilogue, fp32 (K = 288 floats = 9 slabs) and bf16."""
    n, ci, h, w, co = case
    x = rnd((n, ci, h, w), 131)
    wt, b = rnd((co, ci, 3, 3), 132, -0.2, 0.2), rnd((co,), 133)
    xd = to_dev(ops, x, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
    y = ops.new_act(n, co, ho, wo, dtype, "cuda")
    ref = F.silu(F.conv2d(q(x, dtype), q(wt, dtype), b, 2, 1))
    _, layers = _layers_of(ops, lambda: ops.conv2d_fwd(xd, wp, b.cuda(), None, y, ops.conv_params(3, 2, 1, 1, act=ops.ACT_SILU)))
    assert any(k[0].startswith("conv1x1_ws_kernel") and k[1].startswith("k3s2") for k in layers), list(layers)
    assert_close(y.float().cpu(), ref, TOL[dtype], f"s2 fwd fused {case}")
