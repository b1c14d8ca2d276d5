"""BatchNorm backward sums formed in the epilogue of the input-gradient convolution that completes dz (include/desenet_hip.h:
dsn_bnred) on the MI355X: dx must be bit-identical to the plain dgrad, and the sums must equal what the stand-alone reduction
(dsn_bn_act_bwd_reduce) computes from that dx and the producer's y -- for every kernel the dgrad can take (one-trip 1x1, halo-tile
3x3, implicit GEMM incl. dilation, the stride-2 parity classes and the stride-2 depth-to-space form), with a shortcut residual,
with accumulation into an existing dx, with two segments (C3's [m(..) | cv2(x)] gradient) and with a channel-slice destination."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 3], ids=["default", "ws_extras_everywhere"])
def ws_mode(request):
    """1: the library's per-shape kernel choice; 3: every eligible data-gradient launch through the weights-stationary kernels
    with register-prefetched extras (conv_ws.hip) -- the default only picks them where they measured faster."""
    from desenet_amd import _lib
    L = _lib.lib()
    L.dsn_ws_mode(request.param, request.param)
    yield request.param
    L.dsn_ws_mode(1, 1)


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def _fold(acc, c):
    return acc.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)


CASES = [  # k, stride, dil, ci (dx channels), co (dy channels), h, w, segments [(c0, c1)], residual, accumulate
    (1, 1, 1, 64, 128, 20, 20, [(0, 64)], False, False),
    (1, 1, 1, 128, 64, 13, 11, [(0, 64), (64, 128)], True, False),
    (1, 1, 1, 192, 64, 16, 16, [(64, 128)], False, True),
    (3, 1, 1, 64, 64, 16, 24, [(0, 64)], False, False),
    (3, 1, 1, 128, 128, 13, 11, [(0, 128)], True, True),
    (3, 1, 2, 64, 64, 12, 12, [(0, 32), (32, 64)], False, False),
    (3, 1, 1, 48, 40, 9, 9, [(8, 48)], False, False),           # implicit GEMM (no whole slabs)
    (3, 2, 1, 64, 128, 16, 16, [(0, 64)], False, False),        # stride 2: parity classes
    (3, 2, 1, 32, 64, 17, 15, [(0, 32)], False, True),
    (5, 1, 1, 32, 16, 10, 10, [(0, 32)], False, False),
    # maps of >= 2048 pixels: the weights-stationary persistent kernels (conv_ws.hip) with their register-prefetched extras
    (1, 1, 1, 64, 128, 40, 40, [(0, 64)], False, False),
    (1, 1, 1, 128, 64, 33, 31, [(0, 64), (64, 128)], True, False),
    (1, 1, 1, 192, 64, 40, 40, [(64, 128)], False, True),
    (1, 1, 1, 40, 96, 37, 29, [(8, 40)], True, True),           # ragged tiles, partial slabs, residual + accumulate + sums
    (1, 1, 1, 32, 64, 64, 64, [(0, 32)], False, True),          # 128 x 32 tiles
    (3, 1, 1, 64, 64, 40, 48, [(0, 64)], False, False),
    (3, 1, 1, 64, 64, 33, 31, [(0, 32), (32, 64)], True, True),
    (3, 1, 2, 64, 64, 40, 40, [(0, 64)], True, False),
    (3, 1, 3, 64, 64, 40, 40, [(0, 64)], False, True),
    (3, 1, 1, 32, 32, 64, 64, [(0, 32)], True, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,dil,ci,co,h,w,segs,use_res,accumulate", CASES)
def test_dgrad_with_batchnorm_sums_in_its_epilogue(ws_mode, dtype, k, stride, dil, ci, co, h, w, segs, use_res, accumulate):
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
    desenet_amd.set_compute_dtype(dtype)
    try:
        n = 3
        pad = dil * (k // 2)
        ho, wo = ops.conv_out_hw(h, w, k, stride, pad, dil)
        wt = _rand((co, ci, k, k), torch.float32, 1, 0.1)
        wd = ops.pack_weight_dgrad(wt, dtype)
        dy = ops.as_act(_rand((n, co, ho, wo), dtype, 2))
        res = ops.as_act(_rand((n, ci, h, w), dtype, 3)) if use_res and stride == 1 else None
        base = ops.as_act(_rand((n, ci + 8, h, w), dtype, 4))          # dx lives in a channel slice of a wider buffer
        p = ops.conv_params(k, stride, pad, dil, accumulate=accumulate)
        dx_ref = base.clone()[:, :ci]
        ops.conv2d_dgrad(dy, wd, dx_ref, p, residual=res)
        # producers: one raw conv output + saved statistics per segment
        segments, refs = [], []
        for j, (c0, c1) in enumerate(segs):
            c = c1 - c0
            ybuf = ops.as_act(_rand((n, c + 16, h, w), dtype, 10 + j))
            y = ybuf[:, 8:8 + c]
            stats = torch.stack([torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") - 0.5,
                                 torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5])
            act = ACT_SILU if j % 2 == 0 else ACT_NONE
            acc_c = c + 24
            acc, _ = ops.bn_acc(acc_c, "cuda")
            segments.append((c0, c1, y, stats[0], stats[1], stats[2], stats[3], act, acc, acc_c, 16))
            ws, nb = ops.bn_acc(c, "cuda")
            ops.bn_act_bwd_reduce(dx_ref[:, c0:c1], y, stats[0], stats[1], stats[2], stats[3], act, ws)
            refs.append((ws, c, acc, acc_c))
        dx = base.clone()[:, :ci]
        ops.conv2d_dgrad(dy, wd, dx, p, residual=res, red=ops.bnred(segments))
        torch.cuda.synchronize()
        assert torch.equal(dx, dx_ref)
        for ws, c, acc, acc_c in refs:
            want = _fold(ws, c)
            got = _fold(acc, acc_c)
            assert float(got[:, :16].abs().max()) == 0 and float(got[:, 16 + c:].abs().max()) == 0     # only its own channels
            got = got[:, 16:16 + c]
            scale = float(want.abs().max())
            assert scale > 0
            assert float((got - want).abs().max()) <= 2e-5 * scale * (h * w * n) ** 0.5, float((got - want).abs().max()) / scale
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stride2_depth_to_space_dgrad_with_batchnorm_sums(dtype):
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU
    desenet_amd.set_compute_dtype(dtype)
    try:
        n, ci, co, h, w = 2, 32, 64, 24, 20
        conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
        bank = ops.WeightBank([conv], [ci], dtype, "cuda")
        bank.pack()
        s2 = bank.dgrad_s2[0]
        dy = ops.as_act(_rand((n, co, h // 2, w // 2), dtype, 2))
        p = ops.conv_params(3, 2, 1, 1)
        dx_ref = ops.conv2d_dgrad_s2(dy, s2, ops.new_act(n, ci, h, w, dtype, "cuda"), p)
        y = ops.as_act(_rand((n, ci, h, w), dtype, 5))
        stats = torch.stack([torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5,
                             torch.randn(ci, device="cuda") * 0.1, torch.rand(ci, device="cuda") + 0.5])
        ws, _ = ops.bn_acc(ci, "cuda")
        ops.bn_act_bwd_reduce(dx_ref, y, stats[0], stats[1], stats[2], stats[3], ACT_SILU, ws)
        acc, _ = ops.bn_acc(ci, "cuda")
        red = ops.bnred([(0, ci, y, stats[0], stats[1], stats[2], stats[3], ACT_SILU, acc, ci, 0)])
        dx = ops.conv2d_dgrad_s2(dy, s2, ops.new_act(n, ci, h, w, dtype, "cuda"), p, red=red)
        torch.cuda.synchronize()
        assert torch.equal(dx, dx_ref)
        want, got = _fold(ws, ci), _fold(acc, ci)
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) * (h * w * n) ** 0.5
    finally:
        desenet_amd.set_compute_dtype(torch.float32)
