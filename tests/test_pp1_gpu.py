"""conv_pp.hip, 1x1 form: block tiles of 256 consecutive pixels x 128 / 256 channels, both operands streamed through the LDS-DMA
ring, forward and data gradient -- against ATen on the CPU and bit for bit against the kernels it replaces (same K order: 32-channel
steps), with the epilogue options of the training step (bias + activation, residual, accumulation, channel-slice operands,
BatchNorm partial sums forward, BatchNorm backward sums in the data gradient) and on production grids."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["staged_epilogue", "register_epilogue"])
def _epilogue_form(request):
    """Every test of this file runs with both epilogue forms of the ping-pong kernels (dsn_pp_dir)."""
    from desenet_amd import _lib
    L = _lib.lib()
    L.dsn_pp_dir(request.param)
    yield
    L.dsn_pp_dir(0)


@pytest.fixture(params=[2, 3], ids=["bn128", "bn256_where_possible"])
def pp1_mode(request):
    from desenet_amd import _lib
    L = _lib.lib()
    L.dsn_pp1_mode(request.param)
    yield request.param
    L.dsn_pp1_mode(1)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g) * scale


def _fold(acc, c):
    return acc.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)


SHAPES = [  # n, ci, co, h, w
    (2, 64, 128, 32, 32),
    (1, 96, 128, 48, 40),         # three k-halves: the ring is not full once
    (3, 256, 160, 19, 35),        # ragged pixel tail (1995 pixels), a partial 128-channel tile
    (1, 224, 256, 33, 16),        # seven k-halves: one past the six-stage ring
    (2, 512, 512, 16, 16),
    (1, 32, 64, 16, 48),          # one k-half, fewer channels than a tile
    (1, 1024, 256, 7, 9),         # 63 pixels: one partial tile, long K
]


@pytest.mark.parametrize("n,ci,co,h,w", SHAPES)
def test_forward_and_dgrad_vs_aten_and_bit_exact_vs_the_other_kernels(pp1_mode, n, ci, co, h, w):
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        x, wt = _rand((n, ci, h, w), 1), _rand((co, ci, 1, 1), 2, 0.1)
        gy = _rand((n, co, h, w), 3)
        q = lambda t: t.to(dt).float().cpu()
        ref = F.conv2d(q(x), q(wt))
        ref_dx = F.conv_transpose2d(q(gy), q(wt))
        xd, gd = ops.as_act(x.to(dt)), ops.as_act(gy.to(dt))
        wf, wd = ops.pack_weight_fwd(wt, dt), ops.pack_weight_dgrad(wt, dt)
        p = ops.conv_params(1, 1, 0, 1)
        outs = {}
        for mode in (pp1_mode, 0):
            L.dsn_pp1_mode(mode)
            y = ops.conv2d_fwd(xd, wf, None, None, ops.new_act(n, co, h, w, dt, "cuda"), p)
            dx = ops.conv2d_dgrad(gd, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p)
            torch.cuda.synchronize()
            outs[mode] = (y.clone(), dx.clone())
        y, dx = outs[pp1_mode]
        e1 = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
        e2 = float((dx.float().cpu() - ref_dx).abs().max() / ref_dx.abs().max())
        assert e1 < 2e-2 and e2 < 2e-2, (e1, e2)
        assert torch.equal(y, outs[0][0]) and torch.equal(dx, outs[0][1])
    finally:
        L.dsn_pp1_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 128, 32, 32), (1, 128, 256, 19, 35)])
def test_epilogue_options_bit_exact(pp1_mode, n, ci, co, h, w):
    """bias + SiLU + residual (fused inference form), accumulate (fan-in gradients), destination / source as channel slices."""
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        xw = ops.as_act(_rand((n, ci + 64, h, w), 1).to(dt))
        x = xw[:, 64:]                                                 # source = channel slice (ldc > C)
        wt = _rand((co, ci, 1, 1), 2, 0.1)
        bias = _rand((co,), 4)
        res = ops.as_act(_rand((n, co, h, w), 5).to(dt))
        base = ops.as_act(_rand((n, co + 8, h, w), 6).to(dt))
        wf = ops.pack_weight_fwd(wt, dt)
        got = {}
        for mode in (pp1_mode, 0):
            L.dsn_pp1_mode(mode)
            yb = base.clone()
            ops.conv2d_fwd(x, wf, bias, res, yb[:, 8:], ops.conv_params(1, 1, 0, 1, act=ACT_SILU))
            ya = base.clone()
            ops.conv2d_dgrad(x, ops.pack_weight_dgrad(wt.transpose(0, 1).contiguous(), dt), ya[:, :co],
                             ops.conv_params(1, 1, 0, 1, accumulate=True), residual=res)
            torch.cuda.synchronize()
            got[mode] = (yb, ya)
        assert torch.equal(got[pp1_mode][0], got[0][0])
        assert torch.equal(got[pp1_mode][1], got[0][1])
        assert not torch.equal(got[pp1_mode][0][:, 8:], base[:, 8:])
    finally:
        L.dsn_pp1_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 128, 32, 32), (3, 128, 256, 19, 35)])
def test_batchnorm_sums_forward_and_backward(pp1_mode, n, ci, co, h, w):
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        x = ops.as_act(_rand((n, ci, h, w), 1).to(dt))
        wt = _rand((co, ci, 1, 1), 2, 0.1)
        wf, wd = ops.pack_weight_fwd(wt, dt), ops.pack_weight_dgrad(wt, dt)
        p = ops.conv_params(1, 1, 0, 1)
        L.dsn_pp1_mode(pp1_mode)
        y = ops.new_act(n, co, h, w, dt, "cuda")
        acc, _ = ops.conv2d_fwd_acc(x, wf, y, p)
        torch.cuda.synchronize()
        got = _fold(acc, co)
        yf = y.float().double()
        want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
        assert float((got - want).abs().max() / want.abs().max()) < 2e-3
        L.dsn_pp1_mode(0)
        y0 = ops.new_act(n, co, h, w, dt, "cuda")
        acc0, _ = ops.conv2d_fwd_acc(x, wf, y0, p)
        torch.cuda.synchronize()
        assert torch.equal(y, y0)
        w0 = _fold(acc0, co)
        assert float((got - w0).abs().max()) <= 2e-5 * float(w0.abs().max()) * (h * w * n) ** 0.5
        dy = ops.as_act(_rand((n, co, h, w), 3).to(dt))
        res = ops.as_act(_rand((n, ci, h, w), 7).to(dt))
        segs = [(0, ci // 2), (ci // 2, ci)]
        L.dsn_pp1_mode(0)
        dx_ref = ops.conv2d_dgrad(dy, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p, residual=res)
        segments, refs = [], []
        for j, (c0, c1) in enumerate(segs):
            c = c1 - c0
            yseg = ops.as_act(_rand((n, c, h, w), 10 + j).to(dt))
            stats = torch.stack([torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") - 0.5,
                                 torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5])
            act = ACT_SILU if j == 0 else ACT_NONE
            a, _ = ops.bn_acc(c, "cuda")
            segments.append((c0, c1, yseg, stats[0], stats[1], stats[2], stats[3], act, a, c, 0))
            ws, _ = ops.bn_acc(c, "cuda")
            ops.bn_act_bwd_reduce(dx_ref[:, c0:c1], yseg, stats[0], stats[1], stats[2], stats[3], act, ws)
            refs.append((ws, a, c))
        L.dsn_pp1_mode(pp1_mode)
        dx = ops.conv2d_dgrad(dy, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p, residual=res, red=ops.bnred(segments))
        torch.cuda.synchronize()
        assert torch.equal(dx, dx_ref)
        for ws, a, c in refs:
            want, got = _fold(ws, c), _fold(a, c)
            assert float(want.abs().max()) > 0
            assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) * (h * w * n) ** 0.5
    finally:
        L.dsn_pp1_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


def test_production_grids_three_launches_each(pp1_mode):
    """Config 3 / config 5 1x1 layer shapes: hundreds of blocks, several generations of resident blocks."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    try:
        g = torch.Generator().manual_seed(5)
        q = lambda t: t.to(dt).float()
        for (n, ci, h, w, co) in [(4, 512, 80, 80, 512), (4, 1024, 40, 40, 1024), (2, 512, 160, 160, 256), (4, 256, 80, 80, 256),
                                  (8, 256, 40, 40, 256)]:
            x = torch.randn((n, ci, h, w), generator=g)
            wt = torch.randn((co, ci, 1, 1), generator=g) * 0.05
            ref = F.conv2d(q(x), q(wt))
            gy = torch.randn(tuple(ref.shape), generator=g)
            ref_dx = F.conv_transpose2d(q(gy), q(wt))
            xd, gd = ops.as_act(x.cuda().to(dt)), ops.as_act(gy.cuda().to(dt))
            wf, wd = ops.pack_weight_fwd(wt.cuda(), dt), ops.pack_weight_dgrad(wt.cuda(), dt)
            p = ops.conv_params(1, 1, 0, 1)
            for rep in range(3):
                y = ops.conv2d_fwd(xd, wf, None, None, ops.new_act(n, co, h, w, dt, "cuda"), p)
                dx = ops.conv2d_dgrad(gd, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p)
                e1 = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
                e2 = float((dx.float().cpu() - ref_dx).abs().max() / ref_dx.abs().max())
                assert e1 < 2e-2 and e2 < 2e-2, ((n, ci, h, w, co), rep, e1, e2)
    finally:
        desenet_amd.set_compute_dtype(torch.float32)
