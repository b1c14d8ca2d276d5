"""conv_pp.hip: the big-tile (16 x 16 pixels x 128 / 256 channels) ping-pong 3x3 kernel, forward and data gradient, against ATen
on the CPU and -- bit for bit -- against the kernels it replaces (same K order: slab, tap, 32-channel step), with every epilogue
option the training step uses: bias + activation, shortcut residual, accumulation, BatchNorm partial sums (forward) and the
BatchNorm backward sums of the block whose dz the data gradient completes.  Production grids (config 3 / 5 layer shapes, several
resident-block generations, three launches each) are what a synchronisation slip would need to show."""
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


@pytest.fixture(params=[3, 4, 5], ids=["bn256_where_possible", "bn128_two_blocks_per_cu", "bn128_one_block_per_cu"])
def pp_mode(request):
    from desenet_amd import _lib
    L = _lib.lib()
    L.dsn_pp_mode(request.param)
    yield request.param
    L.dsn_pp_mode(1)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g) * scale


def _fold(acc, c):
    return acc.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)


SHAPES = [  # n, ci, co, h, w
    (2, 64, 128, 32, 32),
    (1, 128, 128, 48, 40),        # ragged: 40 = 2 * 16 + 8
    (3, 64, 160, 19, 35),         # ragged both ways, a partial 128-channel tile
    (1, 192, 256, 32, 16),
    (2, 256, 512, 16, 16),
    (1, 64, 64, 16, 48),          # fewer channels than a tile
]


@pytest.mark.parametrize("n,ci,co,h,w", SHAPES)
def test_forward_and_dgrad_vs_aten_and_bit_exact_vs_the_other_kernels(pp_mode, n, ci, co, h, w):
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        x, wt = _rand((n, ci, h, w), 1), _rand((co, ci, 3, 3), 2, 0.1)
        gy = _rand((n, co, h, w), 3)
        q = lambda t: t.to(dt).float().cpu()
        ref = F.conv2d(q(x), q(wt), None, 1, 1)
        ref_dx = F.conv_transpose2d(q(gy), q(wt), None, 1, 1)
        xd, gd = ops.as_act(x.to(dt)), ops.as_act(gy.to(dt))
        wf, wd = ops.pack_weight_fwd(wt, dt), ops.pack_weight_dgrad(wt, dt)
        p = ops.conv_params(3, 1, 1, 1)
        outs = {}
        for mode in (pp_mode, 0):
            L.dsn_pp_mode(mode)
            y = ops.conv2d_fwd(xd, wf, None, None, ops.new_act(n, co, h, w, dt, "cuda"), p)
            dx = ops.conv2d_dgrad(gd, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p)
            torch.cuda.synchronize()
            outs[mode] = (y.clone(), dx.clone())
        y, dx = outs[pp_mode]
        e1 = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
        e2 = float((dx.float().cpu() - ref_dx).abs().max() / ref_dx.abs().max())
        assert e1 < 2e-2 and e2 < 2e-2, (e1, e2)
        assert torch.equal(y, outs[0][0]) and torch.equal(dx, outs[0][1])
    finally:
        L.dsn_pp_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 128, 32, 32), (1, 128, 256, 19, 35)])
def test_epilogue_options_bit_exact(pp_mode, n, ci, co, h, w):
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
        wt = _rand((co, ci, 3, 3), 2, 0.1)
        bias = _rand((co,), 4)
        res = ops.as_act(_rand((n, co, h, w), 5).to(dt))
        base = ops.as_act(_rand((n, co + 8, h, w), 6).to(dt))
        wf = ops.pack_weight_fwd(wt, dt)
        got = {}
        for mode in (pp_mode, 0):
            L.dsn_pp_mode(mode)
            yb = base.clone()
            ops.conv2d_fwd(x, wf, bias, res, yb[:, 8:], ops.conv_params(3, 1, 1, 1, act=ACT_SILU))
            ya = base.clone()
            ops.conv2d_dgrad(x, ops.pack_weight_dgrad(wt.transpose(0, 1).contiguous(), dt), ya[:, :co],
                             ops.conv_params(3, 1, 1, 1, accumulate=True), residual=res)
            torch.cuda.synchronize()
            got[mode] = (yb, ya)
        assert torch.equal(got[pp_mode][0], got[0][0])
        assert torch.equal(got[pp_mode][1], got[0][1])
        assert not torch.equal(got[pp_mode][0][:, 8:], base[:, 8:])
    finally:
        L.dsn_pp_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 128, 32, 32), (3, 128, 256, 19, 35)])
def test_batchnorm_sums_forward_and_backward(pp_mode, n, ci, co, h, w):
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        x = ops.as_act(_rand((n, ci, h, w), 1).to(dt))
        wt = _rand((co, ci, 3, 3), 2, 0.1)
        wf, wd = ops.pack_weight_fwd(wt, dt), ops.pack_weight_dgrad(wt, dt)
        p = ops.conv_params(3, 1, 1, 1)
        # forward: y + per-channel fp64 sums of y
        L.dsn_pp_mode(pp_mode)
        y = ops.new_act(n, co, h, w, dt, "cuda")
        acc, _ = ops.conv2d_fwd_acc(x, wf, y, p)
        torch.cuda.synchronize()
        got = _fold(acc, co)
        yf = y.float().double()
        want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
        # (the sums are taken from the fp32 accumulators BEFORE the bf16 rounding of y: compare at bf16 resolution)
        assert float((got - want).abs().max() / want.abs().max()) < 2e-3
        L.dsn_pp_mode(0)
        y0 = ops.new_act(n, co, h, w, dt, "cuda")
        acc0, _ = ops.conv2d_fwd_acc(x, wf, y0, p)
        torch.cuda.synchronize()
        assert torch.equal(y, y0)
        w0 = _fold(acc0, co)
        assert float((got - w0).abs().max()) <= 2e-5 * float(w0.abs().max()) * (h * w * n) ** 0.5
        # backward: dx bit-identical, sums equal to the stand-alone reduction's
        dy = ops.as_act(_rand((n, co, h, w), 3).to(dt))
        res = ops.as_act(_rand((n, ci, h, w), 7).to(dt))
        segs = [(0, ci // 2), (ci // 2, ci)]
        L.dsn_pp_mode(0)
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
        L.dsn_pp_mode(pp_mode)
        dx = ops.conv2d_dgrad(dy, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p, residual=res, red=ops.bnred(segments))
        torch.cuda.synchronize()
        assert torch.equal(dx, dx_ref)
        for ws, a, c in refs:
            want, got = _fold(ws, c), _fold(a, c)
            assert float(want.abs().max()) > 0
            assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) * (h * w * n) ** 0.5
    finally:
        L.dsn_pp_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


def test_production_grids_three_launches_each():
    """Config 3 / config 5 layer shapes under the DEFAULT selection (mode 1 picks this kernel for them): hundreds of blocks, more
    than one generation of resident blocks -- the regime in which an LDS-stage race would show as sporadic wrong patches."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    try:
        g = torch.Generator().manual_seed(5)
        q = lambda t: t.to(dt).float()
        for (n, ci, h, w, co) in [(8, 256, 80, 80, 128), (8, 128, 80, 80, 256), (4, 128, 160, 160, 128), (4, 256, 80, 80, 256),
                                  (2, 512, 160, 160, 256)]:
            x = torch.randn((n, ci, h, w), generator=g)
            wt = torch.randn((co, ci, 3, 3), generator=g) * 0.05
            ref = F.conv2d(q(x), q(wt), None, 1, 1)
            gy = torch.randn(tuple(ref.shape), generator=g)
            ref_dx = F.conv_transpose2d(q(gy), q(wt), None, 1, 1)
            xd, gd = ops.as_act(x.cuda().to(dt)), ops.as_act(gy.cuda().to(dt))
            wf, wd = ops.pack_weight_fwd(wt.cuda(), dt), ops.pack_weight_dgrad(wt.cuda(), dt)
            p = ops.conv_params(3, 1, 1, 1)
            for rep in range(3):
                y = ops.conv2d_fwd(xd, wf, None, None, ops.new_act(n, co, h, w, dt, "cuda"), p)
                dx = ops.conv2d_dgrad(gd, wd, ops.new_act(n, ci, h, w, dt, "cuda"), p)
                e1 = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
                e2 = float((dx.float().cpu() - ref_dx).abs().max() / ref_dx.abs().max())
                assert e1 < 2e-2 and e2 < 2e-2, ((n, ci, h, w, co), rep, e1, e2)
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


S2_SHAPES = [  # n, ci, co, h, w (the convolution's input map; dy is the stride-2 output map)
    (2, 64, 128, 64, 64),         # 256 GEMM columns: one 256-column tile (mode 3) or two 128-column tiles
    (1, 32, 64, 96, 80),          # ragged dy patches (48 x 40), 128 columns
    (3, 40, 64, 38, 70),          # 160 columns: a partial tile, parity boundaries inside it (40 channels per parity)
    (1, 128, 256, 33, 31),        # odd destination map: the last row / column of parities is cut; 4 slabs
    (2, 256, 512, 32, 32),        # 8 slabs
    (8, 32, 64, 160, 160),        # 1600 dy tiles of 32 pixels: the grids on which the gather form's sums were found wrong (conv_ws.hip;
    (2, 32, 64, 320, 320),        # mode 0 runs that form here, now with a full drain in front of its extras)
    (16, 32, 64, 128, 128),
]


@pytest.mark.parametrize("n,ci,co,h,w", S2_SHAPES)
@pytest.mark.parametrize("opt", ["plain", "accumulate", "bnred"])
@pytest.mark.parametrize("mode", [2, 3], ids=["bn128", "bn256_where_possible"])
def test_stride2_data_gradient_2x2_form(mode, opt, n, ci, co, h, w):
    """The 3x3 / stride-2 data gradient on the ping-pong kernel (NT = 4: 2x2 taps over dy, [4 Ci][2][2][Co] weights, depth-to-space
    store) against ATen's convolution input gradient and against the kernels it replaces (dsn_pp_mode 0), plain / accumulating /
    with the BatchNorm-backward sums of the block whose dz it completes."""
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
        with torch.no_grad():
            conv.weight.copy_(_rand((co, ci, 3, 3), 2, 0.1))
        bank = ops.WeightBank([conv], [ci], dt, "cuda")
        bank.pack()
        s2 = bank.dgrad_s2[0]
        ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
        gy = _rand((n, co, ho, wo), 3)
        q = lambda t: t.to(dt).float().cpu()
        xr = torch.zeros(n, ci, h, w, requires_grad=True)
        F.conv2d(xr, q(conv.weight.detach()), None, 2, 1).backward(q(gy))
        gd = ops.as_act(gy.to(dt))
        base = _rand((n, ci, h, w), 4)
        p = ops.conv_params(3, 2, 1, 1, accumulate=(opt == "accumulate"))
        got = {}
        for md in (mode, 0):
            L.dsn_pp_mode(md)
            dx = ops.as_act(base.to(dt))
            red, acc = None, None
            if opt == "bnred":
                yb = ops.as_act(_rand((n, ci, h, w), 5).to(dt))
                g_ = torch.Generator(device="cuda").manual_seed(9)
                st = torch.stack([torch.rand(ci, device="cuda", generator=g_) + 0.5, torch.rand(ci, device="cuda", generator=g_) - 0.5,
                                  torch.randn(ci, device="cuda", generator=g_) * 0.1, torch.rand(ci, device="cuda", generator=g_) + 0.5])
                acc, _ = ops.bn_acc(ci, "cuda")
                red = ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ACT_SILU, acc, ci, 0)])
            if md:
                ops.profile_enable(True)
            ops.conv2d_dgrad_s2(gd, s2, dx, p, red=red)
            torch.cuda.synchronize()
            if md:
                prof = ops.profile_collect()
                ops.profile_enable(False)
                assert any("conv3x3_pp_kernel" in str(k) for k in prof), list(prof)
            if opt == "bnred":      # the stand-alone reduction over the dz just written: what the fused sums must equal
                ws, _ = ops.bn_acc(ci, "cuda")
                ops.bn_act_bwd_reduce(dx, yb, st[0], st[1], st[2], st[3], ACT_SILU, ws)
                torch.cuda.synchronize()
                a, b = _fold(acc, ci), _fold(ws, ci)
                assert float(b.abs().max()) > 0
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) * (h * w * n) ** 0.5, (md, float((a - b).abs().max() / b.abs().max()))
            got[md] = (dx, acc)
        want = xr.grad + (q(base) if opt == "accumulate" else 0)
        e = float((got[mode][0].float().cpu() - want).abs().max() / want.abs().max())
        assert e < 2e-2, e
        e0 = float((got[mode][0].float() - got[0][0].float()).abs().max() / want.abs().max())
        assert e0 < 1e-2, e0
    finally:
        ops.profile_enable(False)
        L.dsn_pp_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


WGRAD_SHAPES = [  # n, ci, co, h, w
    (2, 128, 128, 32, 32),
    (1, 256, 128, 48, 40),        # ragged columns: 40 = 2 * 16 + 8
    (3, 128, 256, 19, 35),        # ragged both ways
    (2, 128, 128, 16, 16),        # one patch per image
    (1, 384, 128, 33, 17),        # three ci tiles
]


@pytest.mark.parametrize("n,ci,co,h,w", WGRAD_SHAPES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_wgrad_kernel_row_ping_pong_vs_aten_and_the_other_kernels(n, ci, co, h, w, accumulate):
    """wgrad.hip kind 5 (a block = one kernel row x 128 x 128 tile over a range of 16 x 16 patches; the three taps share the staged
    x rows): single-layer and queued launches against ATen on the CPU (2e-2, bf16 operands) and against the kernels it replaces
    (same products, fp32 summation order differs: 1e-4), OIHW output with and without accumulation, split-K slabs."""
    import desenet_amd
    from desenet_amd import _lib, hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    L = _lib.lib()
    try:
        x, gy = _rand((n, ci, h, w), 1), _rand((n, co, h, w), 2)
        q = lambda t: t.to(dt).float().cpu()
        wq = torch.zeros(co, ci, 3, 3, requires_grad=True)
        F.conv2d(q(x), wq, None, 1, 1).backward(q(gy))
        base = _rand((co, ci, 3, 3), 3)
        want = wq.grad + (base.cpu() if accumulate else 0)
        xd, gd = ops.as_act(x.to(dt)), ops.as_act(gy.to(dt))
        p = ops.conv_params(3, 1, 1, 1, accumulate=accumulate)
        got = {}
        for mode in (2, 0):
            L.dsn_wgrad_pp_mode(mode)
            g1 = base.clone() if accumulate else torch.zeros_like(base)
            ops.conv2d_wgrad(xd, gd, g1, ci, p, oihw=True)
            queue = ops.WgradQueue(torch.device("cuda", torch.cuda.current_device()))
            g2 = base.clone() if accumulate else torch.zeros_like(base)
            ops.conv2d_wgrad(xd, gd, g2, ci, p, oihw=True, queue=queue)
            assert queue.n == 1
            queue.flush()
            torch.cuda.synchronize()
            got[mode] = (g1, g2)
        for g in got[2]:
            e = float((g.cpu() - want).abs().max() / want.abs().max())
            assert e < 2e-2, e
        scale = float(got[0][0].abs().max())
        assert float((got[2][0] - got[0][0]).abs().max()) <= 1e-4 * scale
        assert float((got[2][1] - got[2][0]).abs().max()) <= 1e-5 * scale       # queued == single-layer (same kernel, same split)
    finally:
        L.dsn_wgrad_pp_mode(1)
        desenet_amd.set_compute_dtype(torch.float32)


def test_wgrad_kernel_row_production_shapes():
    """Config 3 / config 5 layer shapes under the default selection, three launches each (several resident-block generations)."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    try:
        g = torch.Generator().manual_seed(7)
        q = lambda t: t.to(dt).float()
        for (n, ci, h, w, co) in [(8, 256, 80, 80, 128), (4, 128, 160, 160, 128), (4, 256, 80, 80, 256)]:
            x = torch.randn((n, ci, h, w), generator=g)
            gy = torch.randn((n, co, h, w), generator=g)
            wq = torch.zeros(co, ci, 3, 3, requires_grad=True)
            F.conv2d(q(x), wq, None, 1, 1).backward(q(gy))
            xd, gd = ops.as_act(x.cuda().to(dt)), ops.as_act(gy.cuda().to(dt))
            for rep in range(3):
                gw = torch.zeros((co, ci, 3, 3), device="cuda")
                ops.conv2d_wgrad(xd, gd, gw, ci, ops.conv_params(3, 1, 1, 1), oihw=True)
                e = float((gw.cpu() - wq.grad).abs().max() / wq.grad.abs().max())
                assert e < 2e-2, ((n, ci, h, w, co), rep, e)
    finally:
        desenet_amd.set_compute_dtype(torch.float32)
