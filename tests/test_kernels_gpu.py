"""Kernel-level parity: every C-ABI entry point (through desenet_amd.hip_ops) against stock PyTorch-CPU fp32 ops --
the same ops the oracle is made of.  Tolerances: fp32 1e-3 relative (BASELINE.json), bf16 2e-2 (8-bit mantissa inputs,
fp32 accumulate; compared with an fp32 reference computed from the SAME bf16-rounded inputs where noted)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from desenet_amd import hip_ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip_ops


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(shape, generator=g)


def to_dev(ops, x, dtype):
    """CPU NCHW fp32 -> device NHWC-backed logical-NCHW tensor of `dtype`."""
    n, c, h, w = x.shape
    t = ops.new_act(n, c, h, w, dtype, "cuda")
    t.copy_(x)
    return t


def q(x, dtype):
    """What the device actually sees: x rounded to dtype, as fp32."""
    return x.to(dtype).float()


CONV_CASES = [
    # n, ci, h, w, co, k, s, pad, dil
    (2, 16, 9, 7, 24, 1, 1, 0, 1),
    (2, 8, 10, 6, 16, 3, 1, 1, 1),
    (2, 8, 11, 7, 16, 3, 2, 1, 1),
    (1, 32, 20, 20, 32, 3, 1, 2, 2),
    (1, 16, 13, 17, 8, 3, 1, 3, 3),
    (2, 64, 16, 16, 128, 1, 1, 0, 1),
    (1, 128, 24, 24, 256, 3, 1, 1, 1),
    (2, 256, 8, 8, 33, 1, 1, 0, 1),
    (1, 128, 40, 40, 2, 1, 1, 0, 1),
    (1, 12, 16, 16, 32, 3, 1, 1, 1),     # Focus conv (Ci = 12)
    (3, 40, 5, 5, 72, 3, 2, 1, 1),
    (1, 64, 80, 80, 64, 3, 1, 1, 1),
    (4, 128, 1, 1, 128, 1, 1, 0, 1),     # FFM attention / 1x1 maps
    (1, 6, 7, 5, 10, 3, 1, 1, 1),        # scalar-load path (Ci % 4 != 0)
    (2, 192, 12, 12, 64, 3, 1, 1, 1),    # long K, few tiles: in-block split-K with an ODD chunk count (27 bf16 chunks)
    (1, 128, 16, 16, 64, 3, 1, 2, 2),    # the same path, dilated taps
    # conv3x3.hip (halo-tile kernel: 3x3 / stride 1, whole 128-byte channel slabs, maps that are multiples of 8)
    (2, 64, 16, 24, 96, 3, 1, 1, 1),     # two images, ragged output-channel tile
    (2, 64, 16, 16, 64, 3, 1, 3, 3),     # dilation 3 (14 x 14 halo)
    (1, 256, 8, 16, 128, 3, 1, 1, 1),    # four bf16 slabs: halo prefetch of the next slab
    (3, 128, 8, 8, 32, 3, 1, 2, 2),      # one patch per image, narrow output
    # wgrad.hip 128 x 128 tiles (bf16, whole 128-channel blocks, >= 16384 output pixels, below the all-taps threshold)
    (4, 128, 64, 64, 256, 1, 1, 0, 1),
    (4, 256, 66, 64, 128, 3, 1, 1, 1),   # ragged last pixel range
    (4, 128, 130, 128, 128, 3, 2, 1, 1), # stride 2: the input cursor is re-derived per chunk
    (4, 128, 64, 64, 128, 3, 1, 2, 2),   # dilation
    # wgrad.hip halo-tile all-taps kernel (bf16, same-size 3x3, >= 32768 output pixels): 4 x 8 patches, ragged at both borders
    (8, 64, 70, 67, 96, 3, 1, 1, 1),
    (2, 16, 130, 131, 32, 3, 1, 1, 1),
    (2, 64, 130, 131, 64, 3, 1, 2, 2),   # dilation 2: 8 x 12 halo
    (2, 32, 140, 128, 32, 3, 1, 3, 3),   # dilation 3: 10 x 14 halo (RFB2)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(ops, case, dtype):
    n, ci, h, w, co, k, s, pad, dil = case
    x, wt, b = rnd((n, ci, h, w), 1), rnd((co, ci, k, k), 2, -0.3, 0.3), rnd((co,), 3)
    ref = F.silu(F.conv2d(q(x, dtype), q(wt, dtype), b, s, pad, dil))
    ho, wo = ref.shape[2:]
    res = rnd((n, co, ho, wo), 4)
    ref = ref + q(res, dtype)
    xd, rd = to_dev(ops, x, dtype), to_dev(ops, res, dtype)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    y = ops.new_act(n, co, ho, wo, dtype, "cuda")
    ops.conv2d_fwd(xd, wp, b.cuda(), rd, y, ops.conv_params(k, s, pad, dil, act=ops.ACT_SILU))
    assert_close(y.float().cpu(), ref, TOL[dtype], f"conv {case}")
    if dtype == torch.float32:      # element-wise as well: |a-b| <= 1e-3 |b| + 1e-3 rms(b) for EVERY output (tests/util.py)
        from tests.util import elem_err
        assert elem_err(y.cpu(), ref) <= 1.0, (f"conv {case} element-wise", elem_err(y.cpu(), ref))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_slices_and_accumulate(ops, dtype):
    """Reads a channel slice of a wider buffer, writes into a slice of a concat buffer, then accumulates on top."""
    n, ci, h, w, co = 2, 16, 12, 10, 24
    big_in = rnd((n, 48, h, w), 5)
    wt = rnd((co, ci, 3, 3), 6, -0.3, 0.3)
    xin = to_dev(ops, big_in, dtype)
    ref = F.conv2d(q(big_in[:, 16:32], dtype), q(wt, dtype), None, 1, 1)
    cat = ops.new_act(n, 64, h, w, dtype, "cuda", zero=True)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype)
    ops.conv2d_fwd(xin[:, 16:32], wp, None, None, cat[:, 8:32], ops.conv_params(3))
    assert_close(cat[:, 8:32].float().cpu(), ref, TOL[dtype], "slice write")
    assert float(cat[:, :8].abs().max()) == 0 and float(cat[:, 32:].abs().max()) == 0, "wrote outside the slice"
    ops.conv2d_fwd(xin[:, 16:32], wp, None, None, cat[:, 8:32], ops.conv_params(3, accumulate=True))
    assert_close(cat[:, 8:32].float().cpu(), 2 * ref, 2 * TOL[dtype], "accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(ops, case, dtype):
    n, ci, h, w, co, k, s, pad, dil = case
    x = rnd((n, ci, h, w), 7).requires_grad_(True)
    wt = rnd((co, ci, k, k), 8, -0.3, 0.3).requires_grad_(True)
    xq, wq = q(x.detach(), dtype).requires_grad_(True), q(wt.detach(), dtype).requires_grad_(True)
    y = F.conv2d(xq, wq, None, s, pad, dil)
    gy = rnd(tuple(y.shape), 9)
    y.backward(q(gy, dtype))
    gd = to_dev(ops, gy, dtype)
    p = ops.conv_params(k, s, pad, dil)
    dx = ops.new_act(n, ci, h, w, dtype, "cuda")
    ops.conv2d_dgrad(gd, ops.pack_weight_dgrad(wt.detach().cuda(), dtype), dx, p)
    assert_close(dx.float().cpu(), xq.grad, TOL[dtype], f"dgrad {case}")
    dwp = torch.zeros((co, k, k, ci), dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad(to_dev(ops, x.detach(), dtype), gd, dwp, ci, p)
    dw = ops.unpack_wgrad(dwp, (co, ci, k, k), ci)
    assert_close(dw.cpu(), wq.grad, TOL[dtype], f"wgrad {case}")
    # straight into an OIHW gradient, accumulating on top of an existing value
    g = torch.ones((co, ci, k, k), device="cuda")
    ops.conv2d_wgrad(to_dev(ops, x.detach(), dtype), gd, g, ci, ops.conv_params(k, s, pad, dil, accumulate=True), oihw=True)
    assert_close(g.cpu() - 1.0, wq.grad, 2 * TOL[dtype], f"wgrad oihw+accumulate {case}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_large_reduction_and_padding(ops, dtype):
    """Split-K path (many pixel ranges) and a padded input-channel axis (Focus: 12 -> 16)."""
    n, ci, h, w, co = 4, 12, 64, 64, 32
    x, gy = rnd((n, ci, h, w), 10), rnd((n, co, h, w), 11)
    wq = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(q(x, dtype), wq, None, 1, 1).backward(q(gy, dtype))
    xpad = torch.cat([x, torch.zeros(n, 4, h, w)], 1)
    dwp = torch.zeros((co, 3, 3, 16), dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad(to_dev(ops, xpad, dtype), to_dev(ops, gy, dtype), dwp, 16, ops.conv_params(3))
    dw = ops.unpack_wgrad(dwp, (co, ci, 3, 3), 16)
    assert_close(dw.cpu(), wq.grad, TOL[dtype], "wgrad split-K")
    assert float(dwp[..., 12:].abs().max()) == 0
    g = torch.zeros((co, ci, 3, 3), device="cuda")
    ops.conv2d_wgrad(to_dev(ops, xpad, dtype), to_dev(ops, gy, dtype), g, ci, ops.conv_params(3), oihw=True)
    assert_close(g.cpu(), wq.grad, TOL[dtype], "wgrad split-K, OIHW out, padded x")


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_queue_matches_single_layer(ops, dtype):
    """dsn_conv2d_wgrad_plan / _plan_finish / _run: several layers' weight gradients in three grouped launches give the same
    numbers as one dsn_conv2d_wgrad per layer -- per-tap (64 and, bf16, 128-wide tiles) and (bf16, >= 32k pixels) all-taps blocks, split-K slabs and direct
    writes, accumulation into an existing gradient, and a layer the queue must refuse (odd channel count)."""
    layers = [  # n, ci, h, w, co, k, s
        (2, 16, 24, 24, 32, 3, 1), (8, 32, 72, 72, 32, 3, 1), (2, 64, 20, 20, 128, 1, 1), (1, 256, 8, 8, 256, 3, 1),
        (2, 32, 40, 40, 64, 3, 2), (2, 64, 10, 10, 33, 1, 1), (4, 128, 64, 64, 256, 1, 1), (4, 256, 72, 64, 128, 3, 1),
    ]
    queue = ops.WgradQueue(torch.device("cuda", torch.cuda.current_device()))
    pending = []
    for i, (n, ci, h, w, co, k, s) in enumerate(layers):
        pad = k // 2
        ho, wo = ops.conv_out_hw(h, w, k, s, pad, 1)
        x, gy = to_dev(ops, rnd((n, ci, h, w), 100 + i), dtype), to_dev(ops, rnd((n, co, ho, wo), 200 + i), dtype)
        base = rnd((co, ci, k, k), 300 + i).cuda()
        p = ops.conv_params(k, s, pad, 1, accumulate=True)
        ref = base.clone()
        ops.conv2d_wgrad(x, gy, ref, ci, p, oihw=True)
        got = base.clone()
        ops.conv2d_wgrad(x, gy, got, ci, p, oihw=True, queue=queue)
        pending.append((got, ref, (n, ci, h, w, co, k, s)))
    assert queue.n == len(layers) - 1, "the 33-channel layer must take the single-layer path"
    queue.flush()
    torch.cuda.synchronize()
    for got, ref, shape in pending:
        assert_close(got.cpu(), ref.cpu(), 1e-5 if dtype == torch.float32 else 1e-4, f"queued wgrad {shape}")
    assert queue.n == 0 and not queue.keep


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 9, 7, 24), (4, 64, 20, 20, 64), (8, 16, 64, 64, 32), (2, 64, 16, 24, 96)])
def test_conv_bn_act_fused_statistics(ops, shape, dtype):
    """dsn_conv2d_fwd_bnacc + dsn_bn_act_fwd_acc (statistics out of the conv epilogue, folded in the prologue of the BN + act
    kernel) against conv -> BatchNorm2d(train) -> SiLU from ATen; accumulator slots come from the per-step arena."""
    n, ci, h, w, co = shape
    x, wt = rnd((n, ci, h, w), 30), rnd((co, ci, 3, 3), 31, -0.2, 0.2)
    gamma, beta = rnd((co,), 32, 0.5, 1.5), rnd((co,), 33, -0.1, 0.1)
    rm, rv = rnd((co,), 34, -0.1, 0.1), rnd((co,), 35, 0.5, 1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.conv2d(q(x, dtype), q(wt, dtype), None, 1, 1)
    z_ref = F.silu(F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3))
    ops.bn_arena_begin("cuda")
    xd = to_dev(ops, x, dtype)
    y, z = ops.new_act(n, co, h, w, dtype, "cuda"), ops.new_act(n, co, h, w, dtype, "cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    for _ in range(2):      # twice: consecutive slots of the arena, running statistics advance twice
        scale, shift, mean, rstd = ops.conv2d_fwd_bnstats(xd, ops.pack_weight_fwd(wt.cuda(), dtype), y, ops.conv_params(3),
                                                          gamma.cuda(), beta.cuda(), rmd, rvd, 0.03, 1e-3, ops.ACT_SILU,
                                                          None, z)
    F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3)
    assert_close(z.float().cpu(), z_ref, TOL[dtype], "conv+bn+silu")
    assert_close(mean.cpu(), y_ref.mean((0, 2, 3)), 1e-3 if dtype == torch.float32 else 5e-3, "batch mean")
    assert_close(rmd.cpu(), rm_ref, 1e-3 if dtype == torch.float32 else 5e-3, "running_mean after two steps")
    assert_close(rvd.cpu(), rv_ref, 1e-3 if dtype == torch.float32 else 5e-3, "running_var after two steps")


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_multi_and_backward(ops, dtype):
    """SPP's three pools in one launch (separable LDS kernel) and their gradients in one scatter pass, vs ATen."""
    n, c, h, w = 2, 32, 20, 20
    x = rnd((n, c, h, w), 40)
    xq = q(x, dtype).requires_grad_(True)
    ks = [5, 9, 13]
    refs = [F.max_pool2d(xq, k, 1, k // 2) for k in ks]
    gys = [rnd((n, c, h, w), 41 + i) for i in range(3)]
    sum(((r * q(g, dtype)).sum() for r, g in zip(refs, gys))).backward()
    xd = to_dev(ops, x, dtype)
    ys = [ops.new_act(n, c, h, w, dtype, "cuda") for _ in ks]
    idxs = [torch.empty((n, h, w, c), dtype=torch.int32, device="cuda") for _ in ks]
    ops.maxpool_s1_multi(xd, ys, ks, idxs)
    for y, r, k in zip(ys, refs, ks):
        assert torch.equal(y.float().cpu(), r.detach()), f"maxpool k={k}"
    dx = ops.new_act(n, c, h, w, dtype, "cuda")
    ops.maxpool_s1_bwd_multi([to_dev(ops, g, dtype) for g in gys], idxs, ks, dx)
    assert_close(dx.float().cpu(), xq.grad, TOL[dtype], "maxpool multi bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 24, 24, 32), (1, 64, 17, 23, 32), (8, 32, 64, 64, 64), (2, 256, 10, 10, 512)])
def test_stride2_dgrad_depth_to_space(ops, shape, dtype):
    """dsn_conv2d_dgrad_s2 (one 2x2 stride-1 conv over dy + depth-to-space store, weights from the WeightBank's out_dgrad_s2
    layout) against ATen's conv input gradient -- even and odd map sizes, plain and accumulating stores."""
    n, ci, h, w, co = shape
    conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(rnd((co, ci, 3, 3), 70, -0.2, 0.2))
    bank = ops.WeightBank([conv], [ci], dtype, "cuda")
    bank.pack()
    assert bank.dgrad_s2[0] is not None
    ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
    gy = rnd((n, co, ho, wo), 71)
    xr = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xr, q(conv.weight.detach().cpu(), dtype), None, 2, 1).backward(q(gy, dtype))
    gyd = to_dev(ops, gy, dtype)
    dx = ops.new_act(n, ci, h, w, dtype, "cuda")
    ops.conv2d_dgrad_s2(gyd, bank.dgrad_s2[0], dx, ops.conv_params(3, 2, 1, 1))
    assert_close(dx.float().cpu(), xr.grad, TOL[dtype], "dgrad_s2")
    ref = ops.new_act(n, ci, h, w, dtype, "cuda")
    ops.conv2d_dgrad(gyd, bank.dgrad[0], ref, ops.conv_params(3, 2, 1, 1))
    assert_close(dx.float().cpu(), ref.float().cpu(), 1e-5 if dtype == torch.float32 else 1e-2, "dgrad_s2 vs parity-class dgrad")
    base = rnd((n, ci, h, w), 72)
    acc = to_dev(ops, base, dtype)
    ops.conv2d_dgrad_s2(gyd, bank.dgrad_s2[0], acc, ops.conv_params(3, 2, 1, 1, accumulate=True))
    assert_close(acc.float().cpu(), xr.grad + q(base, dtype), 2 * TOL[dtype], "dgrad_s2 accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_weights(ops, dtype):
    wt, sc = rnd((24, 12, 3, 3), 12), rnd((24,), 13, 0.5, 1.5)
    wp = ops.pack_weight_fwd(wt.cuda(), dtype, sc.cuda(), ci_pad=16)
    ref = torch.zeros(24, 3, 3, 16)
    ref[..., :12] = (wt * sc.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
    assert_close(wp.float().cpu(), ref, 1e-6 if dtype == torch.float32 else 8e-3)
    wd = ops.pack_weight_dgrad(wt.cuda(), dtype)
    assert_close(wd.float().cpu(), q(wt, dtype).permute(1, 2, 3, 0), 0)
    # multi-tensor bank: one launch for several convs, identical bytes to the single-tensor packers
    convs = [torch.nn.Conv2d(12, 24, 3, bias=False), torch.nn.Conv2d(40, 8, 1, bias=False), torch.nn.Conv2d(8, 300, 3, bias=False)]
    convs = [c.cuda() for c in convs]
    bank = ops.WeightBank(convs, [16, 40, 8], dtype, "cuda")
    bank.pack()
    for c, cp, f, d in zip(convs, [16, 40, 8], bank.fwd, bank.dgrad):
        assert torch.equal(f, ops.pack_weight_fwd(c.weight, dtype, None, cp))
        assert torch.equal(d, ops.pack_weight_dgrad(c.weight, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 24, 9, 7), (4, 64, 40, 40), (1, 6, 5, 5), (2, 512, 20, 20), (8, 32, 160, 160)])
def test_bn_train_fwd_bwd(ops, shape, dtype):
    """BatchNorm2d(train, eps 1e-3, momentum .03) + SiLU (+ shortcut), forward and backward, vs ATen."""
    n, c, h, w = shape
    y = rnd(shape, 14, -2, 2)
    yq = q(y, dtype).requires_grad_(True)
    gamma, beta = rnd((c,), 15, 0.5, 1.5).requires_grad_(True), rnd((c,), 16, -0.1, 0.1).requires_grad_(True)
    rm, rv = rnd((c,), 17, -0.1, 0.1), rnd((c,), 18, 0.5, 1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    res = rnd(shape, 19)
    z_ref = F.silu(F.batch_norm(yq, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3)) + q(res, dtype)
    gz = rnd(shape, 20)
    z_ref.backward(q(gz, dtype))

    yd = to_dev(ops, y, dtype)
    rmd, rvd = rm.cuda(), rv.cuda()
    scale, shift, mean, rstd = ops.bn_stats(yd, gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd, 0.03, 1e-3)
    assert_close(rmd.cpu(), rm_ref, 1e-4, "running_mean")
    assert_close(rvd.cpu(), rv_ref, 1e-4, "running_var")
    z = ops.new_act(n, c, h, w, dtype, "cuda")
    ops.bn_act_fwd(yd, scale, shift, ops.ACT_SILU, to_dev(ops, res, dtype), z)
    assert_close(z.float().cpu(), z_ref.detach(), TOL[dtype], "bn_act_fwd")
    if dtype == torch.float32:
        from tests.util import elem_err
        assert elem_err(z.cpu(), z_ref.detach()) <= 1.0, ("bn_act_fwd element-wise", elem_err(z.cpu(), z_ref.detach()))
    dy = ops.new_act(n, c, h, w, dtype, "cuda")
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    ops.bn_act_bwd(to_dev(ops, gz, dtype), yd, scale, shift, mean, rstd, ops.ACT_SILU, dy, dg, db)
    assert_close(dy.float().cpu(), yq.grad, 3 * TOL[dtype], "bn_act_bwd dy")
    assert_close(dg.cpu(), gamma.grad, 2 * TOL[dtype], "dgamma")
    assert_close(db.cpu(), beta.grad, 2 * TOL[dtype], "dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
def test_act_bwd_and_eval_affine(ops, dtype):
    y, gz = rnd((2, 16, 5, 7), 21, -3, 3), rnd((2, 16, 5, 7), 22)
    yq = q(y, dtype).requires_grad_(True)
    F.silu(yq).backward(q(gz, dtype))
    dy = ops.new_act(2, 16, 5, 7, dtype, "cuda")
    ops.act_bwd(to_dev(ops, gz, dtype), to_dev(ops, y, dtype), ops.ACT_SILU, dy)
    assert_close(dy.float().cpu(), yq.grad, TOL[dtype], "silu bwd")
    sc, sh = rnd((16,), 23, 0.5, 1.5), rnd((16,), 24)
    z = ops.new_act(2, 16, 5, 7, dtype, "cuda")
    ops.bn_act_fwd(to_dev(ops, y, dtype), sc.cuda(), sh.cuda(), ops.ACT_SIGMOID, None, z)
    ref = torch.sigmoid(q(y, dtype) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert_close(z.float().cpu(), ref, TOL[dtype], "affine+sigmoid")


def test_focus_s2d_bit_exact(ops):
    g = golden("modules")
    x = torch.from_numpy(g["focus_s2d/x0"])
    y = ops.new_act(1, 12, 4, 6, torch.float32, "cuda")
    ops.focus_s2d(x.cuda(), y)
    assert np.array_equal(y.cpu().numpy(), g["focus_s2d/y0"])
    y16 = ops.new_act(1, 16, 4, 6, torch.bfloat16, "cuda")
    ops.focus_s2d(x.cuda(), y16)
    assert np.array_equal(y16[:, :12].float().cpu().numpy(), torch.from_numpy(g["focus_s2d/y0"]).bfloat16().float().numpy())
    assert float(y16[:, 12:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [5, 9, 13])
def test_maxpool(ops, k, dtype):
    x = rnd((2, 16, 15, 11), 25)
    xq = q(x, dtype).requires_grad_(True)
    ref = F.max_pool2d(xq, k, 1, k // 2)
    gy = rnd(tuple(ref.shape), 26)
    ref.backward(q(gy, dtype))
    xd = to_dev(ops, x, dtype)
    y = ops.new_act(2, 16, 15, 11, dtype, "cuda")
    idx = torch.empty((2, 15, 11, 16), dtype=torch.int32, device="cuda")
    ops.maxpool_s1(xd, y, k, idx)
    assert torch.equal(y.float().cpu(), ref.detach()), "max pool values are exact"
    if dtype == torch.float32:   # bf16 inputs tie often; ATen's tie rule is checked on the tie-free fp32 data
        dx = ops.new_act(2, 16, 15, 11, dtype, "cuda")
        ops.maxpool_s1_bwd(to_dev(ops, gy, dtype), idx, dx, k)
        assert_close(dx.float().cpu(), xq.grad, 1e-5, "maxpool bwd")


def test_maxpool_ties_first_max(ops):
    x = torch.zeros(1, 4, 6, 6)
    xq = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xq, 5, 1, 2)
    ref.backward(torch.ones_like(ref))
    y = ops.new_act(1, 4, 6, 6, torch.float32, "cuda")
    idx = torch.empty((1, 6, 6, 4), dtype=torch.int32, device="cuda")
    ops.maxpool_s1(to_dev(ops, x, torch.float32), y, 5, idx)
    dx = ops.new_act(1, 4, 6, 6, torch.float32, "cuda")
    ops.maxpool_s1_bwd(to_dev(ops, torch.ones(1, 4, 6, 6), torch.float32), idx, dx, 5)
    assert torch.equal(dx.cpu(), xq.grad)


@pytest.mark.parametrize("dtype", DTYPES)
def test_upsample_nearest(ops, dtype):
    g = golden("modules")
    a, b = torch.from_numpy(g["upcat_train/x0"]), torch.from_numpy(g["upcat_train/x1"])
    cat = ops.new_act(2, 12, 10, 14, dtype, "cuda")
    ops.upsample_nearest2x(to_dev(ops, a, dtype), cat[:, :8])
    ops.copy(to_dev(ops, b, dtype), cat[:, 8:])
    assert torch.equal(cat.float().cpu(), q(torch.from_numpy(g["upcat_train/y0"]), dtype))
    gy = torch.from_numpy(g["upcat_train/gy0"])
    gd = to_dev(ops, gy, dtype)
    dx = ops.new_act(2, 8, 5, 7, dtype, "cuda")
    ops.upsample_nearest2x_bwd(gd[:, :8], dx)
    ref = F.avg_pool2d(q(gy[:, :8], dtype), 2) * 4
    assert_close(dx.float().cpu(), ref, TOL[dtype], "nearest bwd")
    if dtype == torch.float32:
        assert_close(dx.cpu(), g["upcat_train/dx0"], 1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [((2, 8, 5, 7), (10, 14)), ((1, 16, 1, 1), (9, 9)), ((2, 4, 6, 6), (20, 14)),
                                  ((1, 2, 16, 12), (128, 96)), ((2, 8, 3, 2), (12, 8))])
def test_bilinear_align_corners(ops, case, dtype):
    shape, out = case
    x = rnd(shape, 27)
    xq = q(x, dtype).requires_grad_(True)
    ref = F.interpolate(xq, out, mode="bilinear", align_corners=True)
    gy = rnd(tuple(ref.shape), 28)
    ref.backward(q(gy, dtype) if dtype != torch.float32 else gy)
    xd = to_dev(ops, x, dtype)
    y = ops.new_act(shape[0], shape[1], out[0], out[1], dtype, "cuda")
    ops.bilinear_ac(xd, y)
    assert_close(y.float().cpu(), ref.detach(), TOL[dtype], "bilinear fwd")
    yn = torch.empty((shape[0], shape[1], out[0], out[1]), device="cuda")
    ops.bilinear_ac(xd, yn, out_nchw=True)
    assert yn.is_contiguous()
    assert_close(yn.cpu(), ref.detach(), 1e-5, "bilinear fwd NCHW fp32 out")
    dx = ops.new_act(*shape, dtype, "cuda")
    ops.bilinear_ac_bwd(to_dev(ops, gy, dtype), dx)
    assert_close(dx.float().cpu(), xq.grad, TOL[dtype], "bilinear bwd")
    dx2 = ops.new_act(*shape, dtype, "cuda")
    ops.bilinear_ac_bwd((q(gy, dtype) if dtype != torch.float32 else gy).cuda().contiguous(), dx2, dy_nchw=True)
    assert_close(dx2.float().cpu(), xq.grad, TOL[dtype], "bilinear bwd from NCHW dy")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [1, 2, 3, 6])
def test_adaptive_avgpool(ops, k, dtype):
    x = rnd((2, 16, 20, 14), 29)
    xq = q(x, dtype).requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xq, k)
    gy = rnd(tuple(ref.shape), 30)
    ref.backward(q(gy, dtype))
    y = ops.new_act(2, 16, k, k, dtype, "cuda")
    ops.adaptive_avgpool(to_dev(ops, x, dtype), y)
    assert_close(y.float().cpu(), ref.detach(), TOL[dtype], "adaptive pool")
    dx = ops.new_act(2, 16, 20, 14, dtype, "cuda")
    ops.adaptive_avgpool_bwd(to_dev(ops, gy, dtype), dx)
    assert_close(dx.float().cpu(), xq.grad, TOL[dtype], "adaptive pool bwd")


@pytest.mark.parametrize("c", [16, 10, 24])          # 16 / 24: 16-byte channel vectors; 10: the scalar form (no vector width divides it)
@pytest.mark.parametrize("dtype", DTYPES)
def test_ffm_scale(ops, dtype, c):
    f, a = rnd((3, c, 8, 6), 31), rnd((3, c, 1, 1), 32, 0, 1)
    fq, aq = q(f, dtype).requires_grad_(True), q(a, dtype).requires_grad_(True)
    ref = fq * aq + fq
    g = rnd((3, c, 8, 6), 33)
    ref.backward(q(g, dtype))
    fd, ad = to_dev(ops, f, dtype), to_dev(ops, a, dtype)
    out = ops.new_act(3, c, 8, 6, dtype, "cuda")
    ops.ffm_scale(fd, ad, out)
    assert_close(out.float().cpu(), ref.detach(), TOL[dtype], "ffm_scale")
    df, da = ops.new_act(3, c, 8, 6, dtype, "cuda"), ops.new_act(3, c, 1, 1, dtype, "cuda")
    ops.ffm_scale_bwd(to_dev(ops, g, dtype), fd, ad, df, da)
    assert_close(df.float().cpu(), fq.grad, TOL[dtype], "ffm dfeat")
    assert_close(da.float().cpu(), aq.grad, TOL[dtype], "ffm datt")
    # accumulate: dfeat += (the attention branch's gradient arrives in a buffer that already holds the other consumer's)
    base = rnd((3, c, 8, 6), 34)
    df2 = to_dev(ops, base, dtype)
    ops.ffm_scale_bwd(to_dev(ops, g, dtype), fd, ad, df2, da, accumulate=True)
    assert_close(df2.float().cpu(), fq.grad + q(base, dtype), 2 * TOL[dtype], "ffm dfeat (accumulate)")


@pytest.mark.parametrize("dtype", DTYPES)
def test_detect_decode(ops, dtype):
    """Detect eval branch (yolo.py:262-275) incl. grid[...,0]=x / [...,1]=y indexing and pixel anchors, vs the oracle."""
    from oracle import desenet_ref as R
    na, no, ny, nx, n = 3, 11, 6, 9, 2
    t = rnd((n, na * no, ny, nx), 34, -3, 3)
    anchors = torch.tensor([[10., 13.], [16., 30.], [33., 23.]])
    tq = q(t, dtype)
    y = tq.view(n, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    s = y.sigmoid()
    xy = (s[..., 0:2] * 2.0 - 0.5 + R.make_grid(nx, ny)) * 8.0
    wh = (s[..., 2:4] * 2) ** 2 * anchors.view(1, na, 1, 1, 2)
    ref = torch.cat((xy, wh, s[..., 4:]), -1).view(n, -1, no)
    raw = torch.empty((n, na, ny, nx, no), device="cuda")
    pred = torch.zeros((n, na * ny * nx + 5, no), device="cuda")
    ops.detect_decode(to_dev(ops, t, dtype), raw, pred, 5, na, no, 8.0, anchors.cuda())
    assert torch.equal(raw.cpu(), y), "raw is a pure permutation"
    assert_close(pred[:, 5:].cpu(), ref, 1e-5, "decode")
    assert float(pred[:, :5].abs().max()) == 0
    dt = ops.new_act(n, na * no, ny, nx, dtype, "cuda")
    ops.detect_raw_bwd(raw, dt, na, no)
    assert torch.equal(dt.float().cpu(), tq)


@pytest.mark.parametrize("dtype", DTYPES)
def test_detect_multi_level_launches(ops, dtype):
    """All Detect levels per launch (dsn_detect_decode_multi / dsn_detect_raw_bwd_multi, 32-bit index arithmetic) on ragged maps:
    raw is the exact permutation, pred the decode of yolo.py:262-275 at each level's row offset, the backward restores the head
    outputs into row-padded tensors (padding zero-filled) and adds the per-channel sums into the bias gradients."""
    from oracle import desenet_ref as R
    na, no, n = 3, 7, 3
    shapes, strides = [(13, 11), (7, 5), (4, 3)], [8.0, 16.0, 32.0]
    anchors = torch.tensor([[[10., 13.], [16., 30.], [33., 23.]], [[30., 61.], [62., 45.], [59., 119.]],
                            [[116., 90.], [156., 198.], [373., 326.]]])
    ts = [rnd((n, na * no, ny, nx), 60 + l, -3, 3) for l, (ny, nx) in enumerate(shapes)]
    rows, row = [], 2
    for ny, nx in shapes:
        rows.append(row)
        row += na * ny * nx
    raws = [torch.empty((n, na, ny, nx, no), device="cuda") for ny, nx in shapes]
    pred = torch.zeros((n, row + 3, no), device="cuda")
    ops.detect_decode_multi([to_dev(ops, t, dtype) for t in ts], raws, pred, rows, na, no, strides, anchors.cuda().contiguous())
    for l, (ny, nx) in enumerate(shapes):
        y = q(ts[l], dtype).view(n, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        assert torch.equal(raws[l].cpu(), y), f"level {l}: raw is a pure permutation"
        sg = y.sigmoid()
        xy = (sg[..., 0:2] * 2.0 - 0.5 + R.make_grid(nx, ny)) * strides[l]
        wh = (sg[..., 2:4] * 2) ** 2 * anchors[l].view(1, na, 1, 1, 2)
        ref = torch.cat((xy, wh, sg[..., 4:]), -1).view(n, -1, no)
        assert_close(pred[:, rows[l]:rows[l] + na * ny * nx].cpu(), ref, 1e-5, f"decode level {l}")
    assert float(pred[:, :2].abs().max()) == 0 and float(pred[:, row:].abs().max()) == 0
    vec = 4 if dtype == torch.float32 else 8
    draws = [torch.randn(n, na, ny, nx, no, device="cuda") for ny, nx in shapes]
    dts = [ops.new_act(n, na * no, ny, nx, dtype, "cuda", ldc_align=vec) for ny, nx in shapes]
    for d in dts:
        ops.padded_view(d).fill_(7.0)                                  # stale contents, padding lanes included
    bias = [torch.full((na * no,), 0.5, device="cuda") for _ in shapes]
    ops.detect_raw_bwd_multi(draws, dts, na, no, bias)
    for l, (ny, nx) in enumerate(shapes):
        want = draws[l].permute(0, 1, 4, 2, 3).reshape(n, na * no, ny, nx)
        assert torch.equal(dts[l].float(), want.to(dtype).float()), f"level {l}: dt"
        pv = ops.padded_view(dts[l])
        assert float(pv[:, na * no:].float().abs().max()) == 0, f"level {l}: row padding must be zero"
        assert_close(bias[l].cpu(), 0.5 + want.sum(dim=(0, 2, 3)).cpu(), 1e-5, f"level {l}: bias gradient")


NMS_CASES = ["default", "val_multilabel", "agnostic", "classes", "ties", "maxdet", "empty", "over30000", "iou_edge"]


@pytest.mark.parametrize("case", NMS_CASES)
def test_nms_bit_exact_vs_golden(ops, case):
    """Selection AND values bit-exact against the reference's non_max_suppression (greedy step = published algorithm)."""
    import ast
    g = golden("nms")
    kw = ast.literal_eval(str(g[f"{case}/kw"]))
    pred = torch.from_numpy(g[f"{case}/pred"]).cuda()
    out, cnt = ops.nms(pred, kw["conf_thres"], kw["iou_thres"], kw.get("multi_label", False), kw.get("agnostic", False),
                       kw.get("classes"), kw["max_det"])
    cnt = cnt.cpu().tolist()
    assert cnt == [int(v) for v in g[f"{case}/n"]]
    for i, c in enumerate(cnt):
        np.testing.assert_array_equal(out[i, :c].cpu().numpy(), g[f"{case}/out{i}"])


def test_nms_apriori_labels(ops):
    """non_max_suppression(labels=...) (general.py:690-697, auto-labelling): per image, rows [box, conf 1, one-hot class] are
    appended behind the image's own candidates.  Exact selection against the from-spec restatement run on the concatenated rows;
    images with different label counts (and none) in one batch."""
    import numpy as np
    from desenet_amd.core.utils.general import non_max_suppression
    from oracle import nms_ref
    rng = np.random.RandomState(11)
    bs, n, nc = 3, 500, 6
    p = np.zeros((bs, n, 5 + nc), np.float32)
    p[..., 0:2] = rng.uniform(0, 320, (bs, n, 2))
    p[..., 2:4] = rng.uniform(8, 120, (bs, n, 2))
    p[..., 4] = rng.uniform(0, 1, (bs, n))
    p[..., 5:] = rng.uniform(0, 1, (bs, n, nc))
    labels = [np.array([[2, 100, 120, 40, 50], [5, 30, 40, 20, 20], [2, 102, 121, 41, 49]], np.float32),      # class, xywh
              np.zeros((0, 5), np.float32),
              np.array([[0, 200, 210, 80, 60]], np.float32)]
    for multi in (False, True):
        got = non_max_suppression(torch.from_numpy(p).cuda(), 0.3, 0.5, multi_label=multi, labels=[torch.from_numpy(l) for l in labels],
                                  max_det=300)
        for xi in range(bs):
            l = labels[xi]
            v = np.zeros((len(l), 5 + nc), np.float32)
            if len(l):
                v[:, :4] = l[:, 1:5]
                v[:, 4] = 1.0
                v[np.arange(len(l)), l[:, 0].astype(int) + 5] = 1.0
            rows = np.concatenate([p[xi], v], 0)[None]
            want = nms_ref.non_max_suppression(rows, 0.3, 0.5, multi_label=multi, max_det=300)[0]
            assert got[xi].shape == want.shape and np.array_equal(got[xi].cpu().numpy(), want), (multi, xi, got[xi].shape, want.shape)
            if len(l):       # a label row (conf 1.0) outranks every candidate of its class
                assert float(got[xi][:, 4].max()) == 1.0


def test_nms_vs_oracle_random_large(ops):
    """25200 x 6 candidates at the val setting (conf .001, multi-label): exercises the >8192-key sort and the 30000 cap."""
    from oracle import nms_ref
    rng = np.random.RandomState(0)
    p = np.zeros((2, 25200, 11), np.float32)
    p[..., 0:2] = rng.uniform(0, 640, (2, 25200, 2))
    p[..., 2:4] = rng.uniform(4, 200, (2, 25200, 2))
    p[..., 4:] = rng.uniform(0, 1, (2, 25200, 7))
    ref = nms_ref.non_max_suppression(p, 0.001, 0.6, multi_label=True, max_det=300)
    out, cnt = ops.nms(torch.from_numpy(p).cuda(), 0.001, 0.6, multi_label=True, max_det=300)
    for i, r in enumerate(ref):
        assert int(cnt[i]) == r.shape[0]
        np.testing.assert_array_equal(out[i, :r.shape[0]].cpu().numpy(), r)
    ref = nms_ref.non_max_suppression(p, 0.25, 0.45, max_det=1000)
    out, cnt = ops.nms(torch.from_numpy(p).cuda(), 0.25, 0.45, max_det=1000)
    for i, r in enumerate(ref):
        assert int(cnt[i]) == r.shape[0]
        np.testing.assert_array_equal(out[i, :r.shape[0]].cpu().numpy(), r)


@pytest.mark.parametrize("n,conf", [(7000, 0.001), (9000, 0.001), (14000, 0.001), (20000, 0.4), (20000, 0.001), (32768, 0.001),
                                    (33000, 0.3), (40000, 0.001), (70000, 0.55)])
def test_nms_sort_ranges_vs_oracle(ops, n, conf):
    """Every path of the hand-written candidate sort (detect_nms.hip stage 2) against the from-spec restatement: one LDS tile (<= 8192
    candidates), two tiles + the 16384-merge, four tiles + both merges, with fewer candidates than tiles (the result then lies in the
    second key buffer) and -- above 32768 candidates -- the radix select of the 30000 best.  Overlapping boxes, so the selection
    depends on the order deep into the list; two images with different candidate counts."""
    from oracle import nms_ref
    rng = np.random.RandomState(n)
    p = np.zeros((2, n, 6), np.float32)
    p[..., 0:2] = rng.uniform(0, 640, (2, n, 2))
    p[..., 2:4] = rng.uniform(4, 120, (2, n, 2))
    p[..., 4] = rng.uniform(0, 1, (2, n))
    p[1, : n // 3, 4] *= 0.5            # the second image keeps fewer candidates
    p[..., 5] = rng.uniform(0.5, 1, (2, n))
    ref = nms_ref.non_max_suppression(p, conf, 0.5, max_det=1000)
    out, cnt = ops.nms(torch.from_numpy(p).cuda(), conf, 0.5, max_det=1000)
    for i, r in enumerate(ref):
        assert int(cnt[i]) == r.shape[0]
        np.testing.assert_array_equal(out[i, :r.shape[0]].cpu().numpy(), r)


def test_fused_sgd_matches_torch(ops):
    """dsn_sgd_step (one launch for every parameter, three groups) against torch.optim.SGD(momentum, nesterov, weight decay)
    over several steps, including a learning-rate change picked up from the device-side hyper table."""
    from desenet_amd.optim import FusedSGD
    torch.manual_seed(0)
    shapes = [(32, 12, 3, 3), (64,), (33, 128, 1, 1), (5,), (256, 256, 3, 3), (1,)]
    ref = [torch.nn.Parameter(rnd(s, 50 + i).cuda()) for i, s in enumerate(shapes)]
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    groups = lambda ps: [dict(params=ps[1:2] + ps[3:4], weight_decay=0.0), dict(params=[ps[0], ps[2], ps[4]], weight_decay=5e-4),
                         dict(params=ps[5:], weight_decay=0.0, lr=0.05)]
    o_ref = torch.optim.SGD(groups(ref), lr=0.01, momentum=0.937, nesterov=True)
    o_got = FusedSGD(groups(got), lr=0.01, momentum=0.937, nesterov=True)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref, got)):
            g = rnd(a.shape, 500 + 10 * step + i).cuda()
            a.grad, b.grad = g.clone(), g.clone()
        if step == 2:
            for o in (o_ref, o_got):
                o.param_groups[1]["lr"] = 0.003
        o_ref.step()
        o_got.step()
    for a, b, s_ in zip(ref, got, shapes):
        assert_close(b.detach().cpu(), a.detach().cpu(), 1e-6, f"FusedSGD param {s_}")
        assert_close(o_got.state[b]["momentum_buffer"].cpu(), o_ref.state[a]["momentum_buffer"].cpu(), 1e-6, f"momentum {s_}")


def test_errors_are_reported_not_fatal(ops):
    x = ops.new_act(1, 8, 4, 4, torch.float32, "cuda")
    y = ops.new_act(1, 8, 5, 5, torch.float32, "cuda")
    w = torch.zeros(8, 3, 3, 8, device="cuda")
    with pytest.raises(RuntimeError, match="expected 4x4"):
        ops.conv2d_fwd(x, w, None, None, y, ops.conv_params(3))
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.desc(torch.zeros(1, 8, 4, 4))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pyramid_multi_launches_equal_the_single_ones(ops, dtype):
    """dsn_adaptive_avgpool_multi / dsn_bilinear_ac_multi / dsn_bilinear_ac_bwd_multi (PyramidPooling's four branches per
    launch) against the one-branch entry points they batch.  The reductions are the same code (bit-identical); the bilinear
    interpolation is the same expression compiled in another kernel (FMA contraction may differ: one ulp)."""
    g = torch.Generator().manual_seed(12)
    n, c, h, w, oc = 2, 32, 20, 24, 8
    x = ops.new_act(n, c, h, w, dtype, "cuda")
    x.copy_(torch.randn(n, c, h, w, generator=g))
    ks = [1, 2, 3, 6]
    single = [ops.adaptive_avgpool(x, ops.new_act(n, c, k, k, dtype, "cuda")) for k in ks]
    multi = ops.adaptive_avgpool_multi(x, [ops.new_act(n, c, k, k, dtype, "cuda") for k in ks])
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    fs = []
    for k in ks:
        f = ops.new_act(n, oc, k, k, dtype, "cuda")
        f.copy_(torch.randn(n, oc, k, k, generator=g))
        fs.append(f)
    out_s, out_m = ops.new_act(n, 4 * oc, h, w, dtype, "cuda"), ops.new_act(n, 4 * oc, h, w, dtype, "cuda")
    for j, f in enumerate(fs):
        ops.bilinear_ac(f, out_s[:, j * oc:(j + 1) * oc])
    ops.bilinear_ac_multi(fs, [out_m[:, j * oc:(j + 1) * oc] for j in range(4)])
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=8e-3, atol=1e-3)
    assert torch.allclose(out_s.float(), out_m.float(), **tol)
    dy = ops.new_act(n, 4 * oc, h, w, dtype, "cuda")
    dy.copy_(torch.randn(n, 4 * oc, h, w, generator=g))
    ds = [ops.bilinear_ac_bwd(dy[:, j * oc:(j + 1) * oc], ops.new_act(n, oc, k, k, dtype, "cuda")) for j, k in enumerate(ks)]
    dm = ops.bilinear_ac_bwd_multi([dy[:, j * oc:(j + 1) * oc] for j in range(4)],
                                   [ops.new_act(n, oc, k, k, dtype, "cuda") for k in ks])
    for a, b in zip(ds, dm):
        assert torch.equal(a, b)


def test_wgrad_halo_tile_kernel_in_its_own_process():
    """The halo-tile all-taps weight-gradient kernel takes 3x3 layers from 32768 output pixels up by default (DSN_WGRAD_HALO=2, read
    once per process): force it onto EVERY eligible layer in a child process (=1) and compare with the result of the all-taps /
    per-tap kernels (=0) for same-size 3x3 layers at dilation 1, 2, 3 and for stride-2 layers, with ragged
    4 x 8 patches, through the single-layer entry point and through the grouped launches."""
    import os, subprocess, sys
    code = r'''
import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import hip_ops as ops
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
out = {}
queue = ops.WgradQueue(torch.device("cuda", torch.cuda.current_device())) if sys.argv[1] == "queue" else None
for i, (n, ci, h, w, co, st, d) in enumerate([(8, 64, 70, 67, 96, 1, 1), (2, 16, 130, 131, 32, 1, 1), (2, 64, 130, 131, 64, 1, 2),
                                             (2, 32, 140, 128, 32, 1, 3), (4, 32, 190, 187, 64, 2, 1), (8, 64, 160, 160, 128, 2, 1)]):
    g = torch.Generator(device="cuda").manual_seed(i)
    ho, wo = ops.conv_out_hw(h, w, 3, st, d, d)
    x = ops.as_act((torch.randn((n, ci, h, w), device="cuda", generator=g)).to(dt))
    dy = ops.as_act((torch.randn((n, co, ho, wo), device="cuda", generator=g)).to(dt))
    dw = torch.zeros(co, ci, 3, 3, device="cuda")
    ops.conv2d_wgrad(x, dy, dw, ci, ops.conv_params(3, st, d, d, accumulate=True), oihw=True, queue=queue)
    out[i] = dw
if queue is not None:
    queue.flush()
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[2])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, env in (("ref", {"DSN_WGRAD_HALO": "0"}), ("halo", {"DSN_WGRAD_HALO": "1"}), ("haloq", {"DSN_WGRAD_HALO": "1"})):
            path = os.path.join(td, tag + ".pt")
            subprocess.run([sys.executable, "-c", code, "queue" if tag == "haloq" else "single", path], cwd=root, check=True,
                           env=dict(os.environ, **env), timeout=300)
            res[tag] = torch.load(path)
        for k, ref in res["ref"].items():
            scale = float(ref.abs().max())
            assert scale > 0
            for tag in ("halo", "haloq"):
                err = float((res[tag][k] - ref).abs().max())
                assert err <= 2e-4 * scale, (tag, k, err / scale)      # fp32 accumulation in a different order


def test_copy_multi_stages_several_buffers_in_one_launch(ops):
    """dsn_copy_multi: flat copies with zero-filled tails -- 16-byte and 4-byte paths, an empty source, a destination view."""
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 255, (8, 3, 64, 64), dtype=torch.uint8, generator=g).cuda()
    rows = torch.rand((5, 6), generator=g).cuda()                       # 120 bytes: the 4-byte path
    mask = torch.randint(0, 2, (8, 64, 64), dtype=torch.int64, generator=g).cuda()
    d_img = torch.full_like(img, 7)
    d_rows = torch.full((32, 6), 9.0, device="cuda")
    d_mask = torch.full_like(mask, 5)
    d_clear = torch.full((33,), 4.0, device="cuda")
    ops.copy_multi([(d_img, img), (d_rows, rows), (d_mask, mask), (d_clear, None)])
    torch.cuda.synchronize()
    assert torch.equal(d_img, img) and torch.equal(d_mask, mask)
    assert torch.equal(d_rows[:5], rows) and float(d_rows[5:].abs().max()) == 0
    assert float(d_clear.abs().max()) == 0
    with pytest.raises(ValueError):
        ops.copy_multi([(d_rows, rows.double())])
    with pytest.raises(RuntimeError):
        ops.copy_multi([(d_rows[:2], rows)])                            # source larger than the destination
