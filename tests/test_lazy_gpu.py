"""Deferred BatchNorm + activation (include/desenet_hip.h: dsn_lazy_in) on the MI355X: a convolution / weight gradient that
applies z = act(y*scale + shift) while staging its operand must produce EXACTLY what it produces from the materialised z (same
fp32 arithmetic, same rounding to the storage type, zero padding after the activation) -- bit for bit for the weight gradients
(same kernel on both sides) and up to the K summation order for the convolutions (the materialised side may take the halo-tile or
one-trip kernels of conv3x3.hip, the deferred side the implicit-GEMM kernel), every staging variant:
uniform-tap chunks (Cs a multiple of the chunk), per-thread tap decode (Cs = 16 / 32), 1x1 / 3x3 / stride 2 / dilation, and a
consumer input that is a concat of two deferred tensors and an ordinary one.  The module- and net-level goldens then hold the whole
deferred forward/backward to the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


def _producer(ops, tape, n, ci, co, h, w, dtype, seed, act, buf=None, c0=0):
    """A BN'd 1x1 conv whose raw output stays in buf[:, c0:c0+co], tagged on the tape.  Returns (view, bn)."""
    from desenet_amd.conv_impl import conv_block_fwd
    conv = torch.nn.Conv2d(ci, co, 1, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(co, eps=1e-3, momentum=0.03).cuda()
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, torch.float32, seed, 0.3))
        bn.weight.copy_(torch.rand(co, device="cuda") + 0.5)
        bn.bias.copy_(torch.rand(co, device="cuda") - 0.5)
    x = ops.as_act(_rand((n, ci, h, w), dtype, seed + 1))
    out = buf[:, c0:c0 + co] if buf is not None else None
    y = conv_block_fwd(x, conv, bn, act, True, tape, out, lazy_out="force")
    return y, bn


CASES = [  # k, stride, dil, channel layout of the consumer input [(kind, channels)], kind: L = deferred, P = plain
    (1, 1, 1, [("L", 64)]),
    (3, 1, 1, [("L", 64)]),
    (3, 2, 1, [("L", 32)]),
    (3, 1, 2, [("L", 64)]),
    (1, 1, 1, [("L", 32), ("P", 64), ("L", 32)]),
    (3, 1, 1, [("P", 16), ("L", 16)]),
    (1, 1, 1, [("L", 128), ("L", 128), ("L", 128), ("L", 128)]),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,dil,layout", CASES)
def test_lazy_conv_and_wgrad_equal_the_materialised_path(dtype, k, stride, dil, layout):
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.hip_ops import ACT_NONE, ACT_SILU
    from desenet_amd.runtime import Tape
    desenet_amd.set_compute_dtype(dtype)
    try:
        n, h, w, co = 2, 13, 11, 48
        ctot = sum(c for _, c in layout)
        tape = Tape()
        buf = ops.new_act(n, ctot, h, w, dtype, "cuda")
        c0 = 0
        for j, (kind, c) in enumerate(layout):
            if kind == "L":
                _producer(ops, tape, n, 24, c, h, w, dtype, 10 * j + 3, ACT_SILU if j % 2 == 0 else ACT_NONE, buf, c0)
            else:
                ops.copy(ops.as_act(_rand((n, c, h, w), dtype, 10 * j + 5)), buf[:, c0:c0 + c])
            c0 += c
        tape.finalize_forward()
        lz = tape.lazy_in(buf)
        assert lz is not None and lz is not False and lz.nseg == sum(kind == "L" for kind, _ in layout)
        z = tape.materialize(buf)                       # one elementwise launch over the whole concat
        assert z.data_ptr() != buf.data_ptr()
        with pytest.raises(RuntimeError, match="deferred-BatchNorm"):
            ops.copy(buf, ops.new_act(n, ctot, h, w, dtype, "cuda"))       # a kernel that cannot apply the transform refuses
        wt = _rand((co, ctot, k, k), torch.float32, 99, 0.2)
        wp = ops.pack_weight_fwd(wt, dtype)
        pad = dil * (k // 2)
        ho, wo = ops.conv_out_hw(h, w, k, stride, pad, dil)
        p = ops.conv_params(k, stride, pad, dil, ACT_NONE)
        y_ref = ops.conv2d_fwd(z, wp, None, None, ops.new_act(n, co, ho, wo, dtype, "cuda"), p)
        y_lazy = ops.conv2d_fwd(buf, wp, None, None, ops.new_act(n, co, ho, wo, dtype, "cuda"), p, lazy=lz)
        # (same values staged; the materialised operand may run on another kernel of the library -- conv3x3.hip's halo-tile / one-trip
        #  kernels do not take deferred inputs -- whose K order differs: equal up to fp32 summation order / one bf16 rounding)
        tol = 2e-5 if dtype == torch.float32 else 1e-2
        assert _rel(y_lazy, y_ref) < tol, _rel(y_lazy, y_ref)
        # with BatchNorm sums in the epilogue (training form)
        y2 = ops.new_act(n, co, ho, wo, dtype, "cuda")
        acc, nb = ops.conv2d_fwd_acc(buf, wp, y2, p, lazy=lz)
        assert _rel(y2, y_ref) < tol
        # weight gradient: queued kernels with the finalised scale / shift arrays vs the materialised operand
        dy = ops.as_act(_rand((n, co, ho, wo), dtype, 7))
        lzb = tape.lazy_in(buf, backward=True)
        g_ref = torch.zeros((co, ctot, k, k), device="cuda")
        g_lazy = torch.zeros_like(g_ref)
        for x_op, lazy, g in ((tape.materialize(buf, backward=True), None, g_ref), (buf, lzb, g_lazy)):
            q = ops.WgradQueue("cuda")
            ops.conv2d_wgrad(x_op, dy, g, ctot, ops.conv_params(k, stride, pad, dil, accumulate=True), oihw=True, queue=q, lazy=lazy)
            q.flush()
        torch.cuda.synchronize()
        assert torch.equal(g_ref, g_lazy), float((g_ref - g_lazy).abs().max())
        assert float(g_ref.abs().max()) > 0
        zb = tape.materialize(buf, backward=True)
        assert torch.equal(zb, z), "scale / shift arrays of the finalisation must reproduce the forward fold exactly"
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


ZCASES = [  # k, dil, h, w, co, channel layout of the consumer input (bf16: 64-channel slabs, fp32: 32-channel slabs)
    (1, 1, 13, 11, 48, [("L", 64)]),
    (1, 1, 16, 16, 128, [("L", 64), ("L", 64)]),
    (1, 1, 20, 20, 72, [("L", 64), ("P", 64), ("L", 128)]),
    (3, 1, 13, 11, 48, [("L", 64)]),
    (3, 1, 16, 24, 136, [("L", 64), ("L", 64)]),
    (3, 2, 17, 9, 64, [("P", 64), ("L", 64)]),
    (3, 3, 12, 12, 64, [("L", 64)]),
    (3, 1, 9, 9, 40, [("L", 32)]),                  # no whole slab in bf16: elementwise launch + plain convolution inside the entry
    (5, 1, 9, 9, 40, [("L", 64)]),                  # not a kernel that stages whole tiles: the same fallback
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,dil,h,w,co,layout", ZCASES)
def test_conv_that_materialises_its_deferred_input_on_the_way(dtype, k, dil, h, w, co, layout):
    """dsn_conv2d_fwd_lazy_z: the halo-tile / one-trip kernels transform the raw tile once per block in LDS and store z as a side
    effect.  z must be bit-identical to the elementwise materialisation, and y to the same kernel run on that z."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.hip_ops import ACT_NONE, ACT_SILU
    from desenet_amd.runtime import Tape
    desenet_amd.set_compute_dtype(dtype)
    try:
        n = 3
        ctot = sum(c for _, c in layout)
        tape = Tape()
        buf = ops.new_act(n, ctot + 8, h, w, dtype, "cuda")[:, :ctot]        # (a channel slice: row stride != channels)
        c0 = 0
        for j, (kind, c) in enumerate(layout):
            if kind == "L":
                _producer(ops, tape, n, 24, c, h, w, dtype, 10 * j + 3, ACT_SILU if j % 2 == 0 else ACT_NONE, buf, c0)
            else:
                ops.copy(ops.as_act(_rand((n, c, h, w), dtype, 10 * j + 5)), buf[:, c0:c0 + c])
            c0 += c
        lz = tape.lazy_in(buf)
        z_ref = tape.materialize(buf)
        wp = ops.pack_weight_fwd(_rand((co, ctot, k, k), torch.float32, 99, 0.2), dtype)
        pad = dil * (k // 2)
        p = ops.conv_params(k, 1, pad, dil, ACT_NONE)
        y_ref = ops.new_act(n, co, h, w, dtype, "cuda")
        acc_ref, _ = ops.conv2d_fwd_acc(z_ref, wp, y_ref, p)
        z = ops.new_act(n, ctot + 16, h, w, dtype, "cuda")[:, 8:8 + ctot]
        z.fill_(7.0)
        y = ops.new_act(n, co, h, w, dtype, "cuda")
        acc, _ = ops.conv2d_fwd_acc(buf, wp, y, p, lazy=lz, z_out=z)
        torch.cuda.synchronize()
        assert torch.equal(z, z_ref), float((z.float() - z_ref.float()).abs().max())
        assert torch.equal(y, y_ref), _rel(y, y_ref)
        assert torch.allclose(acc, acc_ref, rtol=1e-12, atol=0)
        # with bias / activation / residual in the epilogue (the inference-style entry)
        bias = torch.rand(co, device="cuda") - 0.5
        res = ops.as_act(_rand((n, co, h, w), dtype, 17))
        p2 = ops.conv_params(k, 1, pad, dil, ACT_SILU)
        y2_ref = ops.conv2d_fwd(z_ref, wp, bias, res, ops.new_act(n, co, h, w, dtype, "cuda"), p2)
        z2 = ops.new_act(n, ctot, h, w, dtype, "cuda")
        y2 = ops.conv2d_fwd(buf, wp, bias, res, ops.new_act(n, co, h, w, dtype, "cuda"), p2, lazy=lz, z_out=z2)
        torch.cuda.synchronize()
        assert torch.equal(z2, z_ref) and torch.equal(y2, y2_ref)
        tape.finalize_forward()
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


def test_finalize_multi_matches_the_per_layer_statistics():
    """dsn_bn_finalize_multi (one launch for all layers) writes what dsn_bn_stats writes per layer: scale, shift, mean, rstd and
    the running averages (momentum 0.03, unbiased variance)."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.hip_ops import ACT_SILU
    from desenet_amd.runtime import Tape
    desenet_amd.set_compute_dtype(torch.float32)
    tape = Tape()
    ys, bns = [], []
    for j, co in enumerate([32, 64, 40]):
        y, bn = _producer(ops, tape, 2, 24, co, 9, 7, torch.float32, 5 + j, ACT_SILU)
        ys.append(y)
        bns.append(bn)
    recs = list(tape.lazy_pending)
    tape.finalize_forward()
    for y, bn, r in zip(ys, bns, recs):
        rm, rv = torch.zeros_like(bn.running_mean), torch.ones_like(bn.running_var)
        raw = ops.as_act(y.detach().clone())             # an untagged copy of the raw conv output
        sc, sh, mu, rs = ops.bn_stats(raw, bn.weight, bn.bias, rm, rv, 0.03, 1e-3)
        for a, b in ((r.stats[0], sc), (r.stats[1], sh), (r.stats[2], mu), (r.stats[3], rs), (bn.running_mean, rm), (bn.running_var, rv)):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_odd_channel_counts_fall_back_to_the_materialised_path():
    """Channel counts that rule out 16-byte vectors never get deferred: a Bottleneck(12 -> 12) still matches ATen."""
    import desenet_amd
    from desenet_amd.core.models.common import Bottleneck
    from desenet_amd.core.utils.torch_utils import initialize_weights
    desenet_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    m = Bottleneck(12, 12, True).cuda().train()
    initialize_weights(m)
    x = torch.randn(2, 12, 10, 9, device="cuda", requires_grad=True)
    y = m(x)
    y.sum().backward()
    ref = x + torch.nn.functional.silu(torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(torch.nn.functional.silu(torch.nn.functional.batch_norm(
            torch.nn.functional.conv2d(x, m.cv1.conv.weight), None, None, m.cv1.bn.weight, m.cv1.bn.bias, True, 0.03, 1e-3)),
            m.cv2.conv.weight, padding=1), None, None, m.cv2.bn.weight, m.cv2.bn.bias, True, 0.03, 1e-3))
    assert torch.allclose(y, ref, rtol=2e-3, atol=2e-3)
