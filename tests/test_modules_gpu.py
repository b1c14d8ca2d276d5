"""Module-level parity on the MI355X: the mirrored core.models.common / core.models.yolo classes (HIP kernels through the
C ABI) against golden vectors captured from the REFERENCE's own modules (tests/golden/modules.npz), forward and backward.

Tolerances (max-abs error / max-abs reference, per tensor):
  fp32 : 1e-3  (BASELINE.json: "within 1e-3 rel for fp32 conv/BN outputs"); gradients 2e-3
  bf16 : 4e-2 forward / 8e-2 gradients against the fp32 goldens (bf16 storage of every activation, fp32 accumulate)
"""
import numpy as np
import pytest
import torch

from tests.util import assert_close, case_weights, golden

pytestmark = pytest.mark.gpu

FWD_TOL = {torch.float32: 1e-3, torch.bfloat16: 4e-2}
BWD_TOL = {torch.float32: 2e-3, torch.bfloat16: 8e-2}
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


@pytest.fixture(scope="module")
def dsn():
    import desenet_amd
    from desenet_amd.core.models import common, yolo
    assert torch.cuda.is_available()
    return desenet_amd, common, yolo


def _detect(yolo):
    det = yolo.Detect(6, ANCHORS, ch=(16, 32, 64))
    det.stride = torch.tensor([8.0, 16.0, 32.0])
    det.anchors /= det.stride.view(-1, 1, 1)
    return det


CASES = {
    # name: (factory(common, yolo), n_inputs, list_input)
    "conv_k1": (lambda c, y: c.Conv(16, 24, 1, 1), 1, False),
    "conv_k3s1": (lambda c, y: c.Conv(8, 16, 3, 1), 1, False),
    "conv_k3s2": (lambda c, y: c.Conv(8, 16, 3, 2), 1, False),
    "conv_q1": (lambda c, y: c.Conv(16, 8, 1), 1, False),
    "conv_noact": (lambda c, y: c.Conv(8, 8, 1, 1, act=False), 1, False),
    "focus": (lambda c, y: c.Focus(3, 16, 3), 1, False),
    "bneck_add": (lambda c, y: c.Bottleneck(16, 16, True), 1, False),
    "bneck_noadd": (lambda c, y: c.Bottleneck(16, 16, False), 1, False),
    "c3_n1": (lambda c, y: c.C3(16, 16, 1), 1, False),
    "c3_n3": (lambda c, y: c.C3(16, 32, 3), 1, False),
    "c3_n1_noshort": (lambda c, y: c.C3(32, 16, 1, False), 1, False),
    "spp": (lambda c, y: c.SPP(16, 16, (5, 9, 13)), 1, False),
    "rfb2": (lambda c, y: c.RFB2(48, 16, map_reduce=6, d=[2, 3]), 1, False),
    "pyramid": (lambda c, y: c.PyramidPooling(16, k=[1, 2, 3, 6], short_cut=True), 1, False),
    "ffm": (lambda c, y: c.FFM(32, 16, k=3, is_cat=False), 1, False),
    "segpsp": (lambda c, y: y.SegMaskPSP(2, 1, 24, False, ch=(16, 32, 64)), 3, True),
    "detect": (lambda c, y: _detect(y), 3, True),
}


def _flat(y):
    if torch.is_tensor(y):
        return [y]
    out = []
    for e in y:
        out.extend(_flat(e))
    return out


def _spp_reference_with_the_kernels_pool_input(m, x, gy):
    """Tie-aware reference for SPP's bf16 backward: bf16 rounding creates ties inside the 5/9/13 max-pool windows that the fp32
    golden does not have, so the arg-max (hence the routing of the gradient) legitimately differs.  Here the CPU oracle (fp32
    autograd) is teacher-forced with the pool input the KERNELS saw -- cv1's bf16 output, straight-through for the gradient --
    so both sides route through the same ties with the same first-maximum rule (pinned bit-exactly in test_kernels_gpu), and
    everything else is compared at the ordinary bf16 gradient tolerance.  Returns (dx, {param name: grad})."""
    import copy
    import torch.nn.functional as F
    from oracle import desenet_ref as R
    with torch.no_grad():
        z0_hip = copy.deepcopy(m).train().cv1(x.detach()).float().cpu()
    sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    cx = R.Ctx(sd, training=True)
    xc = x.detach().float().cpu().clone().requires_grad_(True)
    z0 = R.conv_bn_act(cx, xc, "cv1", 1)
    z0 = z0 + (z0_hip - z0).detach()
    y = R.conv_bn_act(cx, torch.cat([z0] + [F.max_pool2d(z0, k, 1, k // 2) for k in (5, 9, 13)], 1), "cv2", 1)
    y.backward(gy.float().cpu())
    return xc.grad, {k: v.grad for k, v in sd.items() if v.requires_grad}


def _build(dsn, name, mode, dtype):
    desenet_amd, common, yolo = dsn
    from desenet_amd.core.utils.torch_utils import initialize_weights
    factory, n_in, is_list = CASES[name]
    g = golden("modules")
    case = f"{name}_{mode}"
    m = factory(common, yolo)
    initialize_weights(m)
    m.load_state_dict(case_weights(g, case), strict=True)
    m = m.cuda().train(mode == "train")
    desenet_amd.set_compute_dtype(dtype)
    xs = [torch.from_numpy(g[f"{case}/x{j}"]).cuda() for j in range(n_in)]
    return m, xs, is_list, g, case


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", sorted(CASES))
def test_module_eval(dsn, name, dtype):
    m, xs, is_list, g, case = _build(dsn, name, "eval", dtype)
    try:
        with torch.no_grad():
            y = m(xs) if is_list else m(xs[0])
        for j, o in enumerate(_flat(y)):
            assert_close(o.float().cpu(), g[f"{case}/y{j}"], FWD_TOL[dtype], f"{case} y{j}")
    finally:
        dsn[0].set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", sorted(CASES))
def test_module_train_fwd_bwd(dsn, name, dtype):
    m, xs, is_list, g, case = _build(dsn, name, "train", dtype)
    try:
        needs_dx = name != "focus"          # the image never needs a gradient (no un-Focus kernel on the hot path)
        xs = [x.requires_grad_(needs_dx) for x in xs]
        y = m(xs) if is_list else m(xs[0])
        outs = _flat(y)
        for j, o in enumerate(outs):
            assert_close(o.float().cpu(), g[f"{case}/y{j}"], FWD_TOL[dtype], f"{case} y{j}")
        gys = [torch.from_numpy(g[f"{case}/gy{j}"]).cuda().to(outs[j].dtype) for j in range(len(outs))]
        torch.autograd.backward(outs, gys)
        tol = BWD_TOL[dtype]
        # bf16 rounding creates ties inside the 5/9/13 max-pool windows that do not exist in the fp32 golden, so the
        # arg-max (hence the routing of dx) legitimately differs; the tie rule itself is pinned in test_kernels_gpu.
        spp_bf16 = name == "spp" and dtype == torch.bfloat16
        if needs_dx and not spp_bf16:
            for j, x in enumerate(xs):
                assert_close(x.grad.float().cpu(), g[f"{case}/dx{j}"], tol, f"{case} dx{j}")
        tie_ref = None
        if spp_bf16:
            dx_ref, tie_ref = _spp_reference_with_the_kernels_pool_input(m, xs[0], gys[0])
            assert_close(xs[0].grad.float().cpu(), dx_ref, tol, f"{case} dx0 (tie-aware)")
        for k, p in m.named_parameters():
            ref = g[f"{case}/dw/{k}"]
            if tie_ref is not None and k.startswith("cv1.") and ref.size:
                assert_close(p.grad.float().cpu(), tie_ref[k], tol, f"{case} dw {k} (tie-aware)")     # upstream of the pools
                continue
            if ref.size == 0:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{case}: {k} must stay grad-less (Q1)"
            else:
                assert_close(p.grad.float().cpu(), ref, tol, f"{case} dw {k}")
        sd = m.state_dict()
        for k in sd:
            ak = f"{case}/after/{k}"
            if ak in g.files:
                assert_close(sd[k].float().cpu(), g[ak], 2e-3 if dtype == torch.float32 else 2e-2, f"{case} {k}")
    finally:
        dsn[0].set_compute_dtype(torch.float32)


@pytest.mark.parametrize("case,k,s", [("conv_k3s2_fused", 3, 2), ("conv_q1_fused", 1, 1)])
def test_fused_conv(dsn, case, k, s):
    """fuse_conv_and_bn + forward_fuse: the fused Q1 conv applies the folded BN, the un-fused one skips it."""
    _, common, _ = dsn
    from desenet_amd.core.utils.torch_utils import fuse_conv_and_bn, initialize_weights
    g = golden("modules")
    w = case_weights(g, case)
    co, ci = w["conv.weight"].shape[:2]
    m = common.Conv(ci, co, k, s)
    initialize_weights(m)
    m.load_state_dict(w)
    m.conv = fuse_conv_and_bn(m.conv, m.bn)
    delattr(m, "bn")
    for kk, v in case_weights(g, case, "wf").items():
        assert_close(m.state_dict()[kk], v, 1e-5, kk)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g[f"{case}/x0"]).cuda())
    assert_close(y.cpu(), g[f"{case}/y0"], 1e-3, case)


def test_concat_zero_copy_and_fallback(dsn):
    _, common, _ = dsn
    from desenet_amd import hip_ops as ops
    g = golden("modules")
    a, b = torch.from_numpy(g["upcat_eval/x0"]).cuda(), torch.from_numpy(g["upcat_eval/x1"]).cuda()
    up, cat = common.Upsample(None, 2, "nearest"), common.Concat(1)
    with torch.no_grad():
        y = cat([up(a), b])                       # separate buffers -> copy fallback
    assert np.array_equal(y.cpu().numpy(), g["upcat_eval/y0"])
    buf = ops.new_act(2, 12, 10, 14, torch.float32, "cuda")
    with torch.no_grad():
        up.fwd(ops.as_act(a), None, buf[:, :8])
        ops.copy(ops.as_act(b), buf[:, 8:])
        z = cat([buf[:, :8], buf[:, 8:]])         # adjacent slices -> the covering view, no copy
    assert z.data_ptr() == buf.data_ptr()
    assert np.array_equal(z.cpu().numpy(), g["upcat_eval/y0"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 128, 80, 80), (3, 64, 37, 41)])
def test_pyramid_pooling_fused_branches_match_the_per_branch_path(shape, monkeypatch):
    """PyramidPooling's four conv + BatchNorm + SiLU branches as one launch each way (csrc/pp_fused.hip, bf16 training) against the
    per-branch kernels on the same module: output, input gradient, every parameter gradient, running statistics.
    This is a SELF-comparison (it localises a fault to the fused launch); the oracle legs for this module are the reference golden
    `pyramid` in test_module_{eval,train_fwd_bwd} (non-divisible bins, fp32 1e-3) and, at the production shape 8 x 128 x 80 x 80, the
    teacher-forced layer-24 forward / backward tests of test_net_gpu.py (SegMaskPSP contains this module; both run the fused path)."""
    import copy
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.core.models.common import PyramidPooling
    from desenet_amd.parallel import FlatGradients
    desenet_amd.set_compute_dtype(torch.bfloat16)
    try:
        n, c, h, w = shape
        torch.manual_seed(5)
        ref = PyramidPooling(c).cuda().train()
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                with torch.no_grad():
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
        fused = copy.deepcopy(ref)
        x = torch.randn(n, c, h, w, device="cuda")
        gy = torch.randn(n, 4 * (c // 4), h, w, device="cuda")
        outs = {}
        calls = {"fwd": 0, "bwd": 0}
        real_fwd, real_bwd = ops.pp_stages_fwd, ops.pp_stages_bwd
        monkeypatch.setattr(ops, "pp_stages_fwd", lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), real_fwd(*a, **k))[1])
        monkeypatch.setattr(ops, "pp_stages_bwd", lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), real_bwd(*a, **k))[1])
        for name, mod, flag in (("ref", ref, "0"), ("fused", fused, "1")):
            monkeypatch.setenv("DSN_PP_FUSED", flag)
            flat = FlatGradients(mod.parameters())
            flat.zero()
            xin = x.clone().requires_grad_(True)
            ops.profile_enable(True)
            y = mod(xin)
            y.backward(gy)
            torch.cuda.synchronize()
            prof = ops.profile_collect()
            ops.profile_enable(False)
            outs[name] = (y.detach().float(), xin.grad.float(), {k: p.grad.clone() for k, p in mod.named_parameters()},
                          {k: b.clone() for k, b in mod.named_buffers()})
        assert calls == {"fwd": 1, "bwd": 1}, calls          # the fused launches ran once each, for the second module only
        for a, b, what in ((outs["ref"][0], outs["fused"][0], "output"), (outs["ref"][1], outs["fused"][1], "input gradient")):
            scale = float(a.abs().max())
            assert float((a - b).abs().max()) <= 2e-2 * scale, (what, float((a - b).abs().max()) / scale)
        for k, g in outs["ref"][2].items():
            scale = float(g.abs().max()) + 1e-12
            assert float((g - outs["fused"][2][k]).abs().max()) <= 2e-2 * scale, (k, float((g - outs["fused"][2][k]).abs().max()) / scale)
        for k, b in outs["ref"][3].items():
            if b.dtype.is_floating_point:
                assert torch.allclose(b, outs["fused"][3][k], rtol=1e-4, atol=1e-6), k
            else:
                assert torch.equal(b, outs["fused"][3][k]), k
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("n,hw,ch,nc", [(8, (80, 40, 20), (128, 256, 512), 6), (3, (13, 7, 4), (24, 40, 72), 1),
                                        (2, (20, 10, 5), (64, 128, 256), 16)])
def test_detect_fused_head_forward_matches_the_per_level_path(n, hw, ch, nc, monkeypatch):
    """The training forward of all Detect heads as ONE launch (dsn_detect_head_fwd_multi: 1x1 conv + bias + permute, yolo.py:258-276)
    against (a) a plain fp32 torch reference on the bf16-rounded operands and (b) the per-level conv + permute launches, and the
    backward that follows it (unchanged kernels reading the fused path's tape records): input and parameter gradients."""
    import copy
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.core.models import yolo
    desenet_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(11)
        ref = yolo.Detect(nc, ANCHORS, ch=ch)
        ref.stride = torch.tensor([8.0, 16.0, 32.0])
        for m in ref.m:
            with torch.no_grad():
                m.weight.normal_(0, 0.05); m.bias.uniform_(-1, 1)
        ref = ref.cuda().train()
        fused = copy.deepcopy(ref)
        xs = [torch.randn(n, c, s, s, device="cuda").bfloat16().float() for c, s in zip(ch, hw)]
        gys = [torch.randn(n, 3, s, s, nc + 5, device="cuda") for s in hw]
        calls = {"n": 0}
        real = ops.detect_head_fwd
        monkeypatch.setattr(ops, "detect_head_fwd", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real(*a, **k))[1])
        outs = {}
        for name, mod, flag in (("ref", ref, False), ("fused", fused, True)):
            monkeypatch.setattr(yolo, "_DET_FUSED", flag)
            xin = [x.clone().requires_grad_(True) for x in xs]
            ys = mod(xin)
            torch.autograd.backward(list(ys), gys)
            torch.cuda.synchronize()
            outs[name] = ([y.detach().clone() for y in ys], [x.grad.float().clone() for x in xin],
                          {k: p.grad.float().clone() for k, p in mod.named_parameters()})
        assert calls["n"] == 1, calls
        for l, (a, b) in enumerate(zip(outs["ref"][0], outs["fused"][0])):
            assert a.shape == b.shape and b.dtype == torch.float32
            m = ref.m[l]
            want = torch.nn.functional.conv2d(xs[l], m.weight.detach().bfloat16().float(), m.bias.detach().float())
            want = want.view(n, 3, nc + 5, hw[l], hw[l]).permute(0, 1, 3, 4, 2)
            scale = float(want.abs().max())
            assert float((b - want).abs().max()) <= 8e-3 * scale, ("torch", l, float((b - want).abs().max()) / scale)    # one bf16 rounding
            assert float((a - b).abs().max()) <= 8e-3 * scale, ("per-level", l, float((a - b).abs().max()) / scale)
            assert torch.equal(b, b.bfloat16().float())          # bf16-representable, as the head convolution's output is
        for l, (a, b) in enumerate(zip(outs["ref"][1], outs["fused"][1])):
            assert torch.equal(a, b), ("input gradient", l)      # same kernels, same operands
        for k, g in outs["ref"][2].items():
            assert torch.equal(g, outs["fused"][2][k]), k
    finally:
        desenet_amd.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_ffm_attention_fused_small_convs_match_the_layer_path(monkeypatch):
    """FFM's channel attention (two bias-free 1x1 convs on the pooled [n, c, 1, 1] vector, SiLU / Sigmoid: common.py:222-242) through
    the one-block kernels of csrc/pp_fused.hip against the conv + activation launches: output, input and weight gradients.
    A SELF-comparison of an OPTIONAL path (DSN_PP_FUSED=3, measured slower and off by default); the default layer path is held to the
    reference golden `ffm` and to the oracle inside the teacher-forced layer-24 tests at 640 x 640."""
    import copy
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.core.models.common import FFM
    from desenet_amd.parallel import FlatGradients
    desenet_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(7)
        ref = FFM(64, 128, k=3, is_cat=False).cuda().train()
        fused = copy.deepcopy(ref)
        x = torch.randn(4, 64, 40, 40, device="cuda")
        gy = torch.randn(4, 128, 40, 40, device="cuda")
        calls = {"fwd": 0, "bwd": 0}
        real_fwd, real_bwd = ops.pp_stages_fwd, ops.pp_stages_bwd
        monkeypatch.setattr(ops, "pp_stages_fwd", lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), real_fwd(*a, **k))[1])
        monkeypatch.setattr(ops, "pp_stages_bwd", lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), real_bwd(*a, **k))[1])
        outs = {}
        for name, mod, flag in (("ref", ref, "0"), ("fused", fused, "3")):      # (3: off by default -- measured slower in the step)
            monkeypatch.setenv("DSN_PP_FUSED", flag)
            flat = FlatGradients(mod.parameters())
            flat.zero()
            xin = x.clone().requires_grad_(True)
            y = mod(xin)
            y.backward(gy)
            torch.cuda.synchronize()
            outs[name] = (y.detach().float(), xin.grad.float(), {k: p.grad.clone() for k, p in mod.named_parameters()})
        assert calls == {"fwd": 2, "bwd": 2}, calls
        for a, b, what in ((outs["ref"][0], outs["fused"][0], "output"), (outs["ref"][1], outs["fused"][1], "input gradient")):
            scale = float(a.abs().max())
            assert float((a - b).abs().max()) <= 2e-2 * scale, (what, float((a - b).abs().max()) / scale)
        for k, g in outs["ref"][2].items():
            scale = float(g.abs().max()) + 1e-12
            assert float((g - outs["fused"][2][k]).abs().max()) <= 3e-2 * scale, (k, float((g - outs["fused"][2][k]).abs().max()) / scale)
    finally:
        desenet_amd.set_compute_dtype(torch.float32)
