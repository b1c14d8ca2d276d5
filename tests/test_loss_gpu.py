"""HIP loss kernels (rows a15 / a16) vs the CPU oracle (oracle/loss_ref.py = the reference's ComputeLoss /
SegmentationLosses restated on ATen-CPU ops, pinned by tests/golden/train.npz): loss values and the gradient w.r.t. every
network output.  fp32; tolerance 1e-4 on losses, 1e-3 on gradients (expf / atanf / log1pf device intrinsics)."""
import pytest
import torch

from desenet_amd.synth import synth_targets
from tests.util import assert_close, rel_err

pytestmark = pytest.mark.gpu
ANCHORS = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float().view(3, 3, 2) \
    / torch.tensor([8., 16., 32.]).view(3, 1, 1)


class _Det:
    na, nc, nl, anchors = 3, 6, 3, ANCHORS


class _Model(torch.nn.Module):
    def __init__(self, hyp):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.model = [_Det()]
        self.hyp = hyp


@pytest.mark.parametrize("bs,size,seed,nt", [(2, 128, 21, None), (4, 256, 5, None), (2, 128, 7, 0), (1, 64, 9, 3)])
def test_det_loss_vs_oracle(bs, size, seed, nt):
    from desenet_amd.core.utils.hyp import scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss
    from oracle import loss_ref
    g = torch.Generator().manual_seed(seed)
    p = [(torch.randn(bs, 3, size // s, size // s, 11, generator=g) * 2).requires_grad_(True) for s in (8, 16, 32)]
    det_t, _ = synth_targets(bs, size, seed, boxes_per_image=16)
    if nt is not None:
        det_t = det_t[:nt]
    hyp = scale_hyp(6, size)
    ref, items = loss_ref.det_loss(p, det_t, ANCHORS, hyp, 6)
    (ref * 0.5).sum().backward()
    cl = ComputeLoss(_Model(hyp))
    pd = [t.detach().cuda().requires_grad_(True) for t in p]
    loss, it = cl(pd, det_t.cuda())
    (loss * 0.5).sum().backward()
    assert loss.shape == (1,) and it.shape == (3,)
    assert_close(loss.cpu(), ref.detach(), 1e-4, "det loss")
    assert_close(it.cpu(), items, 1e-4, "loss items")
    for a, b in zip(pd, p):
        assert_close(a.grad.cpu(), b.grad, 1e-3, "d det_loss / d raw")


def test_det_loss_duplicate_cells_take_last_candidate():
    """Two targets in the same cell matched by the same anchor: the objectness target is the later candidate's IoU (what the
    reference's sequential scatter leaves behind on the CPU)."""
    from desenet_amd.core.utils.hyp import scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss
    from oracle import loss_ref
    g = torch.Generator().manual_seed(3)
    p = [torch.randn(1, 3, 64 // s, 64 // s, 11, generator=g).requires_grad_(True) for s in (8, 16, 32)]
    det_t = torch.tensor([[0, 1, 0.52, 0.52, 0.30, 0.35], [0, 4, 0.53, 0.51, 0.32, 0.30], [0, 2, 0.2, 0.7, 0.1, 0.12]])
    hyp = scale_hyp(6, 64)
    ref, items = loss_ref.det_loss(p, det_t, ANCHORS, hyp, 6)
    ref.sum().backward()
    pd = [t.detach().cuda().requires_grad_(True) for t in p]
    loss, it = ComputeLoss(_Model(hyp))(pd, det_t.cuda())
    loss.sum().backward()
    assert_close(loss.cpu(), ref.detach(), 1e-4)
    for a, b in zip(pd, p):
        assert_close(a.grad.cpu(), b.grad, 1e-3)


@pytest.mark.parametrize("shape,ignore_frac", [((2, 2, 64, 96), 0.0), ((1, 5, 33, 17), 0.3), ((8, 2, 640, 640), 0.05)])
def test_seg_ce_vs_aten(shape, ignore_frac):
    from desenet_amd.core.utils.loss import SegmentationLosses
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(shape, generator=g) * 3).requires_grad_(True)
    n, c, h, w = shape
    target = torch.randint(0, c, (n, h, w), generator=g)
    target[torch.rand(n, h, w, generator=g) < ignore_frac] = -1
    ref = torch.nn.functional.cross_entropy(logits, target, ignore_index=-1)
    (ref * 2.0).backward()
    ld = logits.detach().cuda().requires_grad_(True)
    loss = SegmentationLosses()(ld, target.cuda())
    (loss * 2.0).backward()
    assert_close(loss.cpu(), ref.detach(), 1e-5, "CE")
    assert_close(ld.grad.cpu(), logits.grad, 1e-4, "dCE")


def test_losses_replay_in_a_graph():
    """The loss launches captured in a hipGraph give the eager numbers on EVERY replay, with unrelated allocations in
    between (regression: hipMemsetAsync nodes inside the capture left the objectness-owner table stale on replay, i.e.
    garbage indices; the library now clears with fill kernels)."""
    from desenet_amd import hip_ops as ops
    dev = torch.device("cuda", torch.cuda.current_device())
    bs, size, nc = 2, 128, 6
    det_t, seg_t = synth_targets(bs, size, 21)
    det_t, seg_t = det_t.to(dev), seg_t.to(dev)
    g = torch.Generator().manual_seed(0)
    p = [torch.randn(bs, 3, size // s, size // s, 5 + nc, generator=g).to(dev) for s in (8, 16, 32)]
    logits = torch.randn(bs, 2, size, size, generator=g).to(dev)

    def body():
        out, dp = ops.det_loss(p, det_t, [float(v) for v in ANCHORS.reshape(-1)], [4.0, 1.0, 0.4], 0.05, 1.0, 0.5, 1.0, 1.0, 4.0,
                               1.0, 0.0, nc, 1.0)
        sout, dl = ops.seg_ce(logits, seg_t, -1, True)
        return torch.stack([out[0], sout[0], sum(d.abs().sum() for d in dp), dl.abs().sum()])

    want = body().cpu()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = body()
    keep = []
    for i in range(3):
        graph.replay()
        assert_close(res.cpu(), want, 1e-6, f"replay {i}")
        keep.append([torch.randn(n, device=dev) for n in (7, 1000, 100000, 3000000)])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (8, 80, 80), (1, 20, 36)])
def test_seg_ce_fused_with_the_x8_bilinear(dtype, tol, n, h, w):
    """dsn_seg_ce_up (x8 bilinear, align_corners=True, + cross entropy, loss and gradient from the 1/8-resolution logits) against
    the three-kernel form it replaces -- bilinear_ac -> seg_ce -> bilinear_ac_bwd -- which is itself pinned to ATen / the reference
    (test_seg_ce_vs_aten, G3 goldens), including ignored pixels."""
    from desenet_amd import hip_ops as ops
    g = torch.Generator(device="cuda").manual_seed(n * 100 + h)
    lg = ops.as_act((torch.randn((n, 2, h, w), device="cuda", generator=g) * 2.0).to(dtype))
    H, W = 8 * h, 8 * w
    tgt = (torch.rand((n, H, W), device="cuda", generator=g) > 0.7).long()
    tgt[0, :3, :5] = -1                                            # ignore_index
    full = torch.empty((n, 2, H, W), dtype=torch.float32, device="cuda")
    ops.bilinear_ac(lg, full, out_nchw=True)
    out_ref, dfull = ops.seg_ce(full, tgt, -1, want_grad=True)
    vec = 4 if dtype == torch.float32 else 8
    dl_ref = ops.bilinear_ac_bwd(dfull, ops.new_act(n, 2, h, w, dtype, "cuda", zero=True, ldc_align=vec), dy_nchw=True)
    out, dl = ops.seg_ce_up(lg, tgt, (H, W), -1)
    assert abs(float(out[0]) - float(out_ref[0])) <= 1e-5 * abs(float(out_ref[0])) and float(out[1]) == float(out_ref[1])
    assert rel_err(dl.float().cpu(), dl_ref.float().cpu()) < tol
    assert float(ops.padded_view(dl)[:, 2:].abs().max()) == 0.0, "row padding of the gradient must be zero"
    out2, dl2 = ops.seg_ce_up(lg, tgt, (H, W), -1)                 # the accumulator workspace was restored to zero
    assert rel_err(dl2.float().cpu(), dl.float().cpu()) < tol and float(out2[0]) == float(out[0])     # (fp32 atomics: order varies)


@pytest.mark.parametrize("tag", ["focal", "focal_pw", "auto", "focal_auto"])
def test_det_loss_focal_and_autobalance_vs_reference_golden(tag):
    """The options of ComputeLoss that scripts/train.py leaves off (loss.py:106-113,158-164) on the HIP kernels, against vectors the
    REFERENCE produced (tools/gen_golden_loss_opts.py): three consecutive calls -- with autobalance the per-level weights are state,
    kept on the device here -- losses 1e-4, gradients 1e-3, balance 1e-5."""
    import numpy as np
    from desenet_amd.core.utils.loss import ComputeLoss
    from tests.util import golden
    g = golden("loss_opts")
    box, obj, cls, cls_pw, obj_pw, anchor_t, gamma, auto = [float(v) for v in g[f"{tag}/hyp"]]
    hyp = dict(box=box, obj=obj, cls=cls, cls_pw=cls_pw, obj_pw=obj_pw, anchor_t=anchor_t, fl_gamma=gamma, label_smoothing=0.0)

    class Det(_Det):
        stride = torch.tensor([8., 16., 32.])

    m = _Model(hyp)
    m.model = [Det()]
    cl = ComputeLoss(m, autobalance=bool(auto))
    det_t = torch.from_numpy(g[f"{tag}/targets"]).cuda()
    for step in range(3):
        pd = [torch.from_numpy(g[f"{tag}/{step}/p{i}"]).cuda().requires_grad_(True) for i in range(3)]
        loss, it = cl(pd, det_t)
        loss.sum().backward()
        assert_close(loss.cpu(), g[f"{tag}/{step}/loss"], 1e-4, "det loss")
        assert_close(it.cpu(), g[f"{tag}/{step}/items"], 1e-4, "loss items")
        for i in range(3):
            assert_close(pd[i].grad.cpu(), g[f"{tag}/{step}/dp{i}"], 1e-3, f"d det_loss / d raw {i}")
        assert np.allclose(cl.balance, g[f"{tag}/{step}/balance"], rtol=1e-5, atol=0), (cl.balance, g[f"{tag}/{step}/balance"])
