"""Whole-network parity on the MI355X: mirrored Model (HIP) vs golden vectors from the reference (tests/golden/net.npz,
train.npz) and vs the CPU oracle on the same seeded inputs.

fp32 tolerance 1e-3 relative per output tensor (BASELINE.json); losses 1e-3; gradients 5e-3 at 640x640 and 2e-2 at the
tiny 128x128 / batch-2 case (batch statistics over 8..32 samples amplify rounding).

bf16 (bf16 storage of every activation, fp32 accumulation) is judged against the SAME fp32 goldens.  This random-weight,
80-layer, batch-statistics network is chaotic in bf16: PyTorch's own CPU bf16 autocast of the oracle deviates from fp32 by
0.18-0.53 (raw logits, max-norm relative), 0.36 (seg logits), 4% (global gradient norm) and 0.5-0.7 median per-parameter
gradient error (measured with /tmp-style script recorded in DESIGN.md).  Whole-net bf16 bounds are therefore loose --
eval logits 0.4 (one re-rounded bf16 activation early in the net moves this measure by 0.01-0.05: 0.25-0.26 was seen
across kernel revisions), train logits 0.6, losses 3e-2, gradient norm 25% -- and the TIGHT bf16 checks live at kernel and module
level (tests/test_kernels_gpu.py 2e-2, tests/test_modules_gpu.py 4e-2 / 8e-2)."""
import numpy as np
import pytest
import torch

from desenet_amd.synth import synth_images, synth_targets, synthetic_checkpoint
from tests.util import assert_close, golden, load_cfg, rel_err, stats, subsample

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    import desenet_amd
    from desenet_amd.core.models.yolo import Model
    m = Model("desenet_s.yaml", ch=3, nc=6)
    sd = m.state_dict()
    synthetic_checkpoint(sd)
    m.load_state_dict(sd)
    return desenet_amd, m.cuda()


def _outs(det, seg):
    if isinstance(det, tuple):
        return {"pred": det[0], "raw0": det[1][0], "raw1": det[1][1], "raw2": det[1][2], "seg": seg}
    return {"raw0": det[0], "raw1": det[1], "raw2": det[2], "seg": seg}


def _check(tag, det, seg, full, tol, skip=()):
    g = golden("net")
    for k, v in _outs(det, seg).items():
        if k in skip:
            continue
        v = v.float().cpu()
        assert list(v.shape) == [int(i) for i in g[f"{tag}/{k}/shape"]]
        if full:
            assert_close(v, g[f"{tag}/{k}/full"], tol, f"{tag} {k}")
        else:
            assert_close(subsample(v), g[f"{tag}/{k}/sub"], tol, f"{tag} {k}")
            np.testing.assert_allclose(stats(v)[:2], g[f"{tag}/{k}/stats"][:2], rtol=50 * tol)


CASES = [("n1_128", 1, 128, 11, True), ("n2_64x96", 2, (64, 96), 12, True), ("n1_640", 1, 640, 1, False)]


@pytest.mark.parametrize("tag,bs,size,seed,full", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 0.4)])
def test_eval_unfused(net, tag, bs, size, seed, full, dtype, tol):
    dsn, m = net
    dsn.set_compute_dtype(dtype)
    try:
        m.eval()
        with torch.no_grad():
            det, seg = m(synth_images(bs, size, seed).cuda())
        # decoded boxes ((2s)^2 * anchor, up to 1492 px) amplify bf16 logit noise; the decode itself is pinned in fp32
        _check(f"{tag}/eval", det, seg, full, tol, skip=("pred",) if dtype == torch.bfloat16 else ())
    finally:
        dsn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("tag,bs,size,seed,full", CASES)
def test_eval_fused(net, tag, bs, size, seed, full):
    """Model.fuse(): folded state_dict equals the reference's, and the fused forward (where PyramidPooling.conv1 DOES apply
    its folded BN, quirk Q1) matches the reference's fused forward."""
    import copy
    _, m = net
    mf = copy.deepcopy(m).eval().fuse()
    g = golden("net")
    fsd = mf.state_dict()
    assert sorted(fsd.keys()) == sorted(str(k) for k in g["fused_sd/keys"])
    for k in g.files:
        if k.startswith("fused_sd/model"):
            assert_close(fsd[k[len("fused_sd/"):]].cpu(), g[k], 1e-5, k)
    with torch.no_grad():
        det, seg = mf(synth_images(bs, size, seed).cuda())
    _check(f"{tag}/fused", det, seg, full, 1e-3)


def test_train_forward_batch_stats(net):
    import copy
    _, m = net
    mt = copy.deepcopy(m).train()
    with torch.no_grad():
        det, seg = mt(synth_images(2, (64, 96), 12).cuda())
    _check("n2_64x96/train", det, seg, True, 1e-3)
    g = golden("net")
    sd = mt.state_dict()
    for k in g.files:
        if k.startswith("n2_64x96/train/after/"):
            assert_close(sd[k.split("/after/")[1]].cpu(), g[k], 1e-3, k)
    assert int(sd["model.0.conv.bn.num_batches_tracked"]) == 1


def test_eval_vs_oracle_fresh_input(net):
    """Same seeded input through the HIP model and the CPU oracle (no golden involved)."""
    from oracle import desenet_ref as R
    _, m = net
    cfg = load_cfg()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = synth_images(2, (96, 160), 77)
    with torch.no_grad():
        rdet, rseg, _ = R.forward(cfg, sd, x)
        m.eval()
        det, seg = m(x.cuda())
    assert_close(det[0].cpu(), rdet[0], 1e-3, "pred")
    assert_close(seg.cpu(), rseg, 1e-3, "seg")


def test_config2_at_its_own_batch_fused_fp32_vs_oracle(net):
    """BASELINE.json config 2 at ITS size: 16 x 3 x 640 x 640 fp32, Model.fuse()d, forward + Detect decode against the oracle's fused
    forward on the same seeded batch (the goldens stop at batch 2, and the tile / kernel selection depends on M = N*H*W: batch 16
    takes the thin fp32 Focus kernel, the 128 x 32 and long-K weights-stationary fp32 tiles that smaller batches never reach), then
    non_max_suppression(0.25, 0.45, max_det 1000) on the HIP `pred` against oracle.nms_ref on that SAME pred: selection exact.
    Tolerances: 1e-3 per tensor (max|a-b| / max|b|, BASELINE.json) on pred, the three raw heads and seg, AND element-wise
    |a-b| <= 1e-3 |b| + 1e-3 rms(b) (tests/util.py: elem_err) on the raw heads and seg -- `pred` is excluded from the element-wise
    form only because its w/h columns are (2 sigmoid)^2 * anchor of the raw logits already held element-wise."""
    import copy
    from oracle import desenet_ref as R
    from oracle import nms_ref
    from desenet_amd.core.utils.general import non_max_suppression
    from tests.util import elem_err
    _, m = net
    cfg = load_cfg()
    sd = R.fold_bn({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    mf = copy.deepcopy(m).eval().fuse()
    x = synth_images(16, 640, 2)
    with torch.no_grad():
        (rpred, rraws), rseg, _ = R.forward(cfg, sd, x, fused=True)
        (pred, raws), seg = mf(x.cuda())
    assert tuple(pred.shape) == (16, 25200, 11) and tuple(seg.shape) == (16, 2, 640, 640)
    assert_close(pred.cpu(), rpred, 1e-3, "pred")
    assert_close(seg.cpu(), rseg, 1e-3, "seg")
    assert elem_err(seg.cpu(), rseg) <= 1.0, ("seg element-wise", elem_err(seg.cpu(), rseg))
    for i, (a, b) in enumerate(zip(raws, rraws)):
        assert_close(a.cpu(), b, 1e-3, f"raw{i}")
        assert elem_err(a.cpu(), b) <= 1.0, (f"raw{i} element-wise", elem_err(a.cpu(), b))
    out = non_max_suppression(pred, 0.25, 0.45, max_det=1000)
    want = nms_ref.non_max_suppression(pred.cpu().numpy(), 0.25, 0.45, max_det=1000)
    assert len(out) == 16
    kept = 0
    for o, w in zip(out, want):
        assert tuple(o.shape) == w.shape and np.array_equal(o.cpu().numpy(), w), (o.shape, w.shape)
        kept += len(w)
    assert kept > 0


def test_augmented_inference_vs_oracle(net):
    """Model.forward(augment=True) (yolo.py:331-342): three scales, one flip, de-scaled predictions concatenated -- against the
    oracle's restatement on the same seeded image, 1e-3 per tensor.  (The reference's own TTA path raises once the seg head
    exists; see Model._forward_augment.)"""
    from oracle import desenet_ref as R
    _, m = net
    cfg = load_cfg()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = synth_images(1, (96, 160), 41)
    with torch.no_grad():
        ref, none_ref = R.forward_augment(cfg, sd, x)
        m.eval()
        out, none_hip = m(x.cuda(), augment=True)
    assert none_ref is None and none_hip is None
    n1 = 3 * (12 * 20 + 6 * 10 + 3 * 5)
    assert tuple(out.shape) == tuple(ref.shape) and out.shape[1] > 2 * n1
    assert_close(out.cpu(), ref, 1e-3, "augmented pred")


def _train_step(dsn, m, bs, size, seed, dtype):
    import copy
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from oracle.loss_ref import scale_hyp
    mt = copy.deepcopy(m).train()
    mt.hyp = dict(scale_hyp(6, size), label_smoothing=0.0)
    dsn.set_compute_dtype(dtype)
    try:
        x = synth_images(bs, size, seed).cuda()
        det_t, seg_t = synth_targets(bs, size, seed)
        det_pred, seg_pred = mt(x)
        det_loss, items = ComputeLoss(mt)(det_pred, det_t.cuda())
        seg_loss = SegmentationLosses()(seg_pred, seg_t.cuda())
        (det_loss * 0.14 + seg_loss * 1.0).backward()      # train.py:362-367, both backwards summed
    finally:
        dsn.set_compute_dtype(torch.float32)
    return mt, det_loss, items, seg_loss, det_pred, seg_pred


@pytest.mark.parametrize("tag,bs,size,seed", [("n2_128", 2, 128, 21), ("n1_640", 1, 640, 3)])
@pytest.mark.parametrize("dtype,ltol,gtol", [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 3e-2, 0.25)])
def test_train_step_gradients(net, tag, bs, size, seed, dtype, ltol, gtol):
    """G3: losses, global gradient norm, per-parameter gradient energy, selected full gradients, and the grad-less set."""
    dsn, m = net
    g = golden("train")
    mt, det_loss, items, seg_loss, det_pred, seg_pred = _train_step(dsn, m, bs, size, seed, dtype)
    assert_close(det_loss.cpu(), g[f"{tag}/det_loss"], ltol, "det_loss")
    assert_close(items.cpu(), g[f"{tag}/loss_items"], ltol, "loss_items")
    assert_close(seg_loss.cpu(), g[f"{tag}/seg_loss"], ltol, "seg_loss")
    fp32 = dtype == torch.float32
    if fp32 and tag == "n2_128":
        gtol = 2e-2
    if tag == "n2_128":
        for j in range(3):
            assert_close(det_pred[j].float().cpu(), g[f"{tag}/raw{j}"], 1e-3 if fp32 else 0.6, f"raw{j}")
        assert_close(seg_pred.float().cpu(), g[f"{tag}/seg"], 1e-3 if fp32 else 0.6, "seg")
    params = dict(mt.named_parameters())
    gradless = sorted(k for k, p in params.items() if p.grad is None)
    assert gradless == sorted(str(k) for k in g[f"{tag}/gradless"])
    names = [str(k) for k in g[f"{tag}/grad_names"]]
    sq = np.array([(params[k].grad.double() ** 2).sum().item() for k in names])
    np.testing.assert_allclose(np.sqrt(sq.sum()), g[f"{tag}/grad_l2"], rtol=gtol)
    if not fp32:
        return
    np.testing.assert_allclose(np.sqrt(sq), np.sqrt(g[f"{tag}/grad_sq"]), rtol=4 * gtol, atol=gtol * float(g[f"{tag}/grad_l2"]) * 1e-2)
    for k in g.files:
        if k.startswith(f"{tag}/grad/"):
            name = k.split("/grad/")[1]
            assert_close(params[name].grad.float().cpu(), g[k], gtol, name)


def _force_ping_pong(L, on):
    L.dsn_pp_mode(2 if on else 1); L.dsn_pp1_mode(2 if on else 1)
    L.dsn_wgrad_pp_mode(2 if on else 1); L.dsn_pp_dir(1 if on else 0)


@pytest.mark.parametrize("size,bs", [(256, 2), (640, 2)])
@pytest.mark.parametrize("fused", [False, True])
def test_ping_pong_kernels_forced_eval_forward_is_bit_identical(net, size, bs, fused):
    """Round 4's forward kernels in the real network: the bf16 eval forward (running statistics; `fused`: BatchNorm folded, bias +
    SiLU (+ shortcut) in the conv epilogues) with the ping-pong 3x3 and 1x1 kernels and the register epilogue forced onto EVERY
    eligible layer -- one- and four-patch maps, ragged 8 x 8 patches, partial channel tiles, channel-slice operands of the concat
    buffers -- must equal the default selection BIT FOR BIT on every output (these kernels keep the k order of the ones they
    replace; nothing else in an eval forward depends on summation order)."""
    import copy
    from desenet_amd import _lib, hip_ops as ops
    dsn, m = net
    L = _lib.lib()
    me = copy.deepcopy(m).eval()
    if fused:
        me.fuse()
    x = synth_images(bs, size, 41).cuda()
    dsn.set_compute_dtype(torch.bfloat16)
    outs = {}
    try:
        for forced in (False, True):
            _force_ping_pong(L, forced)
            if forced:
                ops.profile_enable(True)
            with torch.no_grad():
                det, seg = me(x)
            torch.cuda.synchronize()
            if forced:
                labels = " ".join(str(k) for k in ops.profile_collect())
                ops.profile_enable(False)
                assert "conv3x3_pp_kernel" in labels and "conv1x1_pp_kernel" in labels, labels
            outs[forced] = {k: v.clone() for k, v in _outs(det, seg).items()}
    finally:
        _force_ping_pong(L, False)
        ops.profile_enable(False)
        dsn.set_compute_dtype(torch.float32)
    for k in outs[False]:
        assert torch.equal(outs[False][k], outs[True][k]), k


@pytest.mark.parametrize("size,bs", [(256, 2), (640, 1)])
def test_ping_pong_kernel_families_forced_through_a_training_step(net, size, bs):
    """The same kernels (+ the stride-2 data-gradient form and the kernel-row weight gradients) forced through a bf16 TRAINING
    step.  Here the comparison cannot be tight: a default step repeats bit for bit, but a forced one sums BatchNorm partials and
    split-K slabs in another order, and bf16 through 80 batch-statistics layers amplifies a last-bit difference to the scale PyTorch's
    own bf16 autocast shows against fp32 (tools/bf16_noise_floor.py: raw outputs 0.2-0.5 of their range, gradient norm 4 %).  Held:
    losses within 1 %, outputs within 0.3 of their range, gradient norm within 15 %, the same set of parameters with gradients, all
    finite.  (The tight checks of these kernels: bit-identical eval forwards above, test_pp_gpu.py / test_pp1_gpu.py per launch.)"""
    from desenet_amd import _lib, hip_ops as ops
    dsn, m = net
    L = _lib.lib()
    runs = []
    try:
        for forced in (False, False, True):
            _force_ping_pong(L, forced)
            if forced:
                ops.profile_enable(True)
            mt, det_loss, items, seg_loss, det_pred, seg_pred = _train_step(dsn, m, bs, size, 31, torch.bfloat16)
            torch.cuda.synchronize()
            if forced:
                labels = " ".join(str(k) for k in ops.profile_collect())
                ops.profile_enable(False)
                for fam in ("conv3x3_pp_kernel", "conv1x1_pp_kernel"):
                    assert fam in labels, labels
            runs.append((float(det_loss.detach()), float(seg_loss.detach()), [p.detach().float().cpu() for p in det_pred] + [seg_pred.detach().float().cpu()],
                         {k: p.grad.float().cpu() for k, p in mt.named_parameters() if p.grad is not None}))
    finally:
        _force_ping_pong(L, False)
        ops.profile_enable(False)

    def dist(a, b):
        gn = lambda r: sum(float((g.double() ** 2).sum()) for g in r[3].values()) ** 0.5
        return dict(loss=max(abs(a[0] - b[0]) / abs(a[0]), abs(a[1] - b[1]) / abs(a[1])),
                    out=max(float((x - y).abs().max()) / float(x.abs().max()) for x, y in zip(a[2], b[2])),
                    gnorm=abs(gn(a) - gn(b)) / gn(a))
    assert sorted(runs[0][3]) == sorted(runs[2][3])
    assert all(bool(torch.isfinite(g).all()) for g in runs[2][3].values())
    same = dist(runs[0], runs[1])
    assert same["loss"] <= 1e-3 and same["gnorm"] <= 1e-2, same            # a default step repeats (bit for bit on the boxes seen so far)
    forced = dist(runs[0], runs[2])
    assert forced["loss"] <= 1e-2 and forced["out"] <= 0.3 and forced["gnorm"] <= 0.15, forced


def test_train_step_with_merged_c3_pairs_vs_reference(net, monkeypatch):
    """With gradient slots attached (FlatGradients) every C3 runs cv2 | cv1 as ONE convolution + ONE BatchNorm launch
    (conv_impl.pair_block_*): the same reference goldens (G3: losses, outputs, gradient norm, full gradients), fp32."""
    import copy
    from desenet_amd import conv_impl
    from desenet_amd.core.models import common
    from desenet_amd.core.utils.hyp import scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.parallel import FlatGradients
    dsn, m = net
    g, tag = golden("train"), "n2_128"
    calls = []
    real = conv_impl.pair_block_fwd
    monkeypatch.setattr(common, "pair_block_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    # (RFB2's merged pair is additionally held to the reference's module golden in test_modules_gpu: its gradients are part
    #  of the per-parameter energy comparison below)
    mt = copy.deepcopy(m).train()
    mt.hyp = dict(scale_hyp(6, 128), label_smoothing=0.0)
    flat = FlatGradients(mt.parameters())
    flat.zero()
    x = synth_images(2, 128, 21).cuda()
    det_t, seg_t = synth_targets(2, 128, 21)
    det_pred, seg_pred = mt(x)
    det_loss, items = ComputeLoss(mt)(det_pred, det_t.cuda())
    seg_loss = SegmentationLosses()(seg_pred, seg_t.cuda())
    (det_loss * 0.14 + seg_loss * 1.0).backward()
    # eight C3 pairs (cv2 | cv1) and RFB2's two 1x1 convs of the same input
    assert len(calls) == sum(isinstance(mod, (common.C3, common.RFB2)) for mod in mt.modules()) == 9
    assert_close(det_loss.cpu(), g[f"{tag}/det_loss"], 1e-3, "det_loss")
    assert_close(seg_loss.cpu(), g[f"{tag}/seg_loss"], 1e-3, "seg_loss")
    for j in range(3):
        assert_close(det_pred[j].float().cpu(), g[f"{tag}/raw{j}"], 1e-3, f"raw{j}")
    assert_close(seg_pred.float().cpu(), g[f"{tag}/seg"], 1e-3, "seg")
    params = dict(mt.named_parameters())
    names = [str(k) for k in g[f"{tag}/grad_names"]]
    sq = np.array([(params[k].grad.double() ** 2).sum().item() for k in names])
    np.testing.assert_allclose(np.sqrt(sq.sum()), g[f"{tag}/grad_l2"], rtol=2e-2)
    # per-parameter gradient energy: covers cv1 / cv2 weights and BatchNorm parameters of every merged pair
    np.testing.assert_allclose(np.sqrt(sq), np.sqrt(g[f"{tag}/grad_sq"]), rtol=8e-2, atol=2e-2 * float(g[f"{tag}/grad_l2"]) * 1e-2)
    for k in g.files:
        if k.startswith(f"{tag}/grad/"):
            name = k.split("/grad/")[1]
            assert_close(params[name].grad.float().cpu(), g[k], 2e-2, name)


def test_two_backward_calls_like_the_reference(net):
    """scripts/train.py:366-367 calls backward twice on one forward (retain_graph=True): det first, seg second; the
    accumulated gradients must equal one combined backward.  Fixed upstream gradients are used instead of the losses:
    ComputeLoss' `tobj[b, a, gj, gi] = iou` scatter has duplicate indices, which is non-deterministic on a GPU (in the
    reference as well) and would mask what this test is about."""
    import copy
    _, m = net
    x = synth_images(2, 128, 21).cuda()

    def grads_of(two_calls):
        mm = copy.deepcopy(m).train()
        raws, seg = mm(x)
        g = torch.Generator(device="cuda").manual_seed(5)
        g_raws = [torch.rand(r.shape, device="cuda", generator=g) - 0.5 for r in raws]
        g_seg = torch.rand(seg.shape, device="cuda", generator=g) - 0.5
        if two_calls:
            torch.autograd.backward(raws, g_raws, retain_graph=True)
            torch.autograd.backward([seg], [g_seg])
        else:
            torch.autograd.backward(list(raws) + [seg], g_raws + [g_seg])
        return {k: p.grad.clone() for k, p in mm.named_parameters() if p.grad is not None}

    one, two = grads_of(False), grads_of(True)
    assert sorted(one) == sorted(two)
    for k in one:
        assert rel_err(two[k].cpu(), one[k].cpu()) < 1e-3, k


def test_direct_accumulation_into_flat_gradients(net):
    """With FlatGradients every .grad is a view of one buffer and the kernels accumulate into it directly (no autograd
    temporaries): same numbers as the plain autograd path, and a second step after zero() is identical (deterministic).
    With gradient slots present C3 runs cv2 | cv1 as one merged convolution (one K = 2c_ dgrad instead of two accumulated
    ones: another fp32 summation order, amplified by ~70 batch-statistics layers to 5e-5 at layer 0) -- tolerance 2e-4."""
    import copy
    from desenet_amd.parallel import FlatGradients
    _, m = net
    x = synth_images(2, 128, 21).cuda()

    def upstream(raws, seg):
        g = torch.Generator(device="cuda").manual_seed(5)
        return [torch.rand(r.shape, device="cuda", generator=g) - 0.5 for r in raws] + \
               [torch.rand(seg.shape, device="cuda", generator=g) - 0.5]

    plain = copy.deepcopy(m).train()
    raws, seg = plain(x)
    torch.autograd.backward(list(raws) + [seg], upstream(raws, seg))
    ref = {k: p.grad.clone() for k, p in plain.named_parameters() if p.grad is not None}

    direct = copy.deepcopy(m).train()
    flat = FlatGradients(direct.parameters())
    for rep in range(2):
        flat.zero()
        direct.load_state_dict(m.state_dict())          # undo the BN running-stat update of the previous repetition
        raws, seg = direct(x)
        torch.autograd.backward(list(raws) + [seg], upstream(raws, seg))
        for k, p in direct.named_parameters():
            assert p.grad.untyped_storage().data_ptr() == flat.flat.untyped_storage().data_ptr()
            if k in ref:
                assert rel_err(p.grad.cpu(), ref[k].cpu()) < 2e-4, (rep, k)
            else:
                assert float(p.grad.abs().max()) == 0.0, k
    assert int(direct.state_dict()["model.0.conv.bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("fused", [False, True])
def test_graph_replay_equals_eager_training(net, fused):
    """desenet_amd.graph.GraphedTrainStep (one hipGraph per step: pack + forward + HIP losses + backward + SGD) must walk
    the same trajectory as the eager autograd path: one SGD step from the same initial weights, fp32 (tolerance 5e-2: the
    tiny batch-statistics network amplifies rounding differences step over step)."""
    import copy
    from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.graph import GraphedTrainStep
    from desenet_amd.optim import FusedSGD
    from desenet_amd.parallel import FlatGradients, sgd_param_groups
    from desenet_amd import hip_ops as ops
    _, m = net
    x = synth_images(2, 128, 21).cuda()
    det_t, seg_t = synth_targets(2, 128, 21)
    det_t, seg_t = det_t.cuda(), seg_t.cuda()

    def setup():
        mm = copy.deepcopy(m).train()
        mm.hyp = scale_hyp(6, 128)
        flat = FlatGradients(mm.parameters())
        opt = (FusedSGD if fused else torch.optim.SGD)(sgd_param_groups(mm), lr=0.01, momentum=0.937, nesterov=True)
        return mm, flat, opt, ComputeLoss(mm), SegmentationLosses()

    me, flat, opt, cl, sl = setup()
    for _ in range(1):                           # (GraphedTrainStep undoes its 3 warm-up steps before capturing)
        flat.zero()
        det, seg = me(x)
        (cl(det, det_t)[0] * DETGAIN + sl(seg, seg_t) * SEGGAIN).backward()
        opt.step()

    mg, flat_g, opt_g, clg, slg = setup()

    def loss_and_grads(det, seg):
        out, d_det = clg.forward_backward(det, det_t, gain=DETGAIN)
        sout, d_seg = slg.forward_backward(seg, seg_t)
        return out[0] + sout[0] * SEGGAIN, d_det, d_seg

    w0 = {k: v.clone() for k, v in mg.state_dict().items()}
    step = GraphedTrainStep(mg, loss_and_grads, flat_g, opt_g, x, warmup=3)
    for k, v in mg.state_dict().items():         # building the step must not train: warm-up steps are undone
        assert torch.equal(v, w0[k]), k
    loss = step()                                # one replay = the first step
    assert torch.isfinite(loss).all()
    sd_e, sd_g = me.state_dict(), mg.state_dict()
    for k in sd_e:
        if sd_e[k].dtype.is_floating_point:
            assert rel_err(sd_g[k].cpu(), sd_e[k].cpu()) < 5e-2, k   # chaotic 128x128 batch-stat net, float atomics in det_scatter, gain folded differently
        else:
            assert torch.equal(sd_g[k].cpu(), sd_e[k].cpu()), k
    # every replay must re-pack the weights the previous replay's optimizer step produced (a stale bank would train on old
    # bf16/fp32 copies forever): the bank after replay n+1 is exactly pack(master weights after replay n)
    convs = list(mg.__dict__["_dsn_bank"].convs)          # the bank's own order (C3's cv2 | cv1 pairs are adjacent)
    snap = [c.weight.detach().clone() for c in convs]
    step()
    bank = mg.__dict__["_dsn_bank"]
    for c, w0, f, cp in list(zip(convs, snap, bank.fwd, mg.__dict__["_dsn_bank_pads"]))[::7]:
        assert torch.equal(f, ops.pack_weight_fwd(w0, f.dtype, None, cp)), "weight bank is stale after a replay"
        assert not torch.equal(c.weight.detach(), w0), "the optimizer step inside the graph did not update the weights"


def test_graphed_inference_equals_eager(net):
    """desenet_amd.graph.GraphedInference: the fused eval forward replayed from a hipGraph returns what eager launches
    return -- for the capture input and for a new batch copied into the captured buffer (uint8 input path)."""
    import copy
    from desenet_amd.graph import GraphedInference
    _, m = net
    mf = copy.deepcopy(m).eval().fuse()
    g = torch.Generator().manual_seed(9)
    xs = [torch.randint(0, 256, (2, 3, 128, 128), generator=g, dtype=torch.uint8).cuda() for _ in range(2)]
    gi = GraphedInference(mf, xs[0])
    for x in xs:
        with torch.no_grad():
            (p0, r0), s0 = mf(x)
        (p1, r1), s1 = gi(x)
        assert torch.equal(p0, p1) and torch.equal(s0, s1) and all(torch.equal(a, b) for a, b in zip(r0, r1))
    with pytest.raises(ValueError):
        GraphedInference(copy.deepcopy(m).train(), xs[0])


@pytest.mark.parametrize("size,bs", [(128, 2), (256, 2)])
def test_config5_desenet_m_train_step_vs_oracle(size, bs):
    """BASELINE.json config 5's graph (DeSeNet-m: the same yaml at width x1.0 / depth x1.0 -- channels up to 1024, K up to
    9216, 24 bottlenecks) through one fp32 training step against the CPU oracle on the same hash-filled weights: forward
    outputs (batch statistics), both losses and every parameter gradient.  128x128 / batch 2 keeps the oracle to seconds;
    256x256 gives the K = 9216 layers 8x8 .. 64x64 maps; the full 1280x1280 / batch 4 / bf16 shape runs in
    test_config5_full_size_bf16_step_vs_fp32 below."""
    import os
    import yaml
    from desenet_amd import hip_ops  # noqa: F401  (fails loudly without the .so)
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.synth import hash_fill_state_dict
    from oracle import desenet_ref as R
    from oracle import loss_ref
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "desenet_amd", "cfg", "desenet_m.yaml")
    cfg = yaml.safe_load(open(path))
    m = Model(path, ch=3, nc=6)
    sd = m.state_dict()
    hash_fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.hyp = dict(loss_ref.scale_hyp(6, size), label_smoothing=0.0)
    x = synth_images(bs, size, 31)
    det_t, seg_t = synth_targets(bs, size, 31)
    # oracle
    is_p = lambda k: "running" not in k and "num_batches" not in k and "anchor" not in k
    sdo = {k: v.detach().cpu().clone().requires_grad_(is_p(k) and v.dtype.is_floating_point) for k, v in sd.items()}
    raws, seg, _ = R.forward(cfg, sdo, x, training=True)
    total, *_ = loss_ref.step_loss(raws, seg, det_t, seg_t, sdo["model.25.anchors"], 6, size)
    total.backward()
    # HIP
    det_pred, seg_pred = m(x.cuda())
    det_loss, _ = ComputeLoss(m)(det_pred, det_t.cuda())
    seg_loss = SegmentationLosses()(seg_pred, seg_t.cuda())
    loss = det_loss * 0.14 + seg_loss * 1.0
    loss.backward()
    for a, b in zip(det_pred, raws):
        assert_close(a.detach().cpu(), b.detach(), 2e-3, "raw detect output")
    assert_close(seg_pred.detach().cpu(), seg.detach(), 2e-3, "seg logits")
    assert_close(loss.detach().cpu().reshape(-1), total.detach().reshape(-1), 2e-3, "total loss")
    bad = []
    gn_h = gn_o = 0.0
    for k, p in m.named_parameters():
        go = sdo[k].grad
        if go is None and p.grad is None:
            continue
        gh = torch.zeros_like(p) if p.grad is None else p.grad
        go = torch.zeros_like(sdo[k]) if go is None else go
        gn_h += float(gh.double().pow(2).sum()); gn_o += float(go.double().pow(2).sum())
        if go.abs().max() > 1e-6 and rel_err(gh.cpu(), go) > 5e-2:
            bad.append((k, rel_err(gh.cpu(), go)))
    assert abs(gn_h ** 0.5 - gn_o ** 0.5) <= 2e-2 * gn_o ** 0.5, (gn_h, gn_o)
    assert len(bad) <= 3, bad[:10]          # tiny batch-statistics maps (4x4, 2 images) amplify fp32 rounding on a few tensors


def test_config5_full_size_bf16_step_vs_fp32():
    """BASELINE.json config 5 at its stated per-GPU size and dtype: DeSeNet-m, 4 x 3 x 1280 x 1280, bf16 storage / fp32
    accumulation, one full training step (forward, both losses, backward).  The same step in fp32 on the HIP path -- itself held
    to the oracle at 128 / 256 above -- is the yard-stick: losses within 3e-2, finite outputs of the right shapes, finite
    gradients for every parameter that has one, gradient norm within 25 % (the bf16 bound of the DeSeNet-s step).
    A SELF-comparison by necessity (the CPU oracle needs minutes and tens of GB at this size): what it adds over the oracle-checked
    128 / 256 runs is the kernel SELECTION of the full-size maps -- the ping-pong 3x3 / 1x1 / stride-2 kernels, two blocks per CU,
    256-channel tiles, kernel-row weight gradients -- each of which is held bit for bit (or to ATen) at these very grids in
    test_pp_gpu.py / test_pp1_gpu.py (`test_production_grids_*`, `test_wgrad_kernel_row_production_shapes`)."""
    import os
    import desenet_amd
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.synth import hash_fill_state_dict
    from oracle import loss_ref
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "desenet_amd", "cfg", "desenet_m.yaml")
    size, bs = 1280, 4
    x = synth_images(bs, size, 7).cuda()
    det_t, seg_t = synth_targets(bs, size, 7)
    det_t, seg_t = det_t.cuda(), seg_t.cuda()
    res = {}
    try:
        for dtype in (torch.float32, torch.bfloat16):
            desenet_amd.set_compute_dtype(dtype)
            m = Model(path, ch=3, nc=6)
            sd = m.state_dict()
            hash_fill_state_dict(sd)
            m.load_state_dict(sd)
            m = m.cuda().train()
            m.hyp = dict(loss_ref.scale_hyp(6, size), label_smoothing=0.0)
            det_pred, seg_pred = m(x)
            assert [tuple(r.shape) for r in det_pred] == [(bs, 3, size // s, size // s, 11) for s in (8, 16, 32)]
            assert tuple(seg_pred.shape) == (bs, 2, size, size)
            det_loss, items = ComputeLoss(m)(det_pred, det_t)
            seg_loss = SegmentationLosses()(seg_pred, seg_t)
            (det_loss * 0.14 + seg_loss * 1.0).backward()
            assert all(torch.isfinite(r).all() for r in det_pred) and torch.isfinite(seg_pred).all()
            gn = 0.0
            for k, p in m.named_parameters():
                if p.grad is not None:
                    assert torch.isfinite(p.grad).all(), k
                    gn += float(p.grad.double().pow(2).sum())
            res[dtype] = (float(det_loss.detach()), float(seg_loss.detach()), gn ** 0.5, items.detach().cpu())
            del m, det_pred, seg_pred
            torch.cuda.empty_cache()
    finally:
        desenet_amd.set_compute_dtype(torch.float32)
    f, b = res[torch.float32], res[torch.bfloat16]
    assert abs(b[0] - f[0]) <= 3e-2 * abs(f[0]), ("det loss", f[0], b[0])
    assert abs(b[1] - f[1]) <= 3e-2 * abs(f[1]), ("seg loss", f[1], b[1])
    assert abs(b[2] - f[2]) <= 0.25 * f[2], ("gradient norm", f[2], b[2])


@pytest.mark.parametrize("training", [False, True])
def test_teacher_forced_layers_bf16_at_640(net, training):
    """bf16 where the headline runs (config 3: 640 x 640): EVERY top-level layer of the mirrored model is fed the ORACLE's fp32
    input to that layer and its bf16 output is held to the oracle's output -- a per-layer bound that localises divergence
     instead of the loose whole-net bf16 bounds above.  rel err = max|a-b| / max|b| <= 4e-2 per output tensor
    (a C3 with three Bottlenecks is ~10 bf16 roundings deep); eval mode (running statistics) and train mode (batch statistics)."""
    import copy
    from oracle import desenet_ref as R
    dsn, m = net
    cfg = load_cfg()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = synth_images(8, 640, 5)       # the headline batch: kernels see their production grid sizes (several block generations per CU)
    with torch.no_grad():
        out, seg, saved = R.forward(cfg, copy.deepcopy(sd), x, training=training, keep=range(26))
    saved = dict(saved)
    saved[24] = seg
    mm = copy.deepcopy(m).train(training)
    layers = list(mm.model)
    dsn.set_compute_dtype(torch.bfloat16)
    worst = {}
    try:
        for L in layers:
            if L.i == 0:
                inp = x
            elif isinstance(L.f, int):
                inp = saved[L.i - 1 if L.f == -1 else L.f]
            else:
                inp = [saved[L.i + j if j < 0 else j] for j in L.f]
            inp = [t.cuda() for t in inp] if isinstance(inp, list) else inp.cuda()
            with torch.no_grad():
                y = L(inp)
            if L.i == 25:
                ref = out if training else out[1]
                got = y if training else y[1]
                errs = [rel_err(g.float().cpu(), r) for g, r in zip(got, ref)]
            else:
                errs = [rel_err(y.float().cpu(), saved[L.i])]
            worst[L.i] = max(errs)
    finally:
        dsn.set_compute_dtype(torch.float32)
    bad = {i: e for i, e in worst.items() if e > 4e-2}
    assert not bad, (bad, worst)


@pytest.mark.parametrize("dtype,tol,forced", [(torch.float32, 5e-3, False), (torch.bfloat16, 6e-2, False), (torch.bfloat16, 6e-2, True)],
                         ids=["fp32", "bf16", "bf16_ping_pong_kernels_forced"])
def test_teacher_forced_layer_backward_at_640(net, dtype, tol, forced):
    """bf16 BACKWARD where the headline runs (config 3: 8 x 640 x 640), against the ORACLE and per top-level layer, so that a
    dgrad / wgrad / BatchNorm-backward error cannot hide behind a comparison of the HIP path with itself: layer L of the mirrored
    model is fed the oracle's fp32 input to L and the oracle's upstream gradient d(total loss)/d(output of L) (train.py:352-367:
    detgain * det loss + seggain * seg loss), and its bf16 input gradient(s) and parameter gradients are held to the oracle's
    for the same isolated layer (torch.autograd.grad over oracle.desenet_ref.apply_layer in fp32 on the CPU).
    rel err = max|a-b| / max|b| per tensor <= 5e-3 in fp32 and <= 6e-2 in bf16; gradients whose oracle value is numerically zero
    (max|b| < 1e-12: the BatchNorm before a 1x1-map Conv, quirk Q1) must be zero or absent.
    SPP (layer 8) in bf16: its three max pools route each output's gradient to the FIRST maximum of a 5x5 / 9x9 / 13x13 window, and
    on bf16-rounded activations the largest values of a window tie in a sizeable share of the windows -- the first of the tied
    pixels is not the pixel whose fp32 value is largest (exactly what ATen's own bf16 max_pool2d does).  The oracle for that layer
    is therefore TEACHER-FORCED with the pool input the kernels saw (cv1's bf16 output, straight-through for the gradient, as in
    tests/test_modules_gpu.py::_spp_reference_with_the_kernels_pool_input), so both sides route through the same ties with the
    same first-maximum rule and the layer is held to the same 6e-2 per tensor as every other one (round 3 bounded its gradient
    NORM by 25 % instead, which bounds nothing per element).
    `forced` (round 4): the same with the ping-pong 3x3 / 1x1 / stride-2-dgrad kernels, the kernel-row weight gradients and the
    register epilogue forced onto every eligible layer, so that each of them meets the ORACLE inside the real layers (batch 8
    selects them for four launches only by default)."""
    import copy
    import torch.nn.functional as F
    from oracle import desenet_ref as R
    from oracle import loss_ref
    from desenet_amd import _lib
    dsn, m = net
    if forced:
        _force_ping_pong(_lib.lib(), True)
    cfg = load_cfg()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    is_p = lambda k: "running" not in k and "num_batches" not in k and "anchor" not in k
    x = synth_images(8, 640, 5)
    det_t, seg_t = synth_targets(8, 640, 5)
    # ---- oracle: whole step once, upstream gradient of every layer output
    sd = {k: v.clone().requires_grad_(is_p(k)) for k, v in sd0.items()}
    layers, _ = R.parse_arch(cfg, 3)
    cx = R.Ctx(sd, training=True)
    ys, out = [], x
    for L in layers:
        if L.f != -1:
            out = ys[L.f] if isinstance(L.f, int) else [out if j == -1 else ys[j] for j in L.f]
        out = R.apply_layer(cx, L, out)
        for t in (out if isinstance(out, (list, tuple)) else [out]):
            t.retain_grad()
        ys.append(out)
    total, *_ = loss_ref.step_loss(ys[25], ys[24], det_t, seg_t, sd["model.25.anchors"], 6, 640)
    total.backward()
    ups = [[t.grad for t in y] if isinstance(y, (list, tuple)) else y.grad for y in ys]
    ins_of = lambda L: (x if L.i == 0 else ys[L.i - 1] if L.f == -1 else ys[L.f] if isinstance(L.f, int)
                        else [ys[L.i + j if j < 0 else j] for j in L.f])
    mm = copy.deepcopy(m).train()
    dsn.set_compute_dtype(dtype)
    worst, bad = {}, {}
    try:
        for L, HL in zip(layers, list(mm.model)):
            if ups[L.i] is None or (isinstance(ups[L.i], list) and any(u is None for u in ups[L.i])):
                continue
            raw = ins_of(L)
            lst = isinstance(raw, (list, tuple))
            # oracle, isolated layer (fresh running statistics: BatchNorm's train-mode forward updates them in place)
            sdl = {k: v.clone().requires_grad_(is_p(k)) for k, v in sd0.items()}
            oin = [t.detach().clone().requires_grad_(L.i != 0) for t in (raw if lst else [raw])]
            if dtype == torch.bfloat16 and L.kind == "SPP":
                # tie-aware reference: the oracle's pools see the pool input the KERNELS saw (cv1's bf16 output, straight-through
                # for the gradient), so both sides route every window's gradient through the same first maximum
                with torch.no_grad():
                    z0_hip = copy.deepcopy(HL).train().cv1(oin[0].detach().cuda()).float().cpu()
                cxl = R.Ctx(sdl, training=True)
                z0 = R.conv_bn_act(cxl, oin[0], f"model.{L.i}.cv1", 1)
                z0 = z0 + (z0_hip - z0).detach()
                oout = R.conv_bn_act(cxl, torch.cat([z0] + [F.max_pool2d(z0, k, 1, k // 2) for k in L.args[2]], 1),
                                     f"model.{L.i}.cv2", 1)
            else:
                oout = R.apply_layer(R.Ctx(sdl, training=True), L, oin if lst else oin[0])
            pk = [k for k in sdl if k.startswith(f"model.{L.i}.") and sdl[k].requires_grad]
            oo = list(oout) if isinstance(oout, (list, tuple)) else [oout]
            uu = ups[L.i] if isinstance(ups[L.i], list) else [ups[L.i]]
            want = [t for t in oin if t.requires_grad] + [sdl[k] for k in pk]
            og = torch.autograd.grad(oo, want, uu, allow_unused=True)
            n_in = sum(t.requires_grad for t in oin)
            # HIP, bf16
            hin = [t.detach().clone().cuda().requires_grad_(L.i != 0) for t in (raw if lst else [raw])]
            for p_ in HL.parameters():
                p_.grad = None
            hout = HL(hin if lst else hin[0])
            ho = list(hout) if isinstance(hout, (list, tuple)) else [hout]
            torch.autograd.backward(ho, [u.cuda().to(o.dtype) for u, o in zip(uu, ho)])
            hp = dict(HL.named_parameters())
            got = [t.grad for t in hin if t.requires_grad] + [hp[k[len(f"model.{L.i}."):]].grad for k in pk]
            names = [f"dx{j}" for j in range(n_in)] + pk
            for nme, g, o in zip(names, got, og):
                if o is None or float(o.abs().max()) < 1e-12:
                    if g is not None and float(g.abs().max()) > 1e-6:
                        bad[(L.i, nme)] = "oracle gradient is zero / absent, HIP's is not"
                    continue
                assert g is not None, (L.i, nme, "no HIP gradient")
                e = rel_err(g.float().cpu(), o)
                worst[(L.i, nme)] = e
                if not e <= tol:
                    bad[(L.i, nme)] = e
    finally:
        dsn.set_compute_dtype(torch.float32)
        _force_ping_pong(_lib.lib(), False)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:8]
    assert len(worst) > 150 and not bad, (bad, top)
