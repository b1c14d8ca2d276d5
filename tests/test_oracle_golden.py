"""Pins the CPU oracle (oracle/) to golden vectors produced by running the real reference (tools/gen_golden.py).

CPU only.  Tolerance: the oracle issues the same ATen ops in the same order, so fp32 results agree to a few ulp;
1e-5 relative per block (1e-4 for gradients and for whole-net outputs, which chain ~80 layers)."""
import ast

import numpy as np
import pytest
import torch

from oracle import desenet_ref as R
from oracle import loss_ref, nms_ref
from desenet_amd.synth import synth_images, synth_targets, synthetic_checkpoint
from tests.util import assert_close, case_weights, golden, load_cfg, stats, subsample

TOL, GTOL = 1e-5, 1e-4


def _flat(y):
    if torch.is_tensor(y):
        return [y]
    out = []
    for e in y:
        out.extend(_flat(e))
    return out


def _run_block(case, fn, n_in, train, list_input=False):
    g = golden("modules")
    w = case_weights(g, case)
    sd = {"m." + k: v.clone().requires_grad_(train and v.dtype.is_floating_point and "running" not in k
                                             and "anchor" not in k) for k, v in w.items()}
    xs = [torch.from_numpy(g[f"{case}/x{j}"]).requires_grad_(train) for j in range(n_in)]
    cx = R.Ctx(sd, training=train, fused=False)
    y = fn(cx, xs if list_input else xs[0])
    outs = _flat(y)
    for j, o in enumerate(outs):
        assert_close(o, g[f"{case}/y{j}"], TOL, f"{case} y{j}")
    if train:
        gs = [torch.from_numpy(g[f"{case}/gy{j}"]) for j in range(len(outs))]
        torch.autograd.backward(outs, gs)
        for j, x in enumerate(xs):
            if f"{case}/dx{j}" in g.files:
                assert_close(x.grad, g[f"{case}/dx{j}"], GTOL, f"{case} dx{j}")
        for k in w:
            gk = f"{case}/dw/{k}"
            if gk in g.files:
                ref = g[gk]
                got = sd["m." + k].grad
                if ref.size == 0:
                    assert got is None or float(got.abs().max()) == 0.0, f"{case} {k} must be grad-less"
                else:
                    assert_close(got, ref, GTOL, f"{case} dw {k}")
            ak = f"{case}/after/{k}"
            if ak in g.files:
                assert_close(sd["m." + k], g[ak], TOL, f"{case} running {k}")


BLOCKS = {
    "conv_k1": (lambda cx, x: R.conv_bn_act(cx, x, "m", 1, 1), 1),
    "conv_k3s1": (lambda cx, x: R.conv_bn_act(cx, x, "m", 3, 1), 1),
    "conv_k3s2": (lambda cx, x: R.conv_bn_act(cx, x, "m", 3, 2), 1),
    "conv_q1": (lambda cx, x: R.conv_bn_act(cx, x, "m", 1, 1), 1),
    "conv_noact": (lambda cx, x: R.conv_bn_act(cx, x, "m", 1, 1, act=False), 1),
    "focus": (lambda cx, x: R.focus(cx, x, "m", 3), 1),
    "bneck_add": (lambda cx, x: R.bottleneck(cx, x, "m", True), 1),
    "bneck_noadd": (lambda cx, x: R.bottleneck(cx, x, "m", False), 1),
    "c3_n1": (lambda cx, x: R.c3(cx, x, "m", 1, True), 1),
    "c3_n3": (lambda cx, x: R.c3(cx, x, "m", 3, True), 1),
    "c3_n1_noshort": (lambda cx, x: R.c3(cx, x, "m", 1, False), 1),
    "spp": (lambda cx, x: R.spp(cx, x, "m", (5, 9, 13)), 1),
    "rfb2": (lambda cx, x: R.rfb2(cx, x, "m"), 1),
    "pyramid": (lambda cx, x: R.pyramid_pooling(cx, x, "m"), 1),
    "ffm": (lambda cx, x: R.ffm(cx, x, "m"), 1),
    "segpsp": (lambda cx, xs: R.seg_mask_psp(cx, xs, "m"), 3),
    "detect": (lambda cx, xs: R.detect(cx, xs, "m"), 3),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_block_vs_reference(name, mode):
    fn, n_in = BLOCKS[name]
    _run_block(f"{name}_{mode}", fn, n_in, mode == "train", list_input=n_in > 1)


def test_q1_bn_skipped_on_1x1_maps():
    """Quirk Q1 (common.py:53): un-fused Conv on a 1x1 map never applies BN, in train and eval, and BN gets no grad."""
    g = golden("modules")
    assert g["conv_q1_train/dw/bn.weight"].size == 0 and g["conv_q1_train/dw/bn.bias"].size == 0
    w = case_weights(g, "conv_q1_eval")
    x = torch.from_numpy(g["conv_q1_eval/x0"])
    y = R.silu(torch.nn.functional.conv2d(x, w["conv.weight"]))
    assert_close(y, g["conv_q1_eval/y0"], TOL)


@pytest.mark.parametrize("case,k,s", [("conv_k3s2_fused", 3, 2), ("conv_q1_fused", 1, 1)])
def test_fold_bn(case, k, s):
    """fuse_conv_and_bn (torch_utils.py:196-216) + forward_fuse; the fused Q1 conv DOES apply the folded BN."""
    g = golden("modules")
    w = case_weights(g, case)
    folded = R.fold_bn({"m." + kk: v for kk, v in w.items()})
    wf = case_weights(g, case, "wf")
    for kk, v in wf.items():
        assert_close(folded["m." + kk], v, TOL, kk)
    x = torch.from_numpy(g[f"{case}/x0"])
    y = R.conv_bn_act(R.Ctx(folded, fused=True), x, "m", k, s)
    assert_close(y, g[f"{case}/y0"], TOL)


def test_focus_slicing_bit_exact():
    g = golden("modules")
    x = torch.from_numpy(g["focus_s2d/x0"])
    y = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    assert np.array_equal(y.numpy(), g["focus_s2d/y0"])


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_upsample_concat(mode):
    g = golden("modules")
    c = f"upcat_{mode}"
    a, b = torch.from_numpy(g[c + "/x0"]), torch.from_numpy(g[c + "/x1"])
    y = torch.cat([torch.nn.functional.interpolate(a, scale_factor=2.0, mode="nearest"), b], 1)
    assert np.array_equal(y.numpy(), g[c + "/y0"])


# ---------------------------------------------------------------------------------------- whole net
@pytest.fixture(scope="module")
def net_sd():
    cfg = load_cfg()
    sd = R.make_state_dict(cfg)
    synthetic_checkpoint(sd)
    return cfg, sd


def test_state_dict_contract(net_sd):
    """Same keys, order and shapes as the reference Model.state_dict(); 7,747,109 parameters (README.md:17-44)."""
    cfg, sd = net_sd
    g = golden("net")
    assert list(sd.keys()) == [str(k) for k in g["meta/keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["meta/shapes"]]
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k and "anchor" not in k)
    assert n == int(g["meta/n_params"]) == 7747109
    assert_close(sd["model.25.anchors"], g["meta/anchors"], 0)
    assert_close(sd["model.25.anchor_grid"], g["meta/anchor_grid"], 0)
    _, save = R.parse_arch(cfg)
    assert save == [int(i) for i in g["meta/save"]]


def _check_forward(tag, det, seg, saved, full):
    g = golden("net")
    if isinstance(det, tuple):
        outs = {"pred": det[0], "raw0": det[1][0], "raw1": det[1][1], "raw2": det[1][2], "seg": seg}
    else:
        outs = {"raw0": det[0], "raw1": det[1], "raw2": det[2], "seg": seg}
    for k, v in outs.items():
        assert list(v.shape) == [int(i) for i in g[f"{tag}/{k}/shape"]]
        if full:
            assert_close(v, g[f"{tag}/{k}/full"], 1e-4, f"{tag} {k}")
        else:
            assert_close(subsample(v), g[f"{tag}/{k}/sub"], 1e-4, f"{tag} {k}")
            np.testing.assert_allclose(stats(v), g[f"{tag}/{k}/stats"], rtol=1e-4)
    for i, v in saved.items():
        if f"{tag}/layer{i}/sub" in g.files:
            assert_close(subsample(v), g[f"{tag}/layer{i}/sub"], 1e-4, f"{tag} layer{i}")


@pytest.mark.parametrize("tag,bs,size,seed,full", [("n1_128", 1, 128, 11, True), ("n2_64x96", 2, (64, 96), 12, True),
                                                   ("n1_640", 1, 640, 1, False)])
@pytest.mark.parametrize("fused", [False, True])
def test_net_eval(net_sd, tag, bs, size, seed, full, fused):
    cfg, sd = net_sd
    x = synth_images(bs, size, seed)
    with torch.no_grad():
        use = R.fold_bn(sd) if fused else sd
        det, seg, saved = R.forward(cfg, use, x, training=False, fused=fused)
    _check_forward(f"{tag}/{'fused' if fused else 'eval'}", det, seg, saved, full)


def test_fold_bn_whole_net(net_sd):
    _, sd = net_sd
    g = golden("net")
    f = R.fold_bn(sd)
    assert sorted(f.keys()) == sorted(str(k) for k in g["fused_sd/keys"])
    for k in g.files:
        if k.startswith("fused_sd/model"):
            assert_close(f[k[len("fused_sd/"):]], g[k], TOL, k)


def test_net_train_forward(net_sd):
    cfg, sd = net_sd
    sd = {k: v.clone() for k, v in sd.items()}
    x = synth_images(2, (64, 96), 12)
    with torch.no_grad():
        det, seg, saved = R.forward(cfg, sd, x, training=True)
    _check_forward("n2_64x96/train", det, seg, saved, True)
    g = golden("net")
    for k in g.files:
        if k.startswith("n2_64x96/train/after/"):
            assert_close(sd[k.split("/after/")[1]], g[k], TOL, k)


@pytest.mark.parametrize("tag,bs,size,seed", [("n2_128", 2, 128, 21), ("n1_640", 1, 640, 3)])
def test_train_step(net_sd, tag, bs, size, seed):
    """G3: losses, per-parameter gradient checksums, and the grad-less set {PyramidPooling.conv1.bn.*} (Q1)."""
    cfg, sd0 = net_sd
    g = golden("train")
    is_param = lambda k: "running" not in k and "num_batches" not in k and "anchor" not in k
    sd = {k: v.clone().requires_grad_(is_param(k)) for k, v in sd0.items()}
    x = synth_images(bs, size, seed)
    det_t, seg_t = synth_targets(bs, size, seed)
    np.testing.assert_array_equal(det_t.numpy(), g[f"{tag}/det_targets"])
    raws, seg, _ = R.forward(cfg, sd, x, training=True)
    total, dl, items, sl = loss_ref.step_loss(raws, seg, det_t, seg_t, sd["model.25.anchors"], 6, size)
    assert_close(dl, g[f"{tag}/det_loss"], 1e-5, "det_loss")
    assert_close(items, g[f"{tag}/loss_items"], 1e-5, "loss_items")
    assert_close(sl, g[f"{tag}/seg_loss"], 1e-5, "seg_loss")
    total.backward()
    gradless = sorted(k for k, v in sd.items() if is_param(k) and v.grad is None)
    assert gradless == sorted(str(k) for k in g[f"{tag}/gradless"])
    names = [str(k) for k in g[f"{tag}/grad_names"]]
    sq = np.array([(sd[k].grad.double() ** 2).sum().item() for k in names])
    np.testing.assert_allclose(sq, g[f"{tag}/grad_sq"], rtol=2e-3, atol=1e-12)
    np.testing.assert_allclose(np.sqrt(sq.sum()), g[f"{tag}/grad_l2"], rtol=1e-4)
    for k in g.files:
        if k.startswith(f"{tag}/grad/"):
            assert_close(sd[k.split("/grad/")[1]].grad, g[k], 2e-4, k)


# ---------------------------------------------------------------------------------------------- NMS
NMS_CASES = ["default", "val_multilabel", "agnostic", "classes", "ties", "maxdet", "empty", "over30000", "iou_edge"]


@pytest.mark.parametrize("case", NMS_CASES)
def test_nms_stages(case):
    """Everything around the greedy step is pinned by the reference's own non_max_suppression (general.py:659-750)."""
    g = golden("nms")
    kw = ast.literal_eval(str(g[f"{case}/kw"]))
    out = nms_ref.non_max_suppression(g[f"{case}/pred"], **kw)
    assert [o.shape[0] for o in out] == [int(n) for n in g[f"{case}/n"]]
    for i, o in enumerate(out):
        np.testing.assert_array_equal(o, g[f"{case}/out{i}"])


def test_nms_strict_threshold_and_ties():
    """Published torchvision semantics: suppress iff IoU > thr (strict); equal scores keep the lower index first."""
    b = np.array([[0, 0, 2, 2], [1, 0, 3, 2], [0.5, 0, 2.5, 2]], np.float32)
    s = np.array([0.9, 0.9, 0.9], np.float32)
    thr = float(np.float32(2.0) / np.float32(6.0))
    assert nms_ref.nms_greedy(b, s, thr).tolist() == [0, 1]      # IoU(0,1)=1/3 not > thr; box 2 overlaps 0 at 0.6
    assert nms_ref.nms_greedy(b, s, 0.7).tolist() == [0, 1, 2]
    assert nms_ref.nms_greedy(b[:0], s[:0], 0.5).tolist() == []


@pytest.mark.parametrize("tag", ["focal", "focal_pw", "auto", "focal_auto"])
def test_det_loss_options_vs_reference(tag):
    """Focal loss (loss.py:36-61,106-110) and autobalance (loss.py:113,158-164) of the oracle's det_loss against the reference's
    ComputeLoss run by tools/gen_golden_loss_opts.py: three consecutive calls (the balance list is state), losses 1e-5, gradients
    1e-4, balance 1e-6."""
    g = golden("loss_opts")
    box, obj, cls, cls_pw, obj_pw, anchor_t, gamma, auto = [float(v) for v in g[f"{tag}/hyp"]]
    hyp = dict(box=box, obj=obj, cls=cls, cls_pw=cls_pw, obj_pw=obj_pw, anchor_t=anchor_t, fl_gamma=gamma)
    anchors = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float().view(3, 3, 2) \
        / torch.tensor([8., 16., 32.]).view(3, 1, 1)
    det_t = torch.from_numpy(g[f"{tag}/targets"])
    balance = [4.0, 1.0, 0.4]
    for step in range(3):
        p = [torch.from_numpy(g[f"{tag}/{step}/p{i}"]).clone().requires_grad_(True) for i in range(3)]
        loss, items = loss_ref.det_loss(p, det_t, anchors, hyp, 6, balance=balance, autobalance=bool(auto), ssi=1)
        loss.sum().backward()
        assert_close(loss.detach(), g[f"{tag}/{step}/loss"], 1e-5, "loss")
        assert_close(items, g[f"{tag}/{step}/items"], 1e-5, "items")
        for i in range(3):
            assert_close(p[i].grad, g[f"{tag}/{step}/dp{i}"], 1e-4, f"dp{i}")
        assert np.allclose(balance, g[f"{tag}/{step}/balance"], rtol=1e-6, atol=0)
