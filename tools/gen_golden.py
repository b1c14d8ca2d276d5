#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING AND RUNNING THE REAL REFERENCE (build container only).

Recipe: cwd=/root/reference, RANK=1 (skips plots.check_font), MagicMock stubs for the optional
third-party imports the hot path never calls (cv2, torchvision, seaborn, imgviz, thop), yaml edited to the
README/BASELINE graph (se_nc=2, SegMaskPSP active).  One in-process patch: loss.py:218 clamps an int64 tensor
with float-tensor bounds, which torch >= 1.12 rejects -> integer bounds.  `torchvision.ops.nms` (absent here) is
monkey-patched to oracle.nms_ref.nms_greedy for the NMS-stage fixtures (greedy step itself stays unpinned).

The fixtures are DATA (inputs + expected outputs); no reference source is written to the repo.

    cd /root/repo && python tools/gen_golden.py
"""
import os
import sys
import unittest.mock as um

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
os.environ["RANK"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
for name in ["cv2", "torchvision", "torchvision.ops", "seaborn", "imgviz", "thop"]:
    sys.modules[name] = um.MagicMock()
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import copy  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from core.models import common as C  # noqa: E402  (reference)
from core.models import yolo as Y  # noqa: E402  (reference)
from core.utils import general as G  # noqa: E402  (reference)
from core.utils import loss as L  # noqa: E402  (reference)
from core.utils.torch_utils import fuse_conv_and_bn, initialize_weights  # noqa: E402  (reference)

from desenet_amd.synth import (BN_CALIB, hash_fill_state_dict, hash_uniform, load_bn_calibration, synth_images,
                               synth_targets)  # noqa: E402
from oracle import nms_ref  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def npy(t):
    return t.detach().cpu().numpy().copy()


def hgrad(tag, like):
    """Deterministic upstream gradient for an output tensor."""
    import zlib
    v = hash_uniform(zlib.crc32(("grad:" + tag).encode()) & 0xFFFFFFFF, like.numel(), -1.0, 1.0)
    return torch.from_numpy(v).view(like.shape)


def hinput(tag, shape, lo=-1.0, hi=1.0):
    import zlib
    n = int(np.prod(shape))
    return torch.from_numpy(hash_uniform(zlib.crc32(("in:" + tag).encode()) & 0xFFFFFFFF, n, lo, hi)).view(shape)


def flat_outputs(y):
    if torch.is_tensor(y):
        return [y]
    out = []
    for e in y:
        out.extend(flat_outputs(e))
    return out


def run_module_case(store, name, module, inputs, train, is_list_input=False):
    """Run reference `module` on `inputs` (list of tensors), record weights/in/out (+grads, BN state in train)."""
    initialize_weights(module)
    sd = module.state_dict()
    hash_fill_state_dict({f"{name}.{k}": v for k, v in sd.items()})
    module.load_state_dict(sd)
    module.train(train)
    for k, v in module.state_dict().items():
        store[f"{name}/w/{k}"] = npy(v)
    xs = [x.clone().requires_grad_(train and x.dtype.is_floating_point) for x in inputs]
    for j, x in enumerate(xs):
        store[f"{name}/x{j}"] = npy(x)
    y = module(list(xs)) if is_list_input else module(*xs)
    outs = flat_outputs(y)
    for j, o in enumerate(outs):
        store[f"{name}/y{j}"] = npy(o)
    if train:
        gs = [hgrad(f"{name}/y{j}", o) for j, o in enumerate(outs)]
        for j, g in enumerate(gs):
            store[f"{name}/gy{j}"] = npy(g)
        torch.autograd.backward(outs, gs)
        for j, x in enumerate(xs):
            if x.grad is not None:
                store[f"{name}/dx{j}"] = npy(x.grad)
        for k, p in module.named_parameters():
            store[f"{name}/dw/{k}"] = npy(p.grad) if p.grad is not None else np.zeros((0,), np.float32)
        for k, v in module.state_dict().items():
            if "running" in k:
                store[f"{name}/after/{k}"] = npy(v)
    return module


def fuse_module(m):
    """What Model.fuse() does to every Conv instance (yolo.py:409-417)."""
    for sub in m.modules():
        if isinstance(sub, C.Conv) and hasattr(sub, "bn"):
            sub.conv = fuse_conv_and_bn(sub.conv, sub.bn)
            delattr(sub, "bn")
            sub.forward = sub.forward_fuse
    return m


def gen_modules():
    s = {}
    # a1: Conv
    for tr in (True, False):
        t = "train" if tr else "eval"
        run_module_case(s, f"conv_k1_{t}", C.Conv(16, 24, 1, 1), [hinput("ck1", (2, 16, 9, 7))], tr)
        run_module_case(s, f"conv_k3s1_{t}", C.Conv(8, 16, 3, 1), [hinput("ck3", (2, 8, 10, 6))], tr)
        run_module_case(s, f"conv_k3s2_{t}", C.Conv(8, 16, 3, 2), [hinput("ck3s2", (2, 8, 11, 7))], tr)
        run_module_case(s, f"conv_q1_{t}", C.Conv(16, 8, 1), [hinput("cq1", (2, 16, 1, 1))], tr)  # quirk Q1
        run_module_case(s, f"conv_noact_{t}", C.Conv(8, 8, 1, 1, act=False), [hinput("cna", (1, 8, 5, 5))], tr)
    # fused (eval) incl. the Q1 case, where fuse() DOES apply the folded BN
    for nm, mod, x in [("conv_k3s2_fused", C.Conv(8, 16, 3, 2), hinput("ck3s2", (2, 8, 11, 7))),
                       ("conv_q1_fused", C.Conv(16, 8, 1), hinput("cq1", (2, 16, 1, 1)))]:
        initialize_weights(mod)
        sd = mod.state_dict()
        hash_fill_state_dict({f"{nm}.{k}": v for k, v in sd.items()})
        mod.load_state_dict(sd)
        mod.eval()
        for k, v in mod.state_dict().items():
            s[f"{nm}/w/{k}"] = npy(v)
        fuse_module(mod)
        for k, v in mod.state_dict().items():
            s[f"{nm}/wf/{k}"] = npy(v)
        s[f"{nm}/x0"] = npy(x)
        s[f"{nm}/y0"] = npy(mod(x))
    # a2: Focus -- arange image, slicing must be bit-exact
    xi = torch.arange(1 * 3 * 8 * 12, dtype=torch.float32).view(1, 3, 8, 12)
    s["focus_s2d/x0"] = npy(xi)
    s["focus_s2d/y0"] = npy(torch.cat([xi[..., ::2, ::2], xi[..., 1::2, ::2], xi[..., ::2, 1::2], xi[..., 1::2, 1::2]], 1))
    for tr in (True, False):
        t = "train" if tr else "eval"
        run_module_case(s, f"focus_{t}", C.Focus(3, 16, 3), [hinput("foc", (2, 3, 12, 16), 0, 1)], tr)
        # a3/a4
        run_module_case(s, f"bneck_add_{t}", C.Bottleneck(16, 16, True), [hinput("bn1", (2, 16, 8, 6))], tr)
        run_module_case(s, f"bneck_noadd_{t}", C.Bottleneck(16, 16, False), [hinput("bn2", (2, 16, 8, 6))], tr)
        run_module_case(s, f"c3_n1_{t}", C.C3(16, 16, 1), [hinput("c31", (2, 16, 8, 6))], tr)
        run_module_case(s, f"c3_n3_{t}", C.C3(16, 32, 3), [hinput("c33", (2, 16, 6, 8))], tr)
        run_module_case(s, f"c3_n1_noshort_{t}", C.C3(32, 16, 1, False), [hinput("c3n", (2, 32, 6, 6))], tr)
        # a5
        run_module_case(s, f"spp_{t}", C.SPP(16, 16, (5, 9, 13)), [hinput("spp", (2, 16, 15, 11))], tr)
        # a6
        up = torch.nn.Upsample(None, 2, "nearest")
        a, b = hinput("uca", (2, 8, 5, 7)).requires_grad_(tr), hinput("ucb", (2, 4, 10, 14)).requires_grad_(tr)
        yc = C.Concat(1)([up(a), b])
        s[f"upcat_{t}/x0"], s[f"upcat_{t}/x1"], s[f"upcat_{t}/y0"] = npy(a), npy(b), npy(yc)
        if tr:
            gy = hgrad(f"upcat_{t}/y0", yc)
            yc.backward(gy)
            s[f"upcat_{t}/gy0"], s[f"upcat_{t}/dx0"], s[f"upcat_{t}/dx1"] = npy(gy), npy(a.grad), npy(b.grad)
        # a10-a12, a9
        run_module_case(s, f"rfb2_{t}", C.RFB2(48, 16, map_reduce=6, d=[2, 3]), [hinput("rfb", (2, 48, 12, 10))], tr)
        run_module_case(s, f"pyramid_{t}", C.PyramidPooling(16, k=[1, 2, 3, 6], short_cut=True),
                        [hinput("pp", (2, 16, 20, 14))], tr)
        run_module_case(s, f"ffm_{t}", C.FFM(32, 16, k=3, is_cat=False), [hinput("ffm", (2, 32, 8, 6))], tr)
        run_module_case(s, f"segpsp_{t}", Y.SegMaskPSP(2, 1, 24, False, ch=(16, 32, 64)),
                        [hinput("sp8", (2, 16, 8, 12)), hinput("sp16", (2, 32, 4, 6)), hinput("sp32", (2, 64, 2, 3))],
                        tr, is_list_input=True)
        # a7: Detect (stride / anchors set as Model.__init__ does)
        anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
        det = Y.Detect(6, anchors, ch=(16, 32, 64))
        det.stride = torch.tensor([8.0, 16.0, 32.0])
        det.anchors /= det.stride.view(-1, 1, 1)
        run_module_case(s, f"detect_{t}", det,
                        [hinput("d8", (2, 16, 8, 12)), hinput("d16", (2, 32, 4, 6)), hinput("d32", (2, 64, 2, 3))],
                        tr, is_list_input=True)
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **s)
    print("modules.npz", len(s), "arrays")


def _raw_ref_model():
    d = yaml.safe_load(open(os.path.join(REF, "core/models/yolov5s_seg.yaml")))
    d["se_nc"] = 2
    d["head"][-2] = [[16, 19, 22], 1, "SegMaskPSP", ["se_nc", 3, 256, False]]
    m = Y.Model(d, ch=3, nc=6)
    sd = m.state_dict()
    hash_fill_state_dict(sd)
    m.load_state_dict(sd)
    return m


def gen_calibration():
    """One train-mode pass of the hash-filled reference with BN momentum 1 -> running stats = batch stats."""
    m = _raw_ref_model().train()
    bns = [b for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d)]
    for b in bns:
        b.momentum = 1.0
    with torch.no_grad():
        m(synth_images(2, 320, 99))
    out = {}
    for k, v in m.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            b = dict(m.named_modules())[k.rsplit(".", 1)[0]]
            if int(b.num_batches_tracked) > 0:
                out[k] = npy(v)
    np.savez_compressed(BN_CALIB, **out)
    print("bn calibration:", len(out), "tensors ->", BN_CALIB)


def build_ref_model():
    m = _raw_ref_model()
    sd = m.state_dict()
    load_bn_calibration(sd)
    m.load_state_dict(sd)
    return m


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.min().item(), t.max().item(), t.numel()], np.float64)


def subsample(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return npy(f[::step][:n])


def record_forward(s, tag, det, seg, saved, full):
    if isinstance(det, tuple):
        pred, raws = det
        outs = {"pred": pred, "raw0": raws[0], "raw1": raws[1], "raw2": raws[2], "seg": seg}
    else:
        outs = {"raw0": det[0], "raw1": det[1], "raw2": det[2], "seg": seg}
    for k, v in outs.items():
        s[f"{tag}/{k}/stats"] = stats(v)
        s[f"{tag}/{k}/shape"] = np.array(v.shape)
        s[f"{tag}/{k}/{'full' if full else 'sub'}"] = npy(v) if full else subsample(v)
    for i, v in saved.items():
        s[f"{tag}/layer{i}/stats"] = stats(v)
        s[f"{tag}/layer{i}/sub"] = subsample(v)


def ref_forward_with_saved(m, x):
    saved = {}
    hooks = []
    for i in m.save:
        hooks.append(m.model[i].register_forward_hook(lambda mod, inp, out, i=i: saved.__setitem__(i, out)))
    det, seg = m(x)
    for h in hooks:
        h.remove()
    saved = {i: v for i, v in saved.items() if torch.is_tensor(v)}
    return det, seg, saved


def gen_net():
    s = {}
    m = build_ref_model()
    s["meta/n_params"] = np.array(sum(p.numel() for p in m.parameters()))
    s["meta/stride"] = npy(m.stride)
    s["meta/save"] = np.array(m.save)
    s["meta/anchors"] = npy(m.model[-1].anchors)
    s["meta/anchor_grid"] = npy(m.model[-1].anchor_grid)
    keys = list(m.state_dict().keys())
    s["meta/keys"] = np.array(keys)
    s["meta/shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    # Detect bias init as the reference's _initialize_biases leaves it BEFORE hash fill is irrelevant (overwritten).
    cases = [("n1_128", synth_images(1, 128, 11), True), ("n2_64x96", synth_images(2, (64, 96), 12), True),
             ("n1_640", synth_images(1, 640, 1), False)]
    with torch.no_grad():
        for tag, x, full in cases:
            m.eval()
            det, seg, saved = ref_forward_with_saved(m, x)
            record_forward(s, f"{tag}/eval", det, seg, saved, full)
        # train-mode forward (batch stats) on a copy so running stats of `m` stay pristine
        mt = copy.deepcopy(m).train()
        x = synth_images(2, (64, 96), 12)
        det, seg, saved = ref_forward_with_saved(mt, x)
        record_forward(s, "n2_64x96/train", det, seg, saved, True)
        s["n2_64x96/train/after/model.0.conv.bn.running_mean"] = npy(mt.state_dict()["model.0.conv.bn.running_mean"])
        s["n2_64x96/train/after/model.24.out.2.convblk.bn.running_var"] = npy(
            mt.state_dict()["model.24.out.2.convblk.bn.running_var"])
        # fused
        mf = copy.deepcopy(m).eval().fuse()
        for tag, x, full in cases:
            det, seg, saved = ref_forward_with_saved(mf, x)
            record_forward(s, f"{tag}/fused", det, seg, saved, full)
        fsd = mf.state_dict()
        for k in ["model.0.conv.conv.weight", "model.0.conv.conv.bias", "model.24.out.1.conv1.conv.bias",
                  "model.9.cv3.conv.bias"]:
            s[f"fused_sd/{k}"] = npy(fsd[k])
        s["fused_sd/keys"] = np.array(list(fsd.keys()))
    np.savez_compressed(os.path.join(OUT, "net.npz"), **s)
    print("net.npz", len(s), "arrays")
    return m


def patch_loss():
    """loss.py:218 -- integer clamp bounds (see module docstring)."""
    orig = torch.Tensor.clamp_

    def clamp_(self, lo=None, hi=None):
        if torch.is_tensor(hi) and not self.dtype.is_floating_point:
            hi = int(hi)
        if torch.is_tensor(lo) and not self.dtype.is_floating_point:
            lo = int(lo)
        return orig(self, lo, hi)

    torch.Tensor.clamp_ = clamp_


def gen_train(m):
    s = {}
    patch_loss()
    hyp = yaml.safe_load(open(os.path.join(REF, "core/hyp/scratch.yaml")))
    for tag, bs, size, seed in [("n2_128", 2, 128, 21), ("n1_640", 1, 640, 3)]:
        mt = copy.deepcopy(m).train()
        for p in mt.parameters():
            p.grad = None
        h = dict(hyp)
        nl = 3
        h["box"] *= 3.0 / nl
        h["cls"] *= 6 / 80.0 * 3.0 / nl
        h["obj"] *= (size / 640) ** 2 * 3.0 / nl
        h["label_smoothing"] = 0.0
        mt.hyp = h
        mt.de_nc, mt.se_nc = 6, 2
        x = synth_images(bs, size, seed)
        det_t, seg_t = synth_targets(bs, size, seed)
        compute_loss = L.ComputeLoss(mt)
        compute_seg_loss = L.SegmentationLosses()
        det_pred, seg_pred = mt(x)
        det_loss, items = compute_loss(det_pred, det_t)
        seg_loss = compute_seg_loss(seg_pred, seg_t)
        (det_loss * 0.14).backward(retain_graph=True)   # train.py:362,366
        (seg_loss * 1).backward()                       # train.py:363,367
        s[f"{tag}/det_loss"] = npy(det_loss)
        s[f"{tag}/loss_items"] = npy(items)
        s[f"{tag}/seg_loss"] = npy(seg_loss)
        s[f"{tag}/det_targets"] = npy(det_t)
        s[f"{tag}/seg_targets_sum"] = np.array(seg_t.sum().item())
        names, gsum, gabs, gl2 = [], [], [], []
        gradless = []
        total = 0.0
        for k, p in mt.named_parameters():
            if p.grad is None:
                gradless.append(k)
                continue
            g = p.grad.double()
            names.append(k)
            gsum.append(g.sum().item())
            gabs.append(g.abs().sum().item())
            gl2.append((g * g).sum().item())
            total += (g * g).sum().item()
        s[f"{tag}/grad_names"] = np.array(names)
        s[f"{tag}/grad_sum"] = np.array(gsum)
        s[f"{tag}/grad_abs"] = np.array(gabs)
        s[f"{tag}/grad_sq"] = np.array(gl2)
        s[f"{tag}/grad_l2"] = np.array(total ** 0.5)
        s[f"{tag}/gradless"] = np.array(gradless)
        for k in ["model.0.conv.conv.weight", "model.4.m.2.cv2.conv.weight", "model.8.cv2.bn.weight",
                  "model.24.out.0.branch1.0.weight", "model.24.out.2.channel_attention.3.weight",
                  "model.24.out.3.bias", "model.25.m.1.bias", "model.25.m.0.weight", "model.17.cv3.bn.bias"]:
            s[f"{tag}/grad/{k}"] = npy(dict(mt.named_parameters())[k].grad)
        if tag == "n2_128":
            for j, r in enumerate(det_pred):
                s[f"{tag}/raw{j}"] = npy(r)
            s[f"{tag}/seg"] = npy(seg_pred)
        print(tag, "det", det_loss.item(), items.tolist(), "seg", seg_loss.item(), "|g|", total ** 0.5, gradless)
    np.savez_compressed(os.path.join(OUT, "train.npz"), **s)
    print("train.npz", len(s), "arrays")


def gen_nms():
    """Reference non_max_suppression with torchvision.ops.nms patched to the published greedy algorithm."""
    def patched_nms(boxes, scores, thr):
        return torch.from_numpy(nms_ref.nms_greedy(boxes.numpy(), scores.numpy(), float(thr)))

    G.torchvision.ops.nms = patched_nms
    s = {}
    rng = np.random.RandomState(5)

    def make_pred(bs, n, nc=6, obj_hi=1.0, spread=640.0, dup=0):
        p = np.zeros((bs, n, 5 + nc), np.float32)
        p[..., 0:2] = rng.uniform(0, spread, (bs, n, 2))
        p[..., 2:4] = rng.uniform(8, 160, (bs, n, 2))
        p[..., 4] = rng.uniform(0, obj_hi, (bs, n))
        p[..., 5:] = rng.uniform(0, 1, (bs, n, nc))
        if dup:  # exact duplicates => exact score ties and IoU == 1
            p[:, n - dup:] = p[:, :dup]
        return p

    cases = {
        "default": (make_pred(2, 3000), dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
        "val_multilabel": (make_pred(1, 1500), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)),
        "agnostic": (make_pred(1, 800), dict(conf_thres=0.25, iou_thres=0.45, agnostic=True, max_det=1000)),
        "classes": (make_pred(1, 800), dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 4], max_det=1000)),
        "ties": (make_pred(1, 600, dup=200), dict(conf_thres=0.1, iou_thres=0.45, max_det=1000)),
        "maxdet": (make_pred(1, 4000, spread=6000.0), dict(conf_thres=0.05, iou_thres=0.45, max_det=50)),
        "empty": (make_pred(2, 300, obj_hi=0.2), dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
        "over30000": (make_pred(1, 7000, spread=3000.0), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)),
    }
    # IoU exactly at threshold: two unit-offset boxes with IoU == 1/3 and thr == 1/3 (fp32) -> NOT suppressed (strict >)
    edge = np.zeros((1, 3, 11), np.float32)
    edge[0, :, 2:4] = 2.0
    edge[0, 0, 0:2] = (10, 10)
    edge[0, 1, 0:2] = (11, 10)     # IoU with box0 = 2/6
    edge[0, 2, 0:2] = (10.5, 10)   # IoU with box0 = 3/5
    edge[0, :, 4] = (0.9, 0.8, 0.7)
    edge[0, :, 5] = 1.0
    cases["iou_edge"] = (edge, dict(conf_thres=0.25, iou_thres=float(np.float32(2.0) / np.float32(6.0)), max_det=10))
    for name, (pred, kw) in cases.items():
        out = G.non_max_suppression(torch.from_numpy(pred.copy()), **kw)
        s[f"{name}/pred"] = pred
        s[f"{name}/kw"] = np.array(repr(kw))
        s[f"{name}/n"] = np.array([o.shape[0] for o in out])
        for i, o in enumerate(out):
            s[f"{name}/out{i}"] = npy(o)
        print("nms", name, [o.shape[0] for o in out])
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **s)


def gen_metrics():
    """Reference evaluation arithmetic: val.py process_batch, metrics.py ap_per_class / compute_ap and
    the two segmentation counters, run on seeded synthetic predictions / labels."""
    import importlib
    for name in ["pycocotools", "pycocotools.coco", "pycocotools.cocoeval"]:
        sys.modules.setdefault(name, um.MagicMock())
    from core.utils import metrics as M          # (general was imported first above: the two import each other)
    V = importlib.import_module("scripts.val")
    s = {}
    rng = np.random.RandomState(11)
    iouv = torch.linspace(0.5, 0.95, 10)

    def boxes(n, lo=0.0, hi=600.0):
        xy = rng.uniform(lo, hi, (n, 2))
        wh = rng.uniform(10, 120, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(np.float32)

    def det_case(nd, nl, jitter):
        lab_box = boxes(nl)
        lab_cls = rng.randint(0, 6, (nl, 1)).astype(np.float32)
        labels = np.concatenate([lab_cls, lab_box], 1)
        idx = rng.randint(0, max(nl, 1), nd) if nl else np.zeros(nd, int)
        db = (lab_box[idx] + rng.normal(0, jitter, (nd, 4))).astype(np.float32) if nl else boxes(nd)
        dc = np.where(rng.rand(nd) < 0.8, lab_cls[idx, 0] if nl else 0, rng.randint(0, 6, nd)).astype(np.float32)
        conf = rng.uniform(0.01, 1.0, nd).astype(np.float32)
        return np.concatenate([db, conf[:, None], dc[:, None]], 1), labels

    stats = []
    for i, (nd, nl, jit) in enumerate([(40, 12, 6.0), (300, 30, 15.0), (7, 1, 2.0), (25, 9, 0.0), (60, 20, 30.0)]):
        det, lab = det_case(nd, nl, jit)
        if i == 3:
            det[5:10] = det[0:5]          # exact duplicates: equal IoUs for the same label
        correct = V.process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv)
        s[f"pb{i}/det"], s[f"pb{i}/lab"], s[f"pb{i}/correct"] = det, lab, npy(correct)
        stats.append((npy(correct), det[:, 4], det[:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]
    p, r, ap, f1, cls = M.ap_per_class(tp, conf, pcls, tcls, plot=False, names={i: str(i) for i in range(6)})
    s["ap/tp"], s["ap/conf"], s["ap/pcls"], s["ap/tcls"] = tp, conf, pcls, tcls
    s["ap/p"], s["ap/r"], s["ap/ap"], s["ap/f1"], s["ap/cls"] = p, r, ap, f1, cls
    # a class with labels but no predictions and one with predictions but no labels
    tcls2 = np.concatenate([tcls, np.full(5, 7.0)])
    pcls2 = np.where(pcls == 2, 9.0, pcls)
    p, r, ap, f1, cls = M.ap_per_class(tp, conf, pcls2, tcls2, plot=False, names={i: str(i) for i in range(10)})
    s["ap2/pcls"], s["ap2/tcls"] = pcls2, tcls2
    s["ap2/p"], s["ap2/r"], s["ap2/ap"], s["ap2/f1"], s["ap2/cls"] = p, r, ap, f1, cls
    for j, (rec, prec) in enumerate([(np.array([0.1, 0.4, 0.4, 0.8]), np.array([1.0, 0.5, 0.66, 0.4])),
                                     (np.array([]), np.array([])), (np.array([1.0]), np.array([1.0]))]):
        a, mpre, mrec = M.compute_ap(rec, prec)
        s[f"cap{j}/rec"], s[f"cap{j}/prec"], s[f"cap{j}/ap"] = rec, prec, np.array(a)
    for j, (shape, ncls) in enumerate([((2, 3, 24, 32), 3), ((1, 2, 17, 9), 2), ((3, 5, 16, 16), 5)]):
        logits = torch.from_numpy(rng.normal(0, 1, shape).astype(np.float32))
        if j == 0:
            logits[:, 1] = logits[:, 2]     # exact ties between classes: torch.max keeps the first
        target = torch.from_numpy(rng.randint(0, ncls, (shape[0],) + shape[2:]).astype(np.int64))
        correct, labeled = M.batch_pix_accuracy(logits, target)
        inter, union = M.batch_intersection_union(logits, target, ncls)
        s[f"seg{j}/logits"], s[f"seg{j}/target"], s[f"seg{j}/ncls"] = npy(logits), npy(target), np.array(ncls)
        s[f"seg{j}/correct"], s[f"seg{j}/labeled"] = np.array(correct), np.array(labeled)
        s[f"seg{j}/inter"], s[f"seg{j}/union"] = np.asarray(inter), np.asarray(union)
    # the per-image statistics loop of val.run (val.py:231-262, 279-289) over two synthetic "batches" of NMS outputs, composed
    # from the reference's own process_batch / scale_coords / xywh2xyxy / ap_per_class; batch 1 has an image without
    # predictions and one without labels, native shapes differ from the 640x640 network input (letterbox ratio + pad)
    from core.utils.general import scale_coords, xywh2xyxy
    from core.utils.plots import segoutput_to_target
    stats, seen = [], 0
    for b in range(2):
        shapes = [((480, 640), ((1.0, 1.0), (0.0, 80.0))), ((720, 1280), ((0.5, 0.5), (0.0, 140.0))), ((640, 640), None)]
        outs, tgts = [], []
        for si in range(3):
            nd, nl = [(30, 8), (0, 5), (12, 0)][si] if b == 1 else [(25, 6), (40, 10), (9, 3)][si]
            det, lab = det_case(nd, nl, 8.0) if nd else (np.zeros((0, 6), np.float32), det_case(1, nl, 1.0)[1])
            if nl:
                xyxy = lab[:, 1:]
                xywh = np.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0],
                                 xyxy[:, 3] - xyxy[:, 1]], 1)
                tgts.append(np.concatenate([np.full((nl, 1), si, np.float32), lab[:, 0:1], xywh], 1).astype(np.float32))
            outs.append(det.astype(np.float32))
        targets = torch.from_numpy(np.concatenate(tgts, 0))
        for si, pred in enumerate(outs):
            s[f"de{b}/out{si}"] = pred
        s[f"de{b}/targets"] = npy(targets)
        for si, pred in enumerate(torch.from_numpy(o) for o in outs):
            labels = targets[targets[:, 0] == si, 1:]
            nl = len(labels)
            tcls = labels[:, 0].tolist() if nl else []
            shape = shapes[si][0]
            seen += 1
            if len(pred) == 0:
                if nl:
                    stats.append((torch.zeros(0, 10, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
                continue
            predn = pred.clone()
            scale_coords((640, 640), predn[:, :4], shape, shapes[si][1])
            if nl:
                tbox = xywh2xyxy(labels[:, 1:5])
                scale_coords((640, 640), tbox, shape, shapes[si][1])
                labelsn = torch.cat((labels[:, 0:1], tbox), 1)
                correct = V.process_batch(predn, labelsn, iouv)
            else:
                correct = torch.zeros(pred.shape[0], 10, dtype=torch.bool)
            stats.append((correct.cpu(), pred[:, 4].cpu(), pred[:, 5].cpu(), tcls))
    st = [np.concatenate(x, 0) for x in zip(*stats)]
    p, r, ap, f1, ap_class = M.ap_per_class(*st, plot=False, names={i: str(i) for i in range(6)})
    ap50, apm = ap[:, 0], ap.mean(1)
    s["de/summary"] = np.array([p.mean(), r.mean(), ap50.mean(), apm.mean()])
    s["de/nt"] = np.bincount(st[3].astype(np.int64), minlength=6)
    s["de/ap_class"], s["de/seen"] = ap_class, np.array(seen)
    # segoutput_to_target (plots.py:222-229) and the align_corners=False resize of seg_validation (val.py:47)
    import torch.nn.functional as F
    lg = torch.from_numpy(rng.normal(0, 1, (2, 3, 20, 28)).astype(np.float32))
    lg[:, 2, :4] = lg[:, 1, :4]
    s["s2t/logits"] = npy(lg)
    for j, size in enumerate([None, (40, 56), (33, 17), (10, 14)]):
        s[f"s2t/out{j}"] = npy(segoutput_to_target(lg, size))
    for j, size in enumerate([(40, 56), (33, 17), (10, 14), (20, 28)]):
        s[f"s2t/bil{j}"] = npy(F.interpolate(lg, size, mode="bilinear", align_corners=False))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **s)
    print("metrics:", len(s), "arrays")


def gen_trainutils():
    """Reference training-step helpers: torch_utils.ModelEMA over a tiny module whose weights move by
    seeded deltas (three updates, starting near the top of the decay ramp and at its bottom), general.one_cycle samples."""
    from core.utils.general import one_cycle
    from core.utils.torch_utils import ModelEMA
    import torch.nn as nn
    s = {}
    lf = one_cycle(1, 0.2, 300)
    xs = np.array([0, 1, 3, 17, 150, 299, 300], dtype=np.float64)
    s["one_cycle/x"], s["one_cycle/y"] = xs, np.array([lf(float(x)) for x in xs], dtype=np.float64)
    for tag, updates0 in (("cold", 0), ("warm", 5000)):
        torch.manual_seed(21)
        net = nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5), nn.Conv2d(5, 7, 1, bias=True))
        with torch.no_grad():
            net[1].running_mean.normal_()
            net[1].running_var.uniform_(0.5, 1.5)
        ema = ModelEMA(net, updates=updates0)
        for k, v in net.state_dict().items():
            s[f"ema_{tag}/init/{k}"] = v.detach().clone().numpy()
        g = torch.Generator().manual_seed(22)
        for step in range(3):
            with torch.no_grad():
                for k, v in net.state_dict().items():
                    if v.dtype.is_floating_point:
                        v.add_(torch.randn(v.shape, generator=g) * 0.05)
                    else:
                        v.add_(1)
            for k, v in net.state_dict().items():
                s[f"ema_{tag}/model{step}/{k}"] = v.detach().clone().numpy()
            ema.update(net)
        for k, v in ema.ema.state_dict().items():
            s[f"ema_{tag}/final/{k}"] = v.detach().clone().numpy()
        s[f"ema_{tag}/updates"] = np.int64(ema.updates)
    np.savez_compressed(os.path.join(OUT, "trainutils.npz"), **s)


if __name__ == "__main__":
    which = sys.argv[1:] or ["calib", "modules", "net", "train", "nms", "metrics", "trainutils"]
    if "calib" in which:
        gen_calibration()
    if "modules" in which:
        gen_modules()
    m = None
    if "net" in which or "train" in which:
        m = gen_net() if "net" in which else build_ref_model()
    if "train" in which:
        gen_train(m)
    if "nms" in which:
        gen_nms()
    if "metrics" in which:
        gen_metrics()
    if "trainutils" in which:
        gen_trainutils()
