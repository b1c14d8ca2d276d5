"""CPU ORACLE for the DeSeNet CNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker / the timed CPU baseline.  The product path (desenet_amd/) never imports it and fails loudly without
its HIP library.

What it is: a functional restatement, on stock PyTorch-CPU fp32 NCHW ops, of the reference's forward graph
(the reference itself is pure-Python PyTorch, so "the reference's algorithm" *is* this op sequence):

    parse_arch              core/models/yolo.py:443-499   (parse_model: depth/width gains, C3 repeat insertion)
    conv_bn_act             core/models/common.py:42-56   (Conv: BN skipped when H*W == 1 -- quirk Q1;
                                                           forward_fuse when folded)
    bottleneck / c3 / spp   core/models/common.py:101-111, 133-145, 172-185
    focus / concat          core/models/common.py:618-627, 686-693
    rfb2 / pyramid / ffm    core/models/common.py:504-545, 588-615, 222-242
    seg_mask_psp            core/models/yolo.py:156-197
    detect                  core/models/yolo.py:238-282   (grid[...,0]=x, [...,1]=y; anchors in pixels for decode)
    forward                 core/models/yolo.py:344-356   (_forward_once routing; returns (det, seg))
    fold_bn                 core/utils/torch_utils.py:196-216 + core/models/yolo.py:409-417 (only Conv instances)
    BN hyper-parameters     core/utils/torch_utils.py:160-168  (eps 1e-3, momentum 0.03)

Parity is PINNED: tests/test_oracle_golden.py checks this module against the fixtures in tests/golden/, which
tools/gen_golden.py produced by importing and running the real reference in the build container.
Gradients come from torch.autograd over these same ops (state_dict tensors with requires_grad=True).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F

BN_EPS = 1e-3       # torch_utils.py:164
BN_MOMENTUM = 0.03  # torch_utils.py:165


# --------------------------------------------------------------------------------------------------
# architecture: yaml -> flat layer list                                           (yolo.py:443-499)
# --------------------------------------------------------------------------------------------------
@dataclass
class Layer:
    i: int
    f: object            # from: int or list
    kind: str
    args: list
    c_out: int
    n: int = 1           # repeats inside C3
    extra: dict = field(default_factory=dict)


def make_divisible(x, d):  # general.py:411-413
    return math.ceil(x / d) * d


def parse_arch(cfg: dict, ch: int = 3):
    gd, gw = cfg["depth_multiple"], cfg["width_multiple"]
    anchors = cfg["anchors"]
    na = len(anchors[0]) // 2
    no = na * (cfg["de_nc"] + 5)
    chans: List[int] = [ch]
    layers: List[Layer] = []
    save = []
    c2 = ch
    for i, (f, n, kind, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = [cfg[a] if isinstance(a, str) and a in cfg else a for a in args]
        args = [None if a == "None" else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        reps = 1
        if kind in ("Conv", "Focus", "SPP", "C3"):
            c1, c2 = chans[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if kind == "C3":
                reps, n = n, 1
        elif kind == "Concat":
            c2 = sum(chans[x] for x in f)
        elif kind == "Detect":
            args = [args[0], args[1], [chans[x] for x in f]]
        elif kind == "SegMaskPSP":
            a1 = max(round(args[1] * gd), 1) if args[1] > 1 else args[1]
            args = [args[0], a1, make_divisible(args[2] * gw, 8), args[3], [chans[x] for x in f]]
            c2 = chans[f[0]] if isinstance(f, list) else chans[f]
        else:  # nn.Upsample
            c2 = chans[f]
        assert n == 1, "only C3 repeats in DeSeNet graphs"
        layers.append(Layer(i, f, kind, args, c2, reps))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chans = []
        chans.append(c2)
    return layers, sorted(save) + [24]   # yolo.py:305 appends the seg layer index


# --------------------------------------------------------------------------------------------------
# state_dict template (reference key names / shapes)
# --------------------------------------------------------------------------------------------------
def _conv_keys(sd, p, c1, c2, k, bn=True):
    sd[p + ".conv.weight"] = torch.zeros(c2, c1, k, k)
    if bn:
        _bn_keys(sd, p + ".bn", c2)


def _bn_keys(sd, p, c):
    sd[p + ".weight"] = torch.ones(c)
    sd[p + ".bias"] = torch.zeros(c)
    sd[p + ".running_mean"] = torch.zeros(c)
    sd[p + ".running_var"] = torch.ones(c)
    sd[p + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def make_state_dict(cfg: dict, ch: int = 3) -> "OrderedDict[str, torch.Tensor]":
    layers, _ = parse_arch(cfg, ch)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for L in layers:
        p = f"model.{L.i}"
        if L.kind == "Focus":
            c1, c2, k = L.args[0], L.args[1], L.args[2]
            _conv_keys(sd, p + ".conv", c1 * 4, c2, k)
        elif L.kind == "Conv":
            c1, c2, k = L.args[0], L.args[1], L.args[2]
            _conv_keys(sd, p, c1, c2, k)
        elif L.kind == "C3":
            c1, c2 = L.args[0], L.args[1]
            c_ = int(c2 * 0.5)
            _conv_keys(sd, p + ".cv1", c1, c_, 1)
            _conv_keys(sd, p + ".cv2", c1, c_, 1)
            _conv_keys(sd, p + ".cv3", 2 * c_, c2, 1)
            for j in range(L.n):
                _conv_keys(sd, f"{p}.m.{j}.cv1", c_, c_, 1)
                _conv_keys(sd, f"{p}.m.{j}.cv2", c_, c_, 3)
        elif L.kind == "SPP":
            c1, c2, ks = L.args[0], L.args[1], L.args[2]
            _conv_keys(sd, p + ".cv1", c1, c1 // 2, 1)
            _conv_keys(sd, p + ".cv2", c1 // 2 * (len(ks) + 1), c2, 1)
        elif L.kind == "SegMaskPSP":
            ncls, _, ch_, _, cin = L.args
            _conv_keys(sd, p + ".m8.0", cin[0], ch_, 1)
            _conv_keys(sd, p + ".m16.0", cin[1], ch_, 1)
            _conv_keys(sd, p + ".m32.0", cin[2], ch_, 1)
            r = p + ".out.0"
            inter = ch_ * 3 // 6
            _conv_keys(sd, r + ".branch0.0", ch_ * 3, inter, 1)
            _conv_keys(sd, r + ".branch0.1", inter, inter, 3)
            for b in ("branch1", "branch2"):
                sd[f"{r}.{b}.0.weight"] = torch.zeros(inter, inter, 3, 3)
                _bn_keys(sd, f"{r}.{b}.1", inter)
            _conv_keys(sd, r + ".branch3.0", ch_ * 3, inter, 1)
            _conv_keys(sd, r + ".ConvLinear", 4 * inter, ch_, 1)
            for j in (1, 2, 3, 4):
                _conv_keys(sd, f"{p}.out.1.conv{j}", ch_, ch_ // 4, 1)
            _conv_keys(sd, p + ".out.2.convblk", ch_ * 2, ch_, 3)
            sd[p + ".out.2.channel_attention.1.weight"] = torch.zeros(ch_, ch_, 1, 1)
            sd[p + ".out.2.channel_attention.3.weight"] = torch.zeros(ch_, ch_, 1, 1)
            sd[p + ".out.3.weight"] = torch.zeros(ncls, ch_, 1, 1)
            sd[p + ".out.3.bias"] = torch.zeros(ncls)
        elif L.kind == "Detect":
            nc, anchors, cin = L.args
            a = torch.tensor(anchors).float().view(len(anchors), -1, 2)
            stride = torch.tensor([8.0, 16.0, 32.0])[: len(anchors)]
            sd[p + ".anchors"] = a / stride.view(-1, 1, 1)          # grid units (yolo.py:316)
            sd[p + ".anchor_grid"] = a.clone().view(len(anchors), 1, -1, 1, 1, 2)  # pixels (yolo.py:251)
            na = a.shape[1]
            for j, c in enumerate(cin):
                sd[f"{p}.m.{j}.weight"] = torch.zeros((nc + 5) * na, c, 1, 1)
                sd[f"{p}.m.{j}.bias"] = torch.zeros((nc + 5) * na)
    return sd


# --------------------------------------------------------------------------------------------------
# functional blocks
# --------------------------------------------------------------------------------------------------
class Ctx:
    """Carries the state_dict and mode flags through the functional graph."""

    def __init__(self, sd: Dict[str, torch.Tensor], training=False, fused=False):
        self.sd, self.training, self.fused = sd, training, fused


def silu(x):
    return x * torch.sigmoid(x)


def batch_norm(cx: Ctx, x, p):
    sd = cx.sd
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        cx.training, BN_MOMENTUM, BN_EPS)


def conv_bn_act(cx: Ctx, x, p, k=1, s=1, act=True):
    """common.py:42-56.  Fused: act(conv(x)+b).  Un-fused: act(bn(conv(x))) unless the input map is 1x1."""
    sd = cx.sd
    pad = k // 2  # autopad, common.py:32-39
    if cx.fused:
        y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], s, pad)
    else:
        y = F.conv2d(x, sd[p + ".conv.weight"], None, s, pad)
        if x.shape[2] * x.shape[3] > 1:   # x[0][0].numel() > 1
            y = batch_norm(cx, y, p + ".bn")
    return silu(y) if act else y


def focus(cx, x, p, k):  # common.py:626
    y = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    return conv_bn_act(cx, y, p + ".conv", k, 1)


def bottleneck(cx, x, p, shortcut):  # common.py:101-111 (c1 == c2 always holds inside C3)
    y = conv_bn_act(cx, conv_bn_act(cx, x, p + ".cv1", 1), p + ".cv2", 3)
    return x + y if shortcut else y


def c3(cx, x, p, n, shortcut):  # common.py:133-145
    y = conv_bn_act(cx, x, p + ".cv1", 1)
    for j in range(n):
        y = bottleneck(cx, y, f"{p}.m.{j}", shortcut)
    return conv_bn_act(cx, torch.cat((y, conv_bn_act(cx, x, p + ".cv2", 1)), 1), p + ".cv3", 1)


def spp(cx, x, p, ks):  # common.py:172-185
    x = conv_bn_act(cx, x, p + ".cv1", 1)
    return conv_bn_act(cx, torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in ks], 1), p + ".cv2", 1)


def rfb2(cx, x, p):  # common.py:504-545, has_global=False, d=[2,3]
    sd = cx.sd
    x3 = conv_bn_act(cx, x, p + ".branch3.0", 1)
    x0 = conv_bn_act(cx, conv_bn_act(cx, x, p + ".branch0.0", 1), p + ".branch0.1", 3)
    x1 = silu(batch_norm(cx, F.conv2d(x0, sd[p + ".branch1.0.weight"], None, 1, 2, 2), p + ".branch1.1"))
    x2 = silu(batch_norm(cx, F.conv2d(x1, sd[p + ".branch2.0.weight"], None, 1, 3, 3), p + ".branch2.1"))
    return conv_bn_act(cx, torch.cat([x0, x1, x2, x3], 1), p + ".ConvLinear", 1)


def pyramid_pooling(cx, x, p, ks=(1, 2, 3, 6)):  # common.py:588-615, short_cut=True
    h, w = x.shape[2:]
    feats = [x]
    for j, k in enumerate(ks, 1):
        f = conv_bn_act(cx, F.adaptive_avg_pool2d(x, k), f"{p}.conv{j}", 1)
        feats.append(F.interpolate(f, (h, w), mode="bilinear", align_corners=True))
    return torch.cat(feats, 1)


def ffm(cx, x, p):  # common.py:222-242, is_cat=False, k=3, reduction=1
    sd = cx.sd
    feat = conv_bn_act(cx, x, p + ".convblk", 3)
    a = F.adaptive_avg_pool2d(feat, 1)
    a = silu(F.conv2d(a, sd[p + ".channel_attention.1.weight"]))
    a = torch.sigmoid(F.conv2d(a, sd[p + ".channel_attention.3.weight"]))
    return feat * a + feat


def seg_mask_psp(cx, xs, p):  # yolo.py:156-197
    sd = cx.sd
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=True)
    f8 = conv_bn_act(cx, xs[0], p + ".m8.0", 1)
    f16 = up(conv_bn_act(cx, xs[1], p + ".m16.0", 1), 2)
    f32 = up(conv_bn_act(cx, xs[2], p + ".m32.0", 1), 4)
    y = rfb2(cx, torch.cat([f8, f16, f32], 1), p + ".out.0")
    y = pyramid_pooling(cx, y, p + ".out.1")
    y = ffm(cx, y, p + ".out.2")
    y = F.conv2d(y, sd[p + ".out.3.weight"], sd[p + ".out.3.bias"])
    return up(y, 8)


def make_grid(nx, ny):  # yolo.py:279-282
    yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
    return torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()


def detect(cx, xs, p, strides=(8.0, 16.0, 32.0)):  # yolo.py:255-277
    sd = cx.sd
    na = sd[p + ".anchors"].shape[1]
    raws, z = [], []
    for i, x in enumerate(xs):
        y = F.conv2d(x, sd[f"{p}.m.{i}.weight"], sd[f"{p}.m.{i}.bias"])
        bs, c, ny, nx = y.shape
        no = c // na
        y = y.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        raws.append(y)
        if not cx.training:
            s = y.sigmoid()
            xy = (s[..., 0:2] * 2.0 - 0.5 + make_grid(nx, ny)) * strides[i]
            wh = (s[..., 2:4] * 2) ** 2 * sd[p + ".anchor_grid"][i]
            z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(bs, -1, no))
    return raws if cx.training else (torch.cat(z, 1), raws)


def apply_layer(cx, L, out):
    """One top-level layer of the yaml graph (the body of yolo.py:352 `x = m(x)`): `out` is the routed input (a tensor, or the
    list of tensors for Concat / SegMaskPSP / Detect)."""
    p = f"model.{L.i}"
    if L.kind == "Focus":
        return focus(cx, out, p, L.args[2])
    if L.kind == "Conv":
        return conv_bn_act(cx, out, p, L.args[2], L.args[3] if len(L.args) > 3 else 1)
    if L.kind == "C3":
        return c3(cx, out, p, L.n, L.args[2] if len(L.args) > 2 else True)
    if L.kind == "SPP":
        return spp(cx, out, p, L.args[2])
    if L.kind == "nn.Upsample":
        return F.interpolate(out, scale_factor=float(L.args[1]), mode=L.args[2])
    if L.kind == "Concat":
        return torch.cat(out, 1)
    if L.kind == "SegMaskPSP":
        return seg_mask_psp(cx, out, p)
    if L.kind == "Detect":
        return detect(cx, list(out), p)
    raise NotImplementedError(L.kind)


def forward(cfg, sd, x, training=False, fused=False, keep=None):
    """yolo.py:344-356.  Returns (det_out, seg_out, saved) where det_out follows Detect's train/eval contract and
    `saved` maps layer index -> output for every index in `keep` (default: the reference's save list)."""
    layers, save = parse_arch(cfg, x.shape[1])
    keep = set(save if keep is None else keep)
    cx = Ctx(sd, training, fused)
    y: List[object] = []
    out = x
    for L in layers:
        if L.f != -1:
            out = y[L.f] if isinstance(L.f, int) else [out if j == -1 else y[j] for j in L.f]
        out = apply_layer(cx, L, out)
        y.append(out if L.i in keep or L.i in save else None)
    saved = {i: y[i] for i in keep if i < len(y) and torch.is_tensor(y[i])}
    return out, y[-2], saved


def forward_augment(cfg, sd, x, fused=False):
    """Test-time augmentation, yolo.py:331-342 + _descale_pred :358-374 + torch_utils.scale_img :262-272 (scales 1 / .83 / .67,
    left-right flip on the second, gs = 32).  The reference's own code path raises (it hands `_descale_pred` the (pred, raws)
    TUPLE that `_forward_once(...)[0]` is once the seg head exists, yolo.py:356); this restates the evident intent -- the
    upstream-YOLOv5 form on the decoded predictions -- and returns what the reference's return statement would: (cat, None)."""
    import math
    img_size = x.shape[-2:]
    ys = []
    for si, fi in zip([1, 0.83, 0.67], [None, 3, None]):
        xi = x.flip(fi) if fi else x
        if si != 1:
            h, w = xi.shape[2:]
            s = (int(h * si), int(w * si))
            xi = F.interpolate(xi, size=s, mode="bilinear", align_corners=False)
            hh, ww = [math.ceil(v * si / 32) * 32 for v in (h, w)]
            xi = F.pad(xi, [0, ww - s[1], 0, hh - s[0]], value=0.447)
        (pred, _), _, _ = forward(cfg, sd, xi, fused=fused)
        p = pred.clone()
        p[..., :4] /= si
        if fi == 3:
            p[..., 0] = img_size[1] - p[..., 0]
        ys.append(p)
    return torch.cat(ys, 1), None


# --------------------------------------------------------------------------------------------------
# BN folding                                              (torch_utils.py:196-216, yolo.py:409-417)
# --------------------------------------------------------------------------------------------------
def fold_bn(sd):
    """Return a new state_dict in which every `<p>.conv.weight` + `<p>.bn.*` group (= a Conv instance) is folded to
    `<p>.conv.weight` + `<p>.conv.bias`.  Raw Conv2d+BatchNorm2d pairs (RFB2.branch1/2) are NOT folded (quirk Q3)."""
    out = OrderedDict()
    for k, v in sd.items():
        if ".bn." in k:
            continue
        if k.endswith(".conv.weight") and (k[: -len("conv.weight")] + "bn.weight") in sd:
            p = k[: -len("conv.weight")]
            g, b = sd[p + "bn.weight"], sd[p + "bn.bias"]
            m, var = sd[p + "bn.running_mean"], sd[p + "bn.running_var"]
            w_bn = torch.diag(g.div(torch.sqrt(BN_EPS + var)))
            out[k] = torch.mm(w_bn, v.reshape(v.shape[0], -1)).view(v.shape)
            out[p + "conv.bias"] = b - g.mul(m).div(torch.sqrt(var + BN_EPS))
        else:
            out[k] = v
    return out
