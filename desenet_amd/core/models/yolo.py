"""Drop-in mirror of the reference's `core.models.yolo` (DeSeNet-s path) on libdesenet_hip.so.

    SegMaskPSP   core/models/yolo.py:156-197        Detect       :238-282
    Model        :285-440  (forward :326-356, _initialize_biases :388-396, fuse :409-417)
    parse_model  :443-499

Differences that are deliberate and documented in INTEGRATION.md:
  * strides are derived from the graph (product of layer strides) instead of a 256x256 probe forward on the CPU
    (yolo.py:313-315) -- there is no CPU compute path; the values are identical ([8, 16, 32]);
  * module names in the yaml are resolved from an explicit table, not `eval` (yolo.py:451);
  * the whole network is ONE autograd node: concat buffers are planned so producers write straight into them, and
    backward walks the layer list in reverse, accumulating fan-in gradients in place.
"""
from __future__ import annotations

import logging
import math
import os
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from ... import hip_ops as ops
from ...conv_impl import conv_block_bwd, conv_block_fwd, packed_fwd
from ...hip_ops import ACT_NONE
from ...runtime import run_module
from ..utils.autoanchor import check_anchor_order
from ..utils.general import make_divisible
from ..utils.torch_utils import fuse_conv_and_bn, initialize_weights, model_info
from .common import (C3, FFM, RFB2, SPP, Bottleneck, Concat, Conv, Focus, HipModule, PyramidPooling, Upsample,
                     _as_input)

LOGGER = logging.getLogger(__name__)


class _Bilinear(nn.Upsample):
    """nn.Upsample(scale_factor=s, mode='bilinear', align_corners=True) placeholder inside the seg head's Sequentials
    (keeps the reference's child indices, hence its state_dict keys); the work is done by SegMaskPSP.fwd."""


class SegMaskPSP(HipModule):
    """PSP-style segmentation head: 3-level fusion -> RFB2 -> PyramidPooling -> FFM -> 1x1 classifier -> x8 bilinear."""

    def __init__(self, n_segcls=19, n=1, c_hid=256, shortcut=False, ch=()):
        super().__init__()
        self.c_in8, self.c_in16, self.c_in32 = ch[0], ch[1], ch[2]
        self.c_out = n_segcls
        self.m8 = nn.Sequential(Conv(self.c_in8, c_hid, k=1))
        self.m16 = nn.Sequential(Conv(self.c_in16, c_hid, k=1),
                                 _Bilinear(scale_factor=2, mode="bilinear", align_corners=True))
        self.m32 = nn.Sequential(Conv(self.c_in32, c_hid, k=1),
                                 _Bilinear(scale_factor=4, mode="bilinear", align_corners=True))
        self.out = nn.Sequential(
            RFB2(c_hid * 3, c_hid, d=[2, 3], map_reduce=6),
            PyramidPooling(c_hid, k=[1, 2, 3, 6], short_cut=True),
            FFM(c_hid * 2, c_hid, k=3, is_cat=False),
            nn.Conv2d(c_hid, self.c_out, kernel_size=1, padding=0),
            _Bilinear(scale_factor=8, mode="bilinear", align_corners=True),
        )

    def fwd(self, xs, tape=None, out=None):
        x8, x16, x32 = (_as_input(t) for t in xs)
        c = self.m8[0].conv.out_channels
        n, _, h, w = x8.shape
        dt, dev = x8.dtype, x8.device
        cat = ops.new_act(n, 3 * c, h, w, dt, dev)                      # [m8 | up2(m16) | up4(m32)]
        self.m8[0].fwd(x8, tape, cat[:, :c])
        f16 = self.m16[0].fwd(x16, tape)
        ops.bilinear_ac(f16, cat[:, c:2 * c])
        f32 = self.m32[0].fwd(x32, tape)
        ops.bilinear_ac(f32, cat[:, 2 * c:])
        pp = ops.new_act(n, 2 * c, h, w, dt, dev)                       # [rfb2 out | pyramid feats]
        self.out[0].fwd(cat, tape, pp[:, :c])
        self.out[1].fwd(pp[:, :c], tape, pp)
        y = self.out[2].fwd(pp, tape)
        logits = conv_block_fwd(y, self.out[3], None, ACT_NONE, self.training, tape)
        if tape is not None and self.__dict__.get("_dsn_lowres_out") and self.c_out == 2:
            # fused-loss mode (desenet_amd.graph.GraphedTrainStep): hand out the 1/8-resolution logits; the x8 bilinear of
            # yolo.py:183 happens inside the loss kernel and the gradient comes back at this resolution
            logits._dsn_seg_upsample = (8 * h, 8 * w)
            tape.push((f16.shape, f32.shape, logits.shape, dt, True))
            return logits
        seg = torch.empty((n, self.c_out, 8 * h, 8 * w), dtype=torch.float32, device=dev)   # caller-facing NCHW fp32
        ops.bilinear_ac(logits, seg, out_nchw=True)
        if tape is not None:
            tape.push((f16.shape, f32.shape, logits.shape, dt, False))
        return seg

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        s16, s32, sl, dt, lowres = tape.pop()
        dev = dy.device
        c = self.m8[0].conv.out_channels
        # rows padded with zeros to the 16-byte vector (2 -> 4 / 8 classes): the classifier's weight and input gradients take the
        # vector paths (and the weight gradient joins the grouped launch) instead of the scalar-load ones
        vec = 4 if dt == torch.float32 else 8
        if lowres:
            if tuple(dy.shape) != tuple(sl) or not getattr(dy, "_dsn_zero_padded", False):
                raise RuntimeError("SegMaskPSP handed out low-resolution logits (fused seg loss): its backward expects the "
                                   "gradient SegmentationLosses.forward_backward returned for them")
            dlog = dy
        else:
            g = dy if (dy.dtype == torch.float32 and dy.is_contiguous()) else dy.float().contiguous()
            dlog = ops.bilinear_ac_bwd(g, ops.new_act(*sl, dt, dev, zero=True, ldc_align=vec), dy_nchw=True)
        dlog._dsn_zero_padded = True
        d_y = conv_block_bwd(tape, dlog)
        dpp = self.out[2].bwd(tape, d_y)
        drfb = self.out[1].bwd(tape, dpp)
        dcat = self.out[0].bwd(tape, drfb)
        dxs = list(dx) if dx is not None else [None, None, None]
        accs = list(acc) if isinstance(acc, (list, tuple)) else [acc] * 3
        df32 = ops.bilinear_ac_bwd(dcat[:, 2 * c:], ops.new_act(*s32, dt, dev))
        dxs[2] = self.m32[0].bwd(tape, df32, dxs[2], accs[2], need_dx)
        df16 = ops.bilinear_ac_bwd(dcat[:, c:2 * c], ops.new_act(*s16, dt, dev))
        dxs[1] = self.m16[0].bwd(tape, df16, dxs[1], accs[1], need_dx)
        dxs[0] = self.m8[0].bwd(tape, dcat[:, :c], dxs[0], accs[0], need_dx)
        return dxs


# the three head convolutions + permute of a training forward as one launch (DSN_DET_FUSED=0: conv per level + one permute launch)
_DET_FUSED = os.environ.get("DSN_DET_FUSED", "1") != "0"


def _conv_is_not_plain_1x1(m) -> bool:
    return not (isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
                and m.dilation == (1, 1) and m.groups == 1)


class Detect(HipModule):
    stride = None
    onnx_dynamic = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def _strides_host(self):
        """Detect.stride as host floats, cached (a .tolist() per forward would be a device sync when it lives on the GPU)."""
        key = (self.stride.data_ptr(), self.stride._version) if torch.is_tensor(self.stride) else None
        hit = self.__dict__.get("_dsn_strides")
        if hit is None or hit[0] != key:
            vals = [float(v) for v in (self.stride.tolist() if torch.is_tensor(self.stride) else self.stride)]
            hit = self.__dict__["_dsn_strides"] = (key, vals)
        return hit[1]

    def fwd(self, xs, tape=None, out=None):
        xs = [_as_input(t) for t in xs]
        n = xs[0].shape[0]
        dev = xs[0].device
        total = sum(self.na * t.shape[2] * t.shape[3] for t in xs)
        pred = None if self.training else torch.empty((n, total, self.no), dtype=torch.float32, device=dev)
        raws, row = [], 0
        if self.training and tape is not None and _DET_FUSED and self._fwd_fused(xs, tape, raws):
            tape.push([(tuple(x.shape), x.dtype) for x in xs])
            return raws
        anchors_px = self.anchor_grid.view(self.nl, self.na, 2).float().contiguous()
        ts, rows = [], []
        for i, x in enumerate(xs):               # the three head convs, then ONE decode launch for all levels
            t = conv_block_fwd(x, self.m[i], None, ACT_NONE, self.training, tape)
            _, _, ny, nx = t.shape
            ts.append(t)
            raws.append(torch.empty((n, self.na, ny, nx, self.no), dtype=torch.float32, device=dev))
            rows.append(row)
            row += self.na * ny * nx
        ops.detect_decode_multi(ts, raws, pred, rows, self.na, self.no, [float(v) for v in self._strides_host()], anchors_px)
        if tape is not None:
            tape.push([(tuple(x.shape), x.dtype) for x in xs])
        return raws if self.training else (pred, raws)

    def _fwd_fused(self, xs, tape, raws) -> bool:
        """Training forward of all levels as ONE launch (ops.detect_head_fwd: 1x1 conv + bias + permute); leaves on the tape the
        records conv_block_fwd would have pushed, so bwd() is unchanged.  False (nothing done) when the kernel does not take it."""
        dt = xs[0].dtype
        if any(x.dtype != dt for x in xs) or any(_conv_is_not_plain_1x1(m) for m in self.m) \
                or not ops.detect_head_fwd_supported(dt, self.na, self.no, [x.shape[1] for x in xs]):
            return False
        xm = list(xs)
        if not all(ops._vec16(x) for x in xm):
            return False
        n = xm[0].shape[0]
        packed = [packed_fwd(m, dt, None, None) for m in self.m]
        for x in xm:
            raws.append(torch.empty((n, self.na, x.shape[2], x.shape[3], self.no), dtype=torch.float32, device=x.device))
        ops.detect_head_fwd(xm, [w for w, _ in packed], [b for _, b in packed], raws, self.na, self.no)
        for i, m in enumerate(self.m):
            tape.push(dict(conv=m, bn=None, act=ACT_NONE, x=xm[i], x_in=xs[i], ci_pad=None, geom=(1, 1, 0, 1), plain=True, y=None))
        return True

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        metas = tape.pop()
        draws = dy if self.training else dy[1]
        dxs = list(dx) if dx is not None else [None] * self.nl
        accs = list(acc) if isinstance(acc, (list, tuple)) else [acc] * self.nl
        # rows padded to a multiple of the 16-byte vector (33 -> 40 channels), padding written as zeros: the heads' weight and
        # input gradients then take the vector paths instead of the scalar-load ones.  ONE launch un-permutes every level and
        # sums the bias gradients (when the biases have gradient slots to add into), one more finalizes them.
        dtls = []
        for i in range(self.nl):
            (n, _, ny, nx), dt = metas[i]
            vec = 4 if dt == torch.float32 else 8
            dtls.append(ops.new_act(n, self.na * self.no, ny, nx, dt, draws[i].device, ldc_align=vec))
        slots = [m.bias.grad if (m.bias is not None and m.bias.requires_grad and m.bias.grad is not None
                                 and m.bias.grad.dtype == torch.float32 and m.bias.grad.is_contiguous()) else None for m in self.m]
        fused_bias = all(s is not None for s in slots)
        ops.detect_raw_bwd_multi(list(draws), dtls, self.na, self.no, slots if fused_bias else None)
        for i in reversed(range(self.nl)):
            dtls[i]._dsn_zero_padded = True
            dtls[i]._dsn_bias_done = fused_bias
            dxs[i] = conv_block_bwd(tape, dtls[i], dxs[i], accs[i], need_dx)
        return dxs


# name table replacing the reference's eval() of yaml strings (yolo.py:451)
MODULES = {"Conv": Conv, "Focus": Focus, "C3": C3, "SPP": SPP, "Bottleneck": Bottleneck, "Concat": Concat,
           "nn.Upsample": Upsample, "Upsample": Upsample, "SegMaskPSP": SegMaskPSP, "Detect": Detect}


def parse_model(d, ch):
    """yaml dict -> (nn.Sequential of layers with .i/.f/._type/.np attached, sorted save list); yolo.py:443-499."""
    LOGGER.info("\n%3s%18s%3s%10s  %-40s%-30s" % ("", "from", "n", "params", "module", "arguments"))
    anchors, de_nc, se_nc, gd, gw = d["anchors"], d["de_nc"], d["se_nc"], d["depth_multiple"], d["width_multiple"]
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (de_nc + 5)
    names = {"de_nc": de_nc, "se_nc": se_nc, "anchors": anchors, "None": None, "False": False, "True": True}
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, mname, args) in enumerate(d["backbone"] + d["head"]):
        if mname not in MODULES:
            raise NotImplementedError(f"module '{mname}' is outside the DeSeNet-s hot path (SURVEY.md 2, rows 20-22)")
        m = MODULES[mname]
        args = [names[a] if isinstance(a, str) and a in names else a for a in args]
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        if m in (Conv, Bottleneck, SPP, Focus, C3):
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if m is C3:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m is Detect:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        elif m is SegMaskPSP:
            args[1] = max(round(args[1] * gd), 1) if args[1] > 1 else args[1]
            args[2] = make_divisible(args[2] * gw, 8)
            args.append([ch[x] for x in f])
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*[m(*args) for _ in range(n)]) if n > 1 else m(*args)
        t = f"{m.__module__}.{m.__name__}".replace("desenet_amd.", "")
        np_ = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_._type, m_.np = i, f, t, np_
        LOGGER.info("%3s%18s%3s%10.0f  %-40s%-30s" % (i, f, n_, np_, t, args))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def _layer_stride(m) -> float:
    """Spatial reduction of one top-level layer (replaces the probe forward of yolo.py:313-315)."""
    if isinstance(m, Focus):
        return 2.0 * m.conv.conv.stride[0]
    if isinstance(m, Conv):
        return float(m.conv.stride[0])
    if isinstance(m, Upsample):
        return 1.0 / float(m.scale_factor)
    return 1.0


class Model(HipModule):
    def __init__(self, cfg="desenet_s.yaml", ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            import yaml
            cfg = Path(cfg)
            if not cfg.exists():
                cfg = Path(__file__).resolve().parents[2] / "cfg" / cfg.name
            self.yaml_file = cfg.name
            with open(cfg, "r", encoding="utf-8") as f:
                self.yaml = yaml.safe_load(f)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["de_nc"]:
            LOGGER.info(f"Overriding model.yaml de_nc={self.yaml['de_nc']} with de_nc={nc}")
            self.yaml["de_nc"] = nc
        if anchors:
            LOGGER.info(f"Overriding model.yaml anchors with anchors={anchors}")
            self.yaml["anchors"] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.seg_index = next((m.i for m in self.model if isinstance(m, SegMaskPSP)), len(self.model) - 2)
        self.save.append(self.seg_index)   # reference hard-codes 24 (yolo.py:305)
        self.de_names = [str(i) for i in range(self.yaml["de_nc"])]
        self.se_names = [str(i) for i in range(self.yaml["se_nc"])]
        self.inplace = self.yaml.get("inplace", True)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            m.stride = torch.tensor(self._infer_strides(m))
            m.anchors /= m.stride.view(-1, 1, 1)
            check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)
        self._plan_concats()
        self.info()
        LOGGER.info("")

    # ---- adopting an unpickled tree (attempt_load / train.py:125-128 with desenet_amd.shim installed) ------------------
    def __setstate__(self, state):
        super().__setstate__(state)
        self._adopt()

    def _adopt(self):
        """A Model that was unpickled (no __init__): give stock torch classes their mirrored type and rebuild this package's
        plans.  Reference pickles hold `nn.Upsample` layers and plain `nn.Sequential` Conv2d+BN+SiLU triples inside RFB2."""
        from .common import _ConvBnAct
        for m in self.model:
            if type(m) is nn.Upsample:
                m.__class__ = Upsample
        for m in self.modules():
            if isinstance(m, RFB2):
                for b in (m.branch1, m.branch2):
                    if type(b) is nn.Sequential:
                        b.__class__ = _ConvBnAct
        self.__dict__.pop("_dsn_bank", None)
        if "seg_index" not in self.__dict__:
            self.seg_index = next((m.i for m in self.model if isinstance(m, SegMaskPSP)), len(self.model) - 2)
        self._plan_concats()

    # ---- construction helpers -------------------------------------------------------------------------------------
    def _infer_strides(self, det):
        scale = []
        for m in self.model:
            src = m.f if isinstance(m.f, int) else m.f[0]
            prev = 1.0 if m.i == 0 else scale[src if src >= 0 else m.i + src]
            scale.append(prev * _layer_stride(m))
        return [scale[j] for j in det.f]

    def _initialize_biases(self, cf=None):
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def _plan_concats(self):
        """For every Concat layer: which producer layer writes which channel range of its buffer."""
        self._cat_slot = {}
        chans = {}
        for m in self.model:
            if isinstance(m, Concat):
                c0 = 0
                srcs = [m.i + j if j < 0 else j for j in m.f]
                sizes = [chans[s] for s in srcs]
                ok = all(s not in self._cat_slot and not isinstance(self.model[s], (Concat, Detect, SegMaskPSP))
                         for s in srcs) and len(set(srcs)) == len(srcs)
                if ok:
                    for s, c in zip(srcs, sizes):
                        self._cat_slot[s] = (m.i, c0, c, sum(sizes))
                        c0 += c
                chans[m.i] = sum(sizes)
            else:
                chans[m.i] = self._out_channels(m, chans)
        self._plan_deferred()

    def _plan_deferred(self):
        """The consumer graph of the top-level layers and, from it, which layer's backward completes the gradient of its input
        layer (BatchNorm backward sums in that launch's epilogue)."""
        consumers = {m.i: [] for m in self.model}
        for m in self.model:
            srcs = [m.i - 1] if m.f == -1 else ([m.f] if isinstance(m.f, int) else [m.i + j if j < 0 else j for j in m.f])
            for s_ in srcs:
                if 0 <= s_ < m.i:
                    consumers[s_].append(m)

        # layers whose backward writes the LAST contribution to the gradient of their (single) input layer -- the consumer with
        # the lowest index: the backward pass walks the layers downwards and every other consumer has already added its share to
        # the buffer this one accumulates into -- so that gradient is complete when they write it and its BatchNorm backward sums can
        # ride in that launch (conv_impl.conv_block_bwd: fuse_up).  The input layer must itself be a convolution block: a
        # Concat's sources have consumers of their own.
        layers = list(self.model)
        self._final_consumer = set()
        for m in layers:
            if isinstance(m, (Conv, C3, SPP)) and isinstance(m.f, int) and m.i > 0:
                src = m.i - 1 if m.f == -1 else m.f
                if min(c.i for c in consumers[src]) == m.i and isinstance(layers[src], (Conv, Focus, C3, SPP)):
                    self._final_consumer.add(m.i)

    @staticmethod
    def _out_channels(m, chans):
        if isinstance(m, Focus):
            return m.conv.conv.out_channels
        if isinstance(m, Conv):
            return m.conv.out_channels
        if isinstance(m, C3):
            return m.cv3.conv.out_channels
        if isinstance(m, SPP):
            return m.cv2.conv.out_channels
        if isinstance(m, Upsample):
            return chans[m.i - 1 if m.f == -1 else m.f]
        return 0

    # ---- forward / backward ---------------------------------------------------------------------------------------
    def forward(self, x, augment=False, profile=False, visualize=False):
        if profile or visualize:
            raise NotImplementedError("profile / visualize are outside the hot path (SURVEY.md 2 row 5)")
        if augment:
            return self._forward_augment(x)
        return run_module(self, x)

    def _forward_augment(self, x):
        """Test-time augmentation (yolo.py:331-342,358-374): scales 1 / 0.83 / 0.67, left-right flip on the second, de-scaled and
        de-flipped predictions concatenated along the box axis; returns `(cat, None)` as the reference's return statement does.
        NOTE: the reference's own path cannot run -- with the seg head `_forward_once(xi)[0]` is the TUPLE (pred, raws)
        (yolo.py:356), and `_descale_pred` indexes it with `p[..., :4]` (TypeError) -- so this follows the evident intent
        (upstream YOLOv5: the decoded predictions): oracle/desenet_ref.forward_augment restates the same arithmetic."""
        from ..utils.torch_utils import scale_img
        if self.training:
            raise RuntimeError("augmented inference needs model.eval()")
        img_size = x.shape[-2:]
        gs = int(float(self.stride.max()))
        y = []
        for si, fi in zip([1, 0.83, 0.67], [None, 3, None]):
            xi = scale_img(x.flip(fi) if fi else x, si, gs=gs)
            (pred, _), _ = run_module(self, xi)
            p = pred.clone()
            p[..., :4] /= si                                   # de-scale
            if fi == 2:
                p[..., 1] = img_size[0] - p[..., 1]            # de-flip ud
            elif fi == 3:
                p[..., 0] = img_size[1] - p[..., 0]            # de-flip lr
            y.append(p)
        return torch.cat(y, 1), None

    # ---- per-step, whole-model preparation (training): one pack launch, one BN-counter increment --------------------------
    def _prepare_training_step(self, dtype, device):
        from ...conv_impl import _cache, _ver
        ops.bn_arena_begin(device)        # one memset clears the BatchNorm accumulator slots of this step
        convs = [m for m in self.modules() if isinstance(m, nn.Conv2d)]
        bank = self.__dict__.get("_dsn_bank")
        if bank is None or not bank.valid_for(dtype) or len(bank.convs) != len(convs):
            vec = 4 if dtype == torch.float32 else 8
            focus_convs = {m.conv.conv for m in self.modules() if isinstance(m, Focus)}
            pads = [((c.in_channels + vec - 1) // vec * vec) if c in focus_convs else c.in_channels for c in convs]
            det_convs = {c for m in self.modules() if isinstance(m, Detect) for c in m.m}
            det_convs |= {m.out[3] for m in self.modules() if isinstance(m, SegMaskPSP)}    # seg classifier: same row padding
            co_pads = [((c.out_channels + vec - 1) // vec * vec) if c in det_convs else c.out_channels for c in convs]
            # C3's cv2 and cv1 read the same input: packed back to back they also run as ONE convolution (conv_impl.pair_block_*)
            pairs = [(m.cv2.conv, m.cv1.conv) for m in self.modules() if isinstance(m, C3) and not m.cv1.fused]
            pairs += [(m.branch3[0].conv, m.branch0[0].conv) for m in self.modules()
                      if isinstance(m, RFB2) and not m.branch3[0].fused]      # RFB2's two 1x1 convs of x (common.py:515-523)
            bank = self.__dict__["_dsn_bank"] = ops.WeightBank(convs, pads, dtype, device, co_pads=co_pads, pairs=pairs)
            self.__dict__["_dsn_bank_pads"] = bank.ci_pads
            self.__dict__["_dsn_bank_version"] = None
        convs = bank.convs                    # the bank's own order (pairs adjacent)
        version = tuple(c.weight._version for c in convs)
        # (while a graph is being captured the re-pack is ALWAYS recorded: every replay follows an optimizer step)
        if version != self.__dict__.get("_dsn_bank_version") or torch.cuda.is_current_stream_capturing():
            bank.pack()
            self.__dict__["_dsn_bank_version"] = version
            for c, cp, f, d, s2 in zip(convs, self.__dict__["_dsn_bank_pads"], bank.fwd, bank.dgrad, bank.dgrad_s2):
                bias = c.bias.detach() if c.bias is not None else None
                cache = _cache(c)
                cache[("fwd", dtype, cp, False)] = (_ver(c.weight, c.bias), f, bias)
                if cp == c.in_channels:
                    cache[("fwd", dtype, None, False)] = cache[("fwd", dtype, cp, False)]
                if s2 is not None:
                    cache[("dgrad_s2", dtype)] = (_ver(c.weight), s2)
                if d is not None:          # (None: second half of a pair, its columns live in the pair's merged matrix)
                    cop = d.shape[-1]      # > out_channels for the row-padded Detect heads and the merged pairs
                    cache[("dgrad", dtype) if cop == c.out_channels else ("dgrad", dtype, cop)] = (_ver(c.weight), d)
            for a, (b, wf) in bank.pair_fwd.items():
                _cache(a)[("pair", dtype)] = (_ver(a.weight, b.weight), b, wf, bank.pair_dgrad[a])
        # BatchNorm `num_batches_tracked`: every counter is a view of one int64 vector -> a single add per step
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None
               and not getattr(m, "_dsn_never_runs", False)]
        flat = self.__dict__.get("_dsn_bn_counters")
        if (flat is None or flat.device != device or flat.numel() != len(bns)
                or any(b.num_batches_tracked.data_ptr() != flat[i].data_ptr() for i, b in enumerate(bns))):
            flat = torch.stack([b.num_batches_tracked.detach().to(device) for b in bns]) if bns else None
            for i, b in enumerate(bns):
                b._buffers["num_batches_tracked"] = flat[i]
                b.__dict__["_dsn_shared_counter"] = True
            self.__dict__["_dsn_bn_counters"] = flat
        if flat is not None:
            ops.add_i64_(flat, 1)

    def fwd(self, x, tape=None, out=None):
        if self.training:
            for pp in (m for m in self.modules() if isinstance(m, PyramidPooling)):
                for pool, conv in pp._branches():      # quirk Q1: a BN behind a 1x1 pool never runs un-fused
                    sz = pool.output_size if isinstance(pool.output_size, int) else pool.output_size[0]
                    if sz == 1 and hasattr(conv, "bn"):
                        conv.bn.__dict__["_dsn_never_runs"] = True
            self._prepare_training_step(self._dtype_of(x), self._device_of(x))
        y = []
        cats = {}
        for m in self.model:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            dst = None
            slot = self._cat_slot.get(m.i)
            if slot is not None:
                cat_i, c0, c, ctot = slot
                n, h, w = self._out_nhw(m, x)
                buf = cats.get(cat_i)
                if buf is None:
                    buf = cats[cat_i] = ops.new_act(n, ctot, h, w, self._dtype_of(x), self._device_of(x))
                dst = buf[:, c0:c0 + c]
            x = m.fwd(x, tape, dst) if dst is not None else m.fwd(x, tape)
            y.append(x if m.i in self.save else None)
        if tape is not None:
            tape.finalize_forward()       # ONE launch: saved statistics + running averages of every BatchNorm of this pass
        return x, y[self.seg_index]

    @staticmethod
    def _dtype_of(x):
        from ...runtime import compute_dtype
        return compute_dtype()

    @staticmethod
    def _device_of(x):
        return (x[0] if isinstance(x, (list, tuple)) else x).device

    @staticmethod
    def _out_nhw(m, x):
        n, _, h, w = x.shape
        if isinstance(m, Upsample):
            return n, 2 * h, 2 * w
        if isinstance(m, Conv):
            k, s, p = m.conv.kernel_size[0], m.conv.stride[0], m.conv.padding[0]
            ho, wo = ops.conv_out_hw(h, w, k, s, p, 1)
            return n, ho, wo
        return n, h, w

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=False):
        if need_dx:
            raise NotImplementedError("gradient w.r.t. the input image is not part of the training path")
        grads = self.bwd_begin(dy)
        self.bwd_layers(tape, grads, len(self.model) - 1, 0)
        return None

    def bwd_begin(self, dy):
        """Gradient table of a backward pass: {top-level layer index: gradient of that layer's output}."""
        d_det, d_seg = dy
        return {len(self.model) - 1: d_det, self.seg_index: d_seg}

    def bwd_layers(self, tape, grads, hi, lo):
        """Walk the top-level layers hi, hi-1, ..., lo (inclusive) of a backward pass begun with bwd_begin(); the pass may be
        continued later with a lower range (desenet_amd.graph cuts it in two around the first gradient all-reduce)."""
        layers = list(self.model)
        for i in range(hi, lo - 1, -1):
            m = layers[i]
            g = grads.pop(m.i, None)
            if g is None:
                raise RuntimeError(f"layer {m.i} ({m._type}) received no gradient")
            if m.i == 0:
                m.bwd(tape, g, need_dx=False)
                break
            srcs = [m.i - 1] if m.f == -1 else ([m.f] if isinstance(m.f, int) else
                                               [m.i + j if j < 0 else j for j in m.f])
            if isinstance(m.f, int):
                have = grads.get(srcs[0])
                if m.i in self._final_consumer:
                    grads[srcs[0]] = m.bwd(tape, g, have, have is not None, fuse_up=True)
                else:
                    grads[srcs[0]] = m.bwd(tape, g, have, have is not None)
            else:
                have = [grads.get(s) for s in srcs]
                outs = m.bwd(tape, g, have, [h is not None for h in have])
                for s, o in zip(srcs, outs):
                    grads[s] = o

    # ---- reference API ----------------------------------------------------------------------------------------------
    def fuse(self):
        """Fold BatchNorm into every `Conv` (and only those: RFB2's raw Conv2d+BN pairs stay, quirk Q3); yolo.py:409-417."""
        LOGGER.info("Fusing layers... ")
        for m in self.model.modules():
            if isinstance(m, Conv) and hasattr(m, "bn"):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, "bn")
        self.info()
        return self

    def info(self, verbose=False, img_size=640):
        model_info(self, verbose, img_size)
