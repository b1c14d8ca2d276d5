"""Model-side helpers that define numerics the kernels must honour (reference core/utils/torch_utils.py):
initialize_weights (:160-168: BN eps 1e-3, momentum 0.03), fuse_conv_and_bn (:196-216), model_info (:219-240, without the
thop FLOP probe), intersect_dicts (:151-157), de_parallel (:47-51), copy_attr (:275-281), ModelEMA (:304-346).  Host-side,
one-off parameter algebra (plain torch) -- except ModelEMA.update, which runs every optimizer step and is ONE HIP launch."""
from __future__ import annotations

import ctypes as C
import logging
import math
from copy import deepcopy

import torch
import torch.nn as nn

LOGGER = logging.getLogger(__name__)


def initialize_weights(model):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03
        elif isinstance(m, (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6)):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """W' = diag(g / sqrt(var + eps)) W,  b' = b - g * mean / sqrt(var + eps) (+ scaled conv bias)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
        fused.weight.copy_(torch.mm(torch.diag(scale), conv.weight.view(conv.out_channels, -1)).view(fused.weight.shape))
        b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
        b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fused.bias.copy_(torch.mm(torch.diag(scale), b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    """torch_utils.py:262-272: img (bs, 3, y, x) scaled by `ratio` (bilinear, align_corners=False) and padded with 0.447 to a
    multiple of gs.  On the GPU the resize is the library's kernel (dsn_resize_bilinear_nchw); the pad is a copy."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    if img.is_cuda:
        from ... import hip_ops as ops
        img = ops.resize_bilinear_nchw(img, s, align_corners=False)
    else:
        img = torch.nn.functional.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = [math.ceil(x * ratio / gs) * gs for x in (h, w)]
    return torch.nn.functional.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def intersect_dicts(da, db, exclude=()):
    return {k: v for k, v in da.items() if k in db and not any(x in k for x in exclude) and v.shape == db[k].shape}


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def de_parallel(model):
    return model.module if is_parallel(model) else model


def model_info(model, verbose=False, img_size=640):
    n_p = sum(x.numel() for x in model.parameters())
    n_g = sum(x.numel() for x in model.parameters() if x.requires_grad)
    if verbose:
        for i, (name, p) in enumerate(model.named_parameters()):
            LOGGER.info("%5g %40s %9s %12g %20s" % (i, name.replace("module_list.", ""), p.requires_grad, p.numel(),
                                                     list(p.shape)))
    LOGGER.info(f"Model Summary: {len(list(model.modules()))} layers, {n_p} parameters, {n_g} gradients")


def copy_attr(a, b, include=(), exclude=()):
    """Copy attributes of b that do not start with '_' onto a (white list `include`, black list `exclude`)."""
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith("_") or k in exclude:
            continue
        setattr(a, k, v)


class ModelEMA:
    """Exponential moving average of everything in the model's state_dict (torch_utils.py:304-346): same constructor,
    `.ema` (an eval-mode deep copy, fp32, gradients off), `.updates`, `.decay`, `update(model)`, `update_attr(...)`.

    `update` is one `dsn_ema_step` launch over a device table of (ema tensor, model tensor) pairs instead of two ATen kernels
    per state_dict entry (~700 launches for DeSeNet-s); the arithmetic is the reference's, bit for bit (see the kernel).
    Under hipGraph capture only the launch is recorded; call `tick()` before each replay to advance `updates` and upload the
    new decay (desenet_amd.graph.GraphedTrainStep does)."""

    RING = 16

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._table = None

    def _build(self, model):
        from ... import _lib
        msd = de_parallel(model).state_dict()
        chunk = _lib.lib().dsn_sgd_chunk()
        pairs = []
        for k, v in self.ema.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            m = msd[k]
            if not (v.is_cuda and m.is_cuda):
                raise RuntimeError("desenet_amd.ModelEMA updates on an MI355X only (there is no CPU fallback)")
            if v.dtype != torch.float32 or m.dtype != torch.float32 or not v.is_contiguous() or not m.is_contiguous() \
                    or v.shape != m.shape:
                raise TypeError(f"ModelEMA: state_dict entry '{k}' must be contiguous fp32 of the model's shape")
            if v.numel():
                pairs.append((v, m))
        descs = (_lib.dsn_ema_desc * len(pairs))()
        first = 0
        for i, (v, m) in enumerate(pairs):
            descs[i] = _lib.dsn_ema_desc(v.data_ptr(), m.data_ptr(), v.numel(), first, 0)
            first += (v.numel() + chunk - 1) // chunk
        dev = pairs[0][0].device
        return dict(key=tuple((v.data_ptr(), m.data_ptr()) for v, m in pairs), n=len(pairs), chunks=first, pairs=pairs,
                    descs=torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev),
                    coef=torch.zeros(2, dtype=torch.float32, device=dev),
                    # ring of pinned upload slots: graph replay lets the host run many steps ahead of the GPU, so a slot is
                    # only rewritten after the async H2D copy that last read it has executed (event per slot)
                    host=[torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(self.RING)],
                    events=[None] * self.RING, slot=0)

    def tick(self):
        """updates += 1 and upload {d, 1 - d} for the next (possibly graph-replayed) launch."""
        self.updates += 1
        d = self.decay(self.updates)
        t = self._table
        i = t["slot"]
        t["slot"] = (i + 1) % self.RING
        if t["events"][i] is not None:
            t["events"][i].synchronize()     # the copy that last read this slot has run (RING steps ago: normally long done)
        h = t["host"][i]
        h[0] = d                         # float(d) and float(1. - d): what ATen's scalar multiply sees
        h[1] = 1.0 - d
        t["coef"].copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        t["events"][i] = ev

    def launch(self):
        from ... import _lib
        from ...hip_ops import stream_ptr
        t = self._table
        _lib.check(_lib.lib().dsn_ema_step(t["descs"].data_ptr(), t["n"], t["chunks"], t["coef"].data_ptr(), stream_ptr()),
                   "ema_step")

    def update(self, model):
        with torch.no_grad():
            capturing = torch.cuda.is_current_stream_capturing()
            if self._table is None or (not capturing and self._table["probe"] != self._probe(model)):
                if capturing:
                    raise RuntimeError("ModelEMA: call update() once eagerly before capturing a graph")
                self._table = self._build(model)
                self._table["probe"] = self._probe(model)
            if not capturing:
                self.tick()
            self.launch()

    @staticmethod
    def _probe(model):
        """Cheap staleness check of the device table: storage of the first / last parameter (a `.to()` or a re-created
        parameter moves them; optimizer steps and load_state_dict do not)."""
        ps = list(de_parallel(model).parameters())
        return (ps[0].data_ptr(), ps[-1].data_ptr(), len(ps))

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        copy_attr(self.ema, model, include, exclude)
