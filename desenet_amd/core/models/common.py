"""Drop-in mirror of the reference's `core.models.common` for the DeSeNet-s hot path, running on libdesenet_hip.so.

Same class names, constructor signatures, attribute names and therefore the same `state_dict` keys/shapes as the
reference (so `intersect_dicts` loading, EMA and checkpoints interchange through `state_dict`):

    autopad, Conv              core/models/common.py:32-56       Bottleneck   :101-111      C3      :133-145
    SPP                        :172-185                          FFM          :222-242      RFB2    :504-545
    PyramidPooling             :588-615                          Focus        :618-627      Concat  :686-693

`nn.Conv2d` / `nn.BatchNorm2d` children are kept as PARAMETER CONTAINERS only (their ATen forward is never called);
every forward/backward is HIP kernels through desenet_amd.hip_ops.  Activations are logical NCHW tensors stored NHWC.
Modules of the reference that DeSeNet-s never instantiates (DWConv, TransformerBlock, GhostConv, AutoShape, ...) are
out of scope.
"""
from __future__ import annotations

import math
from typing import List, Union

import torch
import torch.nn as nn

from ... import hip_ops as ops
from ...conv_impl import (BN_MOMENTUM, _bn_sync, _conv_geom, _grad_slot, act_code, conv_block_bwd, conv_block_fwd, out_shape,
                          packed_fwd, pair_block_bwd, pair_block_fwd, pair_ready)
from ...hip_ops import ACT_NONE, ACT_SIGMOID, ACT_SILU
from ...runtime import compute_dtype, run_module


def autopad(k: Union[int, List[int]], p=None):
    """'same' padding (reference common.py:32-39)."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _as_input(x: torch.Tensor) -> torch.Tensor:
    """Bring a caller tensor to the compute dtype / NHWC storage (no-op inside a network)."""
    dt = compute_dtype()
    if x.dtype != dt:
        x = x.to(dt)
    return ops.as_act(x)


class HipModule(nn.Module):
    def forward(self, x):
        return run_module(self, x)

    def fwd(self, x, tape=None, out=None):  # pragma: no cover - interface
        raise NotImplementedError

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):  # pragma: no cover - interface
        raise NotImplementedError


class Conv(HipModule):
    """Conv2d(bias=False) + BatchNorm2d + SiLU, BN skipped on 1x1 maps while un-fused (quirk Q1)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        ap = autopad(k, p)
        assert isinstance(ap, int)
        self.conv = nn.Conv2d(c1, c2, k, s, ap, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    @property
    def fused(self) -> bool:
        return not hasattr(self, "bn")

    def fwd(self, x, tape=None, out=None, residual=None, ci_pad=None):
        x = _as_input(x)
        bn = None if self.fused else self.bn
        return conv_block_fwd(x, self.conv, bn, act_code(self.act), self.training, tape, out, residual,
                              q1=not self.fused, ci_pad=ci_pad)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True, residual=None, fuse_up=False):
        """fuse_up: this call completes the gradient of the block's input (conv_impl.conv_block_bwd)."""
        return conv_block_bwd(tape, dy, dx, acc, need_dx, residual, fuse_up)

    def forward_fuse(self, x):   # reference API (common.py:55-56); the fused state is detected from the missing `bn`
        return self.forward(x)


class Bottleneck(HipModule):
    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def fwd(self, x, tape=None, out=None):
        x = _as_input(x)
        t = self.cv1.fwd(x, tape)
        # the shortcut is added by the elementwise pass that writes z = act(bn(y))
        return self.cv2.fwd(t, tape, out, residual=x if self.add else None)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True, fuse_up=False):
        dt = self.cv2.bwd(tape, dy, fuse_up=True)               # (cv1's output has no other consumer: dt is complete)
        # shortcut: d x = dy + d cv1 -- dy rides in the epilogue of cv1's input gradient (no copy / add pass)
        return self.cv1.bwd(tape, dt, dx, acc, need_dx, residual=dy if (self.add and need_dx) else None, fuse_up=fuse_up)


class C3(HipModule):
    """cv3(cat(m(cv1(x)), cv2(x))): the cat is never materialised -- both producers write into slices of one buffer."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def fwd(self, x, tape=None, out=None):
        x = _as_input(x)
        c_ = self.cv1.conv.out_channels
        n, _, h, w = x.shape
        blocks = list(self.m)
        hit = pair_ready(self.cv2, self.cv1, x, tape, x.dtype) if blocks else None
        if hit is not None:
            # training: cv2 | cv1 as ONE 2c_-wide convolution + ONE BatchNorm launch.  Buffer [m(..) | cv2(x) | cv1(x)]: the
            # merged block writes the last two thirds, the bottleneck chain reads the last third and its final block writes
            # the first, cv3 reads the first two thirds -- every operand is a channel slice, nothing is copied.
            cat3 = ops.new_act(n, 3 * c_, h, w, x.dtype, x.device)
            # every consumer inside the block is a convolution (Bottleneck.cv1, cv3) or the materialising pass of a shortcut:
            # the pair and the chain's last output stay raw
            pair_block_fwd(x, self.cv2, self.cv1, hit, tape, cat3[:, c_:])
            a = cat3[:, 2 * c_:]
            for i, b in enumerate(blocks):
                a = b.fwd(a, tape, cat3[:, :c_] if i == len(blocks) - 1 else None)
            z = self.cv3.fwd(cat3[:, :2 * c_], tape, out)
            tape.push("c3-merged")
            return z
        cat = ops.new_act(n, 2 * c_, h, w, x.dtype, x.device)
        a = self.cv1.fwd(x, tape, cat[:, :c_] if not blocks else None)
        for i, b in enumerate(blocks):
            a = b.fwd(a, tape, cat[:, :c_] if i == len(blocks) - 1 else None)
        self.cv2.fwd(x, tape, cat[:, c_:])
        z = self.cv3.fwd(cat, tape, out)
        if tape is not None:
            tape.push("c3-plain")
        return z

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True, fuse_up=False):
        """Inside the block every intermediate has exactly one consumer, so each input-gradient launch completes the gradient of its
        producer(s) and carries their BatchNorm backward sums (fuse_up=True); the block's own input gradient does when the caller
        says so."""
        c_ = self.cv1.conv.out_channels
        if tape.pop() == "c3-merged":
            # gradient buffer [d m(..) | d cv2 | d cv1]: cv3's dgrad fills the first two thirds, the chain's first block the
            # last one; the merged block reads the last two thirds
            n, _, h, w = dy.shape
            d3 = ops.new_act(n, 3 * c_, h, w, dy.dtype, dy.device)
            self.cv3.bwd(tape, dy, d3[:, :2 * c_], False, fuse_up=True)
            da = d3[:, :c_]
            blocks = list(self.m)
            for i, b in enumerate(reversed(blocks)):
                da = b.bwd(tape, da, d3[:, 2 * c_:], False, fuse_up=True) if i == len(blocks) - 1 else b.bwd(tape, da, fuse_up=True)
            return pair_block_bwd(tape, d3[:, c_:], dx, acc, need_dx, fuse_up=fuse_up)
        dcat = self.cv3.bwd(tape, dy, fuse_up=True)
        dx = self.cv2.bwd(tape, dcat[:, c_:], dx, acc, need_dx)
        da = dcat[:, :c_]
        for b in reversed(list(self.m)):
            da = b.bwd(tape, da, fuse_up=True)
        return self.cv1.bwd(tape, da, dx, True, need_dx, fuse_up=fuse_up)


class SPP(HipModule):
    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def fwd(self, x, tape=None, out=None):
        x = _as_input(x)
        c_ = self.cv1.conv.out_channels
        n, _, h, w = x.shape
        cat = ops.new_act(n, c_ * (len(self.m) + 1), h, w, x.dtype, x.device)
        x0 = self.cv1.fwd(x, tape, cat[:, :c_])                  # (the max pools read z: materialised)
        idxs = [torch.empty((n, h, w, c_), dtype=torch.int32, device=x.device) if tape is not None else None for _ in self.m]
        ops.maxpool_s1_multi(x0, [cat[:, i * c_:(i + 1) * c_] for i in range(1, len(self.m) + 1)],
                             [int(mp.kernel_size) for mp in self.m], idxs)          # one launch, x0 read once
        if tape is not None:
            tape.push(idxs)
        return self.cv2.fwd(cat, tape, out)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True, fuse_up=False):
        c_ = self.cv1.conv.out_channels
        dcat = self.cv2.bwd(tape, dy)          # (cv1's output also feeds the pools: its gradient is completed by the scatter below)
        idxs = tape.pop()
        d0 = dcat[:, :c_]
        ops.maxpool_s1_bwd_multi([dcat[:, i * c_:(i + 1) * c_] for i in range(1, len(self.m) + 1)], idxs,
                                 [int(mp.kernel_size) for mp in self.m], d0, accumulate=True)      # one pass for the three pools
        return self.cv1.bwd(tape, d0, dx, acc, need_dx, fuse_up=fuse_up)


class Focus(HipModule):
    """Space-to-depth straight from the NCHW image into the conv's NHWC input (bit-exact index map)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    def fwd(self, x, tape=None, out=None):
        if not x.is_cuda:
            raise RuntimeError("desenet_amd kernels run on an MI355X only: got a CPU tensor (there is no CPU fallback)")
        dt = compute_dtype()
        n, c, h, w = x.shape
        vec = 4 if dt == torch.float32 else 8
        cpad = (4 * c + vec - 1) // vec * vec          # 12 -> 12 (fp32) / 16 (bf16): 16-byte channel vectors
        s2d = ops.new_act(n, cpad, h // 2, w // 2, dt, x.device)
        ops.focus_s2d(x, s2d)
        return self.conv.fwd(s2d, tape, out, ci_pad=cpad)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        if need_dx:
            raise NotImplementedError("gradient w.r.t. the input image is not part of the training path")
        return self.conv.bwd(tape, dy, need_dx=False)


class Concat(HipModule):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    @staticmethod
    def _adjacent(xs):
        """xs are consecutive channel slices of ONE NHWC buffer -> the covering view, else None."""
        x0 = xs[0]
        es = x0.element_size()
        ptr = x0.data_ptr()
        for t in xs:
            if (t.dtype != x0.dtype or t.shape[0] != x0.shape[0] or t.shape[2:] != x0.shape[2:]
                    or t.stride() != x0.stride() or t.data_ptr() != ptr
                    or t.untyped_storage().data_ptr() != x0.untyped_storage().data_ptr()):
                return None
            ptr += t.shape[1] * es
        ctot = sum(t.shape[1] for t in xs)
        ldc = ops._nhwc_ldc(x0)
        if ldc is None or ldc < ctot:
            return None
        return torch.as_strided(x0, (x0.shape[0], ctot, x0.shape[2], x0.shape[3]), x0.stride(), x0.storage_offset())

    def fwd(self, xs, tape=None, out=None):
        assert self.d == 1, "channel concat only"
        xs = [_as_input(t) for t in xs]
        sizes = [t.shape[1] for t in xs]
        if tape is not None:
            tape.push(sizes)
        merged = self._adjacent(xs) if out is None else None
        if merged is not None:
            return merged
        n, _, h, w = xs[0].shape
        if out is None:
            out = ops.new_act(n, sum(sizes), h, w, xs[0].dtype, xs[0].device)
        c0 = 0
        for t in xs:
            dst = out[:, c0:c0 + t.shape[1]]
            if dst.data_ptr() != t.data_ptr():
                ops.copy(t, dst)
            c0 += t.shape[1]
        return out

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        sizes = tape.pop()
        outs, c0 = [], 0
        for i, c in enumerate(sizes):
            g = dy[:, c0:c0 + c]
            if dx is not None and dx[i] is not None:
                ops.copy(g, dx[i], accumulate=bool(acc[i]) if isinstance(acc, (list, tuple)) else bool(acc))
                g = dx[i]
            outs.append(g)
            c0 += c
        return outs


class Upsample(nn.Upsample, HipModule):
    """nn.Upsample(None, 2, 'nearest') of the model yaml (yolov5s_seg.yaml:31,36)."""

    def forward(self, x):
        return run_module(self, x)

    def fwd(self, x, tape=None, out=None):
        if self.mode != "nearest" or float(self.scale_factor) != 2.0:
            raise NotImplementedError("only nearest x2 up-sampling is on the DeSeNet path")
        x = _as_input(x)
        n, c, h, w = x.shape
        if out is None:
            out = ops.new_act(n, c, 2 * h, 2 * w, x.dtype, x.device)
        if tape is not None:
            tape.push((n, c, h, w))
        return ops.upsample_nearest2x(x, out)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        n, c, h, w = tape.pop()
        if not need_dx:
            return None
        if dx is None:
            dx, acc = ops.new_act(n, c, h, w, dy.dtype, dy.device), False
        return ops.upsample_nearest2x_bwd(dy, dx, accumulate=acc)


# ----------------------------------------------------------------------------------------------------- seg blocks
class _ConvBnAct(nn.Sequential):
    """The raw nn.Conv2d + nn.BatchNorm2d + nn.SiLU triple of RFB2.branch1/2 (never folded by Model.fuse(): quirk Q3)."""

    def __init__(self, c, d):
        super().__init__(nn.Conv2d(c, c, kernel_size=3, stride=1, padding=d, dilation=d, bias=False),
                         nn.BatchNorm2d(c), nn.SiLU())

    def forward(self, x):
        return run_module(self, x)

    def fwd(self, x, tape=None, out=None):
        return conv_block_fwd(_as_input(x), self[0], self[1], ACT_SILU, self.training, tape, out)

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True, fuse_up=False):
        return conv_block_bwd(tape, dy, dx, acc, need_dx, fuse_up=fuse_up)


import os as _os
_RFB_FUSE = int(_os.environ.get("DSN_BNRED_RFB", "1"))


class RFB2(HipModule):
    def __init__(self, in_planes, out_planes, map_reduce=4, d=[2, 3], has_global=False):
        super().__init__()
        if has_global:
            raise NotImplementedError("RFB2(has_global=True) is not instantiated by SegMaskPSP (yolo.py:178)")
        self.out_channels = out_planes
        self.has_global = has_global
        inter = in_planes // map_reduce
        self.branch0 = nn.Sequential(Conv(in_planes, inter, k=1, s=1), Conv(inter, inter, k=3, s=1))
        self.branch1 = _ConvBnAct(inter, d[0])
        self.branch2 = _ConvBnAct(inter, d[1])
        self.branch3 = nn.Sequential(Conv(in_planes, inter, k=1, s=1))
        self.ConvLinear = Conv(4 * inter, out_planes, k=1, s=1)

    def fwd(self, x, tape=None, out=None):
        x = _as_input(x)
        i = self.branch1[0].in_channels
        n, _, h, w = x.shape
        hit = pair_ready(self.branch3[0], self.branch0[0], x, tape, x.dtype)
        if hit is not None:
            # training: branch3's and branch0's first 1x1 read the same x -> ONE convolution + ONE BatchNorm launch writing
            # [x3 | t] into the tail of a 5i-wide buffer [x0 | x1 | x2 | x3 | t]; ConvLinear reads the first four fifths
            cat = ops.new_act(n, 5 * i, h, w, x.dtype, x.device)
            # every intermediate of the block feeds convolutions only (branch chain, ConvLinear): all of them stay raw
            pair_block_fwd(x, self.branch3[0], self.branch0[0], hit, tape, cat[:, 3 * i:])
            t = cat[:, 4 * i:]
        else:
            cat = ops.new_act(n, 4 * i, h, w, x.dtype, x.device)     # [x0 | x1 | x2 | x3]
            self.branch3[0].fwd(x, tape, cat[:, 3 * i:])
            t = self.branch0[0].fwd(x, tape)
        x0 = self.branch0[1].fwd(t, tape, cat[:, :i])
        x1 = self.branch1.fwd(x0, tape, cat[:, i:2 * i])
        self.branch2.fwd(x1, tape, cat[:, 2 * i:3 * i])
        z = self.ConvLinear.fwd(cat[:, :4 * i], tape, out)
        if tape is not None:
            tape.push("rfb-merged" if hit is not None else "rfb-plain")
        return z

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        i = self.branch1[0].in_channels
        merged = tape.pop() == "rfb-merged"
        # BatchNorm backward sums ride in the launch that completes a gradient (conv_impl.conv_block_bwd: fuse_up): ConvLinear's
        # for x2 and x3 (its only consumer), branch2's and branch1's accumulating dgrads for x1 and x0, branch0[1]'s for t
        fu, fr = (True, (2 * i, 4 * i)) if _RFB_FUSE else (False, False)
        if merged:
            n, _, h, w = dy.shape
            d5 = ops.new_act(n, 5 * i, h, w, dy.dtype, dy.device)     # [d x0 | d x1 | d x2 | d x3 | d t]
            dcat = self.ConvLinear.bwd(tape, dy, d5[:, :4 * i], False, fuse_up=fr)
        else:
            dcat = self.ConvLinear.bwd(tape, dy, fuse_up=fr)
        d1 = self.branch2.bwd(tape, dcat[:, 2 * i:3 * i], dcat[:, i:2 * i], True, fuse_up=fu)     # dx1 += ...
        d0 = self.branch1.bwd(tape, d1, dcat[:, :i], True, fuse_up=fu)                            # dx0 += ...
        if merged:
            self.branch0[1].bwd(tape, d0, d5[:, 4 * i:], False, fuse_up=fu)
            return pair_block_bwd(tape, d5[:, 3 * i:], dx, acc, need_dx)
        dt = self.branch0[1].bwd(tape, d0, fuse_up=fu)
        dx = self.branch0[0].bwd(tape, dt, dx, acc, need_dx)
        return self.branch3[0].bwd(tape, dcat[:, 3 * i:], dx, True, need_dx)


class PyramidPooling(HipModule):
    def __init__(self, in_channels, k=[1, 2, 3, 6], short_cut=False):
        super().__init__()
        self.short_cut = short_cut
        self.pool1, self.pool2, self.pool3, self.pool4 = (nn.AdaptiveAvgPool2d(v) for v in k)
        oc = in_channels // 4
        self.conv1 = Conv(in_channels, oc, k=1)
        self.conv2 = Conv(in_channels, oc, k=1)
        self.conv3 = Conv(in_channels, oc, k=1)
        self.conv4 = Conv(in_channels, oc, k=1)

    def _branches(self):
        return [(self.pool1, self.conv1), (self.pool2, self.conv2), (self.pool3, self.conv3), (self.pool4, self.conv4)]

    def fwd(self, x, tape=None, out=None):
        """`out` (optional): the full [x | f1..f4] buffer whose first slice already IS x (SegMaskPSP arranges that)."""
        x = _as_input(x)
        n, c, h, w = x.shape
        oc = self.conv1.conv.out_channels
        base = c if self.short_cut else 0
        if out is None:
            out = ops.new_act(n, base + 4 * oc, h, w, x.dtype, x.device)
        if self.short_cut and out[:, :c].data_ptr() != x.data_ptr():
            ops.copy(x, out[:, :c])
        # each stage of the four branches is ONE launch: pools (x read once), then the tiny convs, then the upsampling
        ks = [p.output_size if isinstance(p.output_size, int) else p.output_size[0] for p, _ in self._branches()]
        pooled = ops.adaptive_avgpool_multi(x, [ops.new_act(n, c, k, k, x.dtype, x.device) for k in ks])
        fs = self._fwd_branches_fused(pooled, tape) if self._fusable(x, ks, tape) else None
        if fs is None:
            fs = [conv.fwd(pooled[j], tape) for j, (_, conv) in enumerate(self._branches())]
        ops.bilinear_ac_multi(fs, [out[:, base + j * oc: base + (j + 1) * oc] for j in range(len(fs))])
        if tape is not None:
            tape.push((n, c, h, w))
        return out

    # ---- the four branches' conv + BatchNorm + act as ONE launch each way (csrc/pp_fused.hip) ---------------------------------
    def _fusable(self, x, ks, tape) -> bool:
        """Training step in bf16, ordinary (un-fused, un-synchronised, trainable) branches that fit the kernel's LDS budget."""
        import os
        if not self.training or tape is None or x.dtype != torch.bfloat16 or os.environ.get("DSN_PP_FUSED", "1") == "0":
            return False
        convs = [cv for _, cv in self._branches()]
        if len(convs) > 4:
            return False
        bn0 = convs[0].bn if not convs[0].fused else None
        for cv in convs:
            if cv.fused or cv.conv.bias is not None or not cv.conv.weight.requires_grad or _conv_geom(cv.conv) != (1, 1, 0, 1):
                return False
            bn = cv.bn
            if _bn_sync(bn) is not None or bn.weight is None or not (bn.weight.requires_grad and bn.bias.requires_grad):
                return False
            if bn.eps != bn0.eps or bn.momentum != bn0.momentum or act_code(cv.act) != act_code(convs[0].act):
                return False
            # a frozen / eval-mode BatchNorm inside a training model keeps its running statistics (conv_block_fwd's `frozen`
            # record); the fused kernel always computes batch statistics and updates the running averages
            if not (cv.training and bn.training) or bn.running_mean is None or bn.running_var is None:
                return False
        n, c = x.shape[0], x.shape[1]
        return ops.pp_stages_supported(n * max(ks) ** 2, c, convs[0].conv.out_channels, x.dtype)

    def _fwd_branches_fused(self, pooled, tape):
        convs = [cv for _, cv in self._branches()]
        dtype, dev = pooled[0].dtype, pooled[0].device
        n, oc = pooled[0].shape[0], convs[0].conv.out_channels
        act = act_code(convs[0].act)
        ws = [packed_fwd(cv.conv, dtype, None, None)[0] for cv in convs]
        # quirk Q1 (common.py:53): no BatchNorm on a 1 x 1 map
        bns = [None if p.shape[2] * p.shape[3] == 1 else cv.bn for p, cv in zip(pooled, convs)]
        zs = [ops.new_act(n, oc, p.shape[2], p.shape[3], dtype, dev) for p in pooled]
        ys = [ops.new_act(n, oc, p.shape[2], p.shape[3], dtype, dev) for p in pooled]
        stats = [None if bn is None else torch.empty((4, oc), dtype=torch.float32, device=dev) for bn in bns]
        bn0 = convs[0].bn
        ops.pp_stages_fwd(pooled, ws, bns, zs, ys, stats, act, bn0.momentum if bn0.momentum is not None else BN_MOMENTUM, bn0.eps)
        for p, cv, bn, z, st in zip(pooled, convs, bns, zs, stats):
            # the records conv_block_bwd reads: the per-branch backward stays valid on them
            rec = dict(conv=cv.conv, bn=bn, act=act, x=p, x_in=p, ci_pad=None, geom=_conv_geom(cv.conv), plain=False, y=z, pp_fused=True)
            if bn is not None:
                if bn.num_batches_tracked is not None and not bn.__dict__.get("_dsn_shared_counter"):
                    bn.num_batches_tracked.add_(1)
                rec.update(scale=st[0], shift=st[1], mean=st[2], rstd=st[3], stats=st, frozen=False, sync=None)
            tape.push(rec)
        return ys

    def _bwd_branches_fused(self, tape, dfs):
        """dpools (one per branch) or None when the branches were not recorded by the fused forward / have no gradient slots."""
        nb = len(self._branches())
        recs = [tape.pop() for _ in range(nb)][::-1]
        slots = []
        ok = all(isinstance(r, dict) and r.get("pp_fused") for r in recs)
        if ok:
            for r in recs:
                bn = r["bn"]
                sl = (_grad_slot(r["conv"].weight), _grad_slot(bn.weight) if bn is not None else None,
                      _grad_slot(bn.bias) if bn is not None else None)
                if sl[0] is None or (bn is not None and (sl[1] is None or sl[2] is None)) or r["conv"].weight in tape.grads:
                    ok = False
                    break
                slots.append(sl)
        if not ok:
            tape.cursor += nb                        # un-pop: the per-branch backward walks the same records
            return None
        dtype = dfs[0].dtype
        xs = [r["x"] for r in recs]
        dxs = [ops.new_act(*x.shape, dtype, x.device) for x in xs]
        ops.pp_stages_bwd(xs, [packed_fwd(r["conv"], dtype, None, None)[0] for r in recs], [r["y"] for r in recs], dfs, dxs,
                          [r.get("stats") for r in recs], [s[1] for s in slots], [s[2] for s in slots], [s[0] for s in slots],
                          recs[0]["act"], accumulate=True)
        return dxs

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        n, c, h, w = tape.pop()
        oc = self.conv1.conv.out_channels
        base = c if self.short_cut else 0
        if dx is None:
            dx, acc = ops.new_act(n, c, h, w, dy.dtype, dy.device), False
            if self.short_cut:
                ops.copy(dy[:, :c], dx)
                acc = True
        elif self.short_cut:
            ops.copy(dy[:, :c], dx, accumulate=acc)
            acc = True
        br = list(enumerate(self._branches()))
        ks = [p.output_size if isinstance(p.output_size, int) else p.output_size[0] for _, (p, _) in br]
        dfs = ops.bilinear_ac_bwd_multi([dy[:, base + j * oc: base + (j + 1) * oc] for j, _ in br],
                                        [ops.new_act(n, oc, k, k, dy.dtype, dy.device) for k in ks])    # one launch pair
        dpools = self._bwd_branches_fused(tape, dfs)
        if dpools is None:
            dpools = [None] * len(br)
            for j, (_, conv) in reversed(br):         # tape order: the branches' convs were recorded 0..3
                dpools[j] = conv.bwd(tape, dfs[j])
        ops.adaptive_avgpool_bwd_multi(dpools[::-1], dx, accumulate=acc)      # the four grids in one pass over dx
        return dx


def _small_conv_fused_ok(x, conv, tape, training) -> bool:
    """A bias-free 1x1 convolution + activation on a handful of pixels (FFM's attention vectors) through the one-block kernels of
    csrc/pp_fused.hip instead of a convolution launch and an activation launch."""
    import os
    # measured in the step: 1935 img/s with FFM's two attention convs on these kernels against 1942 without (a 1024-thread block
    # staging a 128 x 128 weight matrix for 8 pixels is a longer chain than the conv + activation launches): DSN_PP_FUSED=3 only
    if not training or tape is None or x.dtype != torch.bfloat16 or os.environ.get("DSN_PP_FUSED", "1") != "3":
        return False
    if conv.bias is not None or not conv.weight.requires_grad or _conv_geom(conv) != (1, 1, 0, 1):
        return False
    n, c, h, w = x.shape
    return ops._nhwc_ldc(x) == c and ops.pp_stages_supported(n * h * w, c, conv.out_channels, x.dtype)


def _small_conv_fused_fwd(x, conv, act, tape):
    n, _, h, w = x.shape
    z = ops.new_act(n, conv.out_channels, h, w, x.dtype, x.device)
    y = ops.new_act(n, conv.out_channels, h, w, x.dtype, x.device)
    ops.pp_stages_fwd([x], [packed_fwd(conv, x.dtype, None, None)[0]], [None], [z], [y], [None], act, 0.0, 0.0)
    tape.push(dict(conv=conv, bn=None, act=act, x=x, x_in=x, ci_pad=None, geom=_conv_geom(conv), plain=False, y=z, pp_fused=True))
    return y


def _small_conv_fused_bwd(tape, dy):
    """Input gradient, or None (record un-popped) when the record is not a fused one / the weight has no gradient slot."""
    rec = tape.pop()
    slot = _grad_slot(rec["conv"].weight) if isinstance(rec, dict) and rec.get("pp_fused") and rec.get("bn") is None else None
    if slot is None or rec["conv"].weight in tape.grads or ops._nhwc_ldc(dy) != dy.shape[1]:
        tape.cursor += 1
        return None
    x = rec["x"]
    dx = ops.new_act(*x.shape, dy.dtype, x.device)
    ops.pp_stages_bwd([x], [packed_fwd(rec["conv"], dy.dtype, None, None)[0]], [rec["y"]], [dy], [dx], [None], [None], [None], [slot],
                      rec["act"], accumulate=True)
    return dx


class FFM(HipModule):
    def __init__(self, in_chan, out_chan, reduction=1, is_cat=True, k=1):
        super().__init__()
        self.convblk = Conv(in_chan, out_chan, k=k, s=1)
        self.channel_attention = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            nn.Conv2d(out_chan, out_chan // reduction, kernel_size=1, stride=1, padding=0, bias=False),
            nn.SiLU(inplace=True),
            nn.Conv2d(out_chan // reduction, out_chan, kernel_size=1, stride=1, padding=0, bias=False),
            nn.Sigmoid(),
        )
        self.is_cat = is_cat

    def fwd(self, x, tape=None, out=None):
        if self.is_cat:
            x = Concat._adjacent([_as_input(t) for t in x]) if isinstance(x, (list, tuple)) else x
            if x is None:
                raise NotImplementedError("FFM(is_cat=True) expects adjacent slices; SegMaskPSP uses is_cat=False")
        x = _as_input(x)
        ca = self.channel_attention
        feat = self.convblk.fwd(x, tape)
        n, c, h, w = feat.shape
        gap = ops.adaptive_avgpool(feat, ops.new_act(n, c, 1, 1, feat.dtype, feat.device))
        if _small_conv_fused_ok(gap, ca[1], tape, self.training) and _small_conv_fused_ok(gap, ca[3], tape, self.training) \
                and ca[1].out_channels == ca[3].in_channels == c:
            a1 = _small_conv_fused_fwd(gap, ca[1], ACT_SILU, tape)
            att = _small_conv_fused_fwd(a1, ca[3], ACT_SIGMOID, tape)
        else:
            a1 = conv_block_fwd(gap, ca[1], None, ACT_SILU, self.training, tape)
            att = conv_block_fwd(a1, ca[3], None, ACT_SIGMOID, self.training, tape)
        if out is None:
            out = ops.new_act(n, c, h, w, feat.dtype, feat.device)
        ops.ffm_scale(feat, att, out)
        if tape is not None:
            tape.push((feat, att))
        return out

    def bwd(self, tape, dy, dx=None, acc=False, need_dx=True):
        feat, att = tape.pop()
        n, c, h, w = feat.shape
        dfeat = ops.new_act(n, c, h, w, feat.dtype, feat.device)
        datt = ops.new_act(n, c, 1, 1, feat.dtype, feat.device)
        ops.ffm_scale_bwd(dy, feat, att, dfeat, datt)
        da1 = _small_conv_fused_bwd(tape, datt)
        if da1 is None:
            da1 = conv_block_bwd(tape, datt)
        dgap = _small_conv_fused_bwd(tape, da1)
        if dgap is None:
            dgap = conv_block_bwd(tape, da1)
        ops.adaptive_avgpool_bwd(dgap, dfeat, accumulate=True)
        return self.convblk.bwd(tape, dfeat, dx, acc, need_dx)
