"""Mirror of the reference's training losses (core/utils/loss.py:91-243, core/utils/metrics.py:202-244) on HIP kernels
(desenet_amd/csrc/loss.hip): anchor matching (build_targets), CIoU box loss, BCE objectness / class losses and the
segmentation cross entropy run on the GPU with no host synchronisation and fixed launch shapes, so the whole training step
is hipGraph-capturable.  `ComputeLoss(model)` / `SegmentationLosses()` keep the reference's call signatures:

    det_loss, items = compute_loss(det_pred, det_labels)      # det_loss is scaled by the batch size, items = (box, obj, cls)
    seg_loss = compute_seg_loss(seg_pred, seg_labels)

and are differentiable through `torch.autograd.Function`s whose backward hands out the gradients the kernels already
produced.  `forward_backward(...)` returns loss and gradients directly (what desenet_amd.graph captures).

Deviations from the reference, both deliberate: the int64 clamp with float-tensor bounds of loss.py:218 (rejected by
torch >= 1.12) is an integer clamp; a cell matched by several candidates takes the LAST one in the reference's candidate
order as its objectness target (the reference's scatter with duplicate indices is order-dependent on a GPU).
Reference options off in scripts/train.py but inside loss.py:91-168 are implemented: focal loss (hyp['fl_gamma'] > 0) and
`autobalance=True` -- the latter keeps the per-level balance on the DEVICE (the reference reads each level's loss back with
`.item()`, a host synchronisation per level and step); `.balance` returns it as a list on demand.  Still raising: auxiliary / SE
segmentation losses (heads outside the hot path).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import hip_ops as ops
from .torch_utils import de_parallel


def smooth_BCE(eps=0.1):
    return 1.0 - 0.5 * eps, 0.5 * eps


class _DetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, *p):
        out, dp = owner._launch([t.detach() for t in p], targets, 1.0)
        ctx.dp = dp
        ctx.mark_non_differentiable(out)
        return (out[0:1].clone(), out[1:4].clone())

    @staticmethod
    def backward(ctx, g, _g_items):
        return (None, None, *[d * g for d in ctx.dp])


class ComputeLoss:
    def __init__(self, model, autobalance=False):
        h = model.hyp
        self.fl_gamma = float(h.get("fl_gamma", 0.0))                   # loss.py:106-110: FocalLoss around both criteria when > 0
        self.cp, self.cn = smooth_BCE(eps=h.get("label_smoothing", 0.0))
        det = de_parallel(model).model[-1]
        self._balance0 = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.autobalance = bool(autobalance)
        self.ssi = list(float(s) for s in det.stride).index(16.0) if autobalance else 0      # loss.py:113
        self._balance_dev = None                                        # created on the first call (device of the predictions)
        self.gr, self.hyp = 1.0, h
        self.na, self.nc, self.nl, self.anchors = det.na, det.nc, det.nl, det.anchors
        self._anchors_host = det.anchors.detach().float().cpu().reshape(-1).tolist()    # once: (nl, na, 2) grid units

    @property
    def balance(self):
        """The per-level objectness weights as a list (the reference's attribute).  With autobalance they live on the device and
        this read synchronises; the step itself never does."""
        if self._balance_dev is not None:
            return [float(v) for v in self._balance_dev.detach().cpu()]
        return list(self._balance0)

    def _launch(self, p, targets, gain):
        h = self.hyp
        if self.autobalance and self._balance_dev is None:
            self._balance_dev = torch.tensor(self._balance0[:len(p)], dtype=torch.float32, device=p[0].device)
        return ops.det_loss(p, targets, self._anchors_host, self._balance0[:len(p)], h["box"], h["obj"], h["cls"], h["cls_pw"],
                            h["obj_pw"], h["anchor_t"], self.cp, self.cn, self.nc, gain, fl_gamma=self.fl_gamma,
                            balance_dev=self._balance_dev, autobalance=self.autobalance, ssi=self.ssi)

    def __call__(self, p, targets):
        return _DetLossFn.apply(self, targets, *p)

    def forward_backward(self, p, targets, gain=1.0):
        """(out [4] = gain * {(lbox+lobj+lcls)*bs, lbox, lobj, lcls}, [d out[0] / d p_i]) without autograd."""
        return self._launch(p, targets, gain)


class _SegCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        out, dl = ops.seg_ce(logits.detach(), target, ignore_index, want_grad=True)
        ctx.dl = dl
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.dl * g, None, None


class SegmentationLosses(nn.CrossEntropyLoss):
    """Plain 2-D cross entropy, ignore_index=-1 (the no-aux, no-se_loss mode SegMaskPSP trains with; loss.py:242-243)."""

    def __init__(self, se_loss=False, se_weight=0.2, nclass=-1, aux_num=2, aux=False, aux_weight=0.1, weight=None,
                 ignore_index=-1):
        super().__init__(weight, None, ignore_index)
        if se_loss or aux:
            raise NotImplementedError("auxiliary / SE losses belong to heads outside the hot path (BiSe)")
        if weight is not None:
            raise NotImplementedError("class weights are not used by the reference training script")

    def forward(self, pred, target):
        return _SegCEFn.apply(pred, target, self.ignore_index)

    def forward_backward(self, pred, target, gain=1.0):
        """(out [2] = {mean CE, 1/valid}, gain * d loss / d pred) without autograd.  `pred` may be the seg head's LOW-resolution
        logits (SegMaskPSP under desenet_amd.graph.GraphedTrainStep(fuse_seg_loss=True): tagged `_dsn_seg_upsample = (H, W)`): the
        x8 bilinear up-sampling of yolo.py:183 is then fused with the cross entropy (dsn_seg_ce_up) and the returned gradient is
        the low-resolution one the head's backward expects.  Shapes the fused kernel does not take run the same arithmetic
        un-fused (up-sample, full-resolution cross entropy, up-sampling backward) and return the same kind of gradient."""
        up = getattr(pred, "_dsn_seg_upsample", None)
        if up is not None:
            try:
                return ops.seg_ce_up(pred, target, up, self.ignore_index, gain=gain)
            except ops.KernelUnsupported:
                n, c, h, w = pred.shape
                seg = torch.empty((n, c, int(up[0]), int(up[1])), dtype=torch.float32, device=pred.device)
                ops.bilinear_ac(pred, seg, out_nchw=True)
                out, dseg = ops.seg_ce(seg, target, self.ignore_index, want_grad=True)
                if gain != 1.0:
                    dseg = dseg * gain
                vec = 4 if pred.dtype == torch.float32 else 8
                dl = ops.bilinear_ac_bwd(dseg, ops.new_act(n, c, h, w, pred.dtype, pred.device, zero=True, ldc_align=vec),
                                         dy_nchw=True)
                dl._dsn_zero_padded = True
                return out, dl
        out, dl = ops.seg_ce(pred, target, self.ignore_index, want_grad=True)
        return out, (dl if gain == 1.0 else dl * gain)
