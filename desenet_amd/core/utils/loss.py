"""Mirror of the reference's training losses (core/utils/loss.py:91-243, core/utils/metrics.py:202-244).

v1 (this round): PyTorch-ROCm tensor ops on the GPU, same arithmetic as the reference -- SURVEY.md 8f ranks fused HIP loss
kernels (x8 bilinear + log-softmax + NLL; BCE-obj over N x 25200; build_targets) as the NEXT widening step.
`ComputeLoss(model)` / `SegmentationLosses()` keep the reference's call signatures:

    det_loss, items = compute_loss(det_pred, det_labels)      # det_loss is scaled by the batch size, items = (box, obj, cls)
    seg_loss = compute_seg_loss(seg_pred, seg_labels)

Deviations: the int64 clamp with float-tensor bounds of loss.py:218 (rejected by torch >= 1.12) uses integer bounds.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .torch_utils import de_parallel


def bbox_ciou(box1, box2, eps=1e-7):
    """CIoU of box1 (4,n) to box2 (n,4), both xywh (metrics.py:202-244 with x1y1x2y2=False, CIoU=True)."""
    box2 = box2.T
    b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    chh = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + chh ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def smooth_BCE(eps=0.1):
    return 1.0 - 0.5 * eps, 0.5 * eps


class ComputeLoss:
    def __init__(self, model, autobalance=False):
        if autobalance:
            raise NotImplementedError("autobalance is off in the reference training script")
        device = next(model.parameters()).device
        h = model.hyp
        if h.get("fl_gamma", 0.0) > 0:
            raise NotImplementedError("focal loss (fl_gamma > 0) is outside the scratch.yaml configuration")
        self.cls_pw = torch.tensor([h["cls_pw"]], device=device)
        self.obj_pw = torch.tensor([h["obj_pw"]], device=device)
        self.cp, self.cn = smooth_BCE(eps=h.get("label_smoothing", 0.0))
        det = de_parallel(model).model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.gr, self.hyp = 1.0, h
        self.na, self.nc, self.nl, self.anchors = det.na, det.nc, det.nl, det.anchors
        self.sort_obj_iou = False

    def __call__(self, p, targets):
        device = targets.device
        lcls, lbox, lobj = (torch.zeros(1, device=device) for _ in range(3))
        tcls, tbox, indices, anchors = self.build_targets(p, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros_like(pi[..., 0])
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, :2].sigmoid() * 2. - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                iou = bbox_ciou(torch.cat((pxy, pwh), 1).T, tbox[i])
                lbox = lbox + (1.0 - iou).mean()
                tobj[b, a, gj, gi] = ((1.0 - self.gr) + self.gr * iou.detach().clamp(0)).type(tobj.dtype)
                if self.nc > 1:
                    t = torch.full_like(ps[:, 5:], self.cn)
                    t[range(n), tcls[i]] = self.cp
                    lcls = lcls + F.binary_cross_entropy_with_logits(ps[:, 5:], t, pos_weight=self.cls_pw)
            lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj, pos_weight=self.obj_pw) * self.balance[i]
        lbox = lbox * self.hyp["box"]
        lobj = lobj * self.hyp["obj"]
        lcls = lcls * self.hyp["cls"]
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()

    def build_targets(self, p, targets):
        """Anchor matching (loss.py:170-223): a target (img, cls, x, y, w, h) is assigned to anchor a of level i when
        max(wh/anchor, anchor/wh) < anchor_t, at its own grid cell and at the up-to-two neighbouring cells its centre is
        closest to (|frac| < .5 and not on the border).  Returns per level: class ids, (dx, dy, w, h) boxes in grid
        units, (image, anchor, gy, gx) indices and the matched anchors -- in the reference's candidate order
        (offset-major, then anchor-major, then target order)."""
        dev = targets.device
        nt = targets.shape[0]
        thr = self.hyp["anchor_t"]
        shifts = torch.tensor([[0., 0.], [.5, 0.], [0., .5], [-.5, 0.], [0., -.5]], device=dev)
        tcls, tbox, indices, anch = [], [], [], []
        for i in range(self.nl):
            anchors = self.anchors[i]                                   # (na, 2) grid units
            ny, nx = p[i].shape[2], p[i].shape[3]
            size = torch.tensor([nx, ny], device=dev, dtype=torch.float32)
            if nt == 0:
                z = torch.zeros(0, dtype=torch.long, device=dev)
                indices.append((z, z, z, z))
                tbox.append(torch.zeros(0, 4, device=dev))
                anch.append(anchors[z])
                tcls.append(z)
                continue
            gxy_all = targets[:, 2:4] * size                            # (nt, 2)
            gwh_all = targets[:, 4:6] * size
            ratio = gwh_all[None] / anchors[:, None]                    # (na, nt, 2)
            ok = torch.max(ratio, 1. / ratio).max(2)[0] < thr           # (na, nt)
            a_id, t_id = ok.nonzero(as_tuple=True)                      # anchor-major
            gxy, gwh = gxy_all[t_id], gwh_all[t_id]
            inv = size - gxy
            near_lo = (gxy % 1. < .5) & (gxy > 1.)                      # (n, 2): x-1 / y-1 neighbours
            near_hi = (inv % 1. < .5) & (inv > 1.)                      # x+1 / y+1 neighbours
            take = torch.stack((torch.ones_like(near_lo[:, 0]), near_lo[:, 0], near_lo[:, 1], near_hi[:, 0],
                                near_hi[:, 1]))                         # (5, n)
            o_id, m_id = take.nonzero(as_tuple=True)                    # offset-major
            gxy, gwh = gxy[m_id], gwh[m_id]
            cell = (gxy - shifts[o_id]).long()
            img = targets[t_id[m_id], 0].long()
            a = a_id[m_id]
            indices.append((img, a, cell[:, 1].clamp(0, ny - 1), cell[:, 0].clamp(0, nx - 1)))
            tbox.append(torch.cat((gxy - cell, gwh), 1))
            anch.append(anchors[a])
            tcls.append(targets[t_id[m_id], 1].long())
        return tcls, tbox, indices, anch


class SegmentationLosses(nn.CrossEntropyLoss):
    """Plain 2-D cross entropy, ignore_index=-1 (the no-aux, no-se_loss mode SegMaskPSP trains with; loss.py:242-243)."""

    def __init__(self, se_loss=False, se_weight=0.2, nclass=-1, aux_num=2, aux=False, aux_weight=0.1, weight=None,
                 ignore_index=-1):
        super().__init__(weight, None, ignore_index)
        if se_loss or aux:
            raise NotImplementedError("auxiliary / SE losses belong to heads outside the hot path (BiSe)")

    def forward(self, *inputs):
        return super().forward(*inputs)
