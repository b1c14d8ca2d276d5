"""CPU ORACLE for the training losses -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see desenet_ref.py header).

Restates on PyTorch-CPU fp32 ops (autograd supplies the gradients):

    det_loss      core/utils/loss.py:91-223   ComputeLoss.__call__ / build_targets, label_smoothing 0 (cp=1, cn=0), gr = 1,
                                              balance [4, 1, .4]; options: focal loss (loss.py:36-61,106-110, fl_gamma > 0,
                                              alpha .25) and autobalance (loss.py:113,158-164) -- pinned by
                                              tests/golden/loss_opts.npz (tools/gen_golden_loss_opts.py runs the reference)
    ciou          core/utils/metrics.py:202-244  bbox_iou(x1y1x2y2=False, CIoU=True)
    seg_loss      core/utils/loss.py:227-243  plain nn.CrossEntropyLoss(ignore_index=-1), mean
    scale_hyp     scripts/train.py:258-260    box*=3/nl, cls*=nc/80*3/nl, obj*=(imgsz/640)^2*3/nl
    step_loss     scripts/train.py:285,356-363  detgain .14, seggain 1; x world_size under DDP

Known reference incompatibility: loss.py:218 clamps an int64 tensor with a float-tensor bound, which
torch >= 1.12 rejects; the golden generator patches that expression to integer bounds, as done here.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=0.7, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0)  # core/hyp/scratch.yaml
BALANCE = (4.0, 1.0, 0.4)
DETGAIN, SEGGAIN = 0.14, 1.0


def scale_hyp(nc: int, imgsz: int, nl: int = 3, hyp=HYP):
    h = dict(hyp)
    h["box"] *= 3.0 / nl
    h["cls"] *= nc / 80.0 * 3.0 / nl
    h["obj"] *= (imgsz / 640) ** 2 * 3.0 / nl
    return h


def ciou(b1, b2, eps=1e-7):
    """b1: (4,n) xywh predicted; b2: (n,4) xywh target."""
    b2 = b2.T
    b1_x1, b1_x2 = b1[0] - b1[2] / 2, b1[0] + b1[2] / 2
    b1_y1, b1_y2 = b1[1] - b1[3] / 2, b1[1] + b1[3] / 2
    b2_x1, b2_x2 = b2[0] - b2[2] / 2, b2[0] + b2[2] / 2
    b2_y1, b2_y2 = b2[1] - b2[3] / 2, b2[1] + b2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def build_targets(p, targets, anchors, anchor_t):
    """loss.py:170-223.  p: list of (bs,na,ny,nx,no); targets (nt,6) [img,cls,x,y,w,h]; anchors (nl,na,2) grid units."""
    na, nt = anchors.shape[1], targets.shape[0]
    tcls, tbox, indices, anch = [], [], [], []
    gain = torch.ones(7)
    ai = torch.arange(na).float().view(na, 1).repeat(1, nt)
    targets = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)   # (na, nt, 7)
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * g
    for i in range(len(p)):
        a_i = anchors[i]
        ny, nx = p[i].shape[2], p[i].shape[3]
        gain[2:6] = torch.tensor([nx, ny, nx, ny]).float()
        t = targets * gain
        if nt:
            r = t[:, :, 4:6] / a_i[:, None]
            j = torch.max(r, 1.0 / r).max(2)[0] < anchor_t
            t = t[j]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % 1.0 < g) & (gxy > 1.0)).T
            l, m = ((gxi % 1.0 < g) & (gxi > 1.0)).T
            j = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[j]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[j]
        else:
            t = targets[0]
            offsets = 0
        b, c = t[:, :2].long().T
        gxy, gwh = t[:, 2:4], t[:, 4:6]
        gij = (gxy - offsets).long()
        gi, gj = gij.T
        a = t[:, 6].long()
        indices.append((b, a, gj.clamp(0, ny - 1), gi.clamp(0, nx - 1)))
        tbox.append(torch.cat((gxy - gij, gwh), 1))
        anch.append(a_i[a])
        tcls.append(c)
    return tcls, tbox, indices, anch


def _bce(x, t, pw, gamma, alpha=0.25):
    """nn.BCEWithLogitsLoss(pos_weight) -- mean -- or, for gamma > 0, FocalLoss wrapped around it (loss.py:36-61): the element-wise
    loss times alpha_factor * (1 - p_t)^gamma, then the mean."""
    if gamma <= 0:
        return F.binary_cross_entropy_with_logits(x, t, pos_weight=pw)
    loss = F.binary_cross_entropy_with_logits(x, t, pos_weight=pw, reduction="none")
    prob = torch.sigmoid(x)
    p_t = t * prob + (1 - t) * (1 - prob)
    alpha_factor = t * alpha + (1 - t) * (1 - alpha)
    return (loss * alpha_factor * (1.0 - p_t) ** gamma).mean()


def det_loss(p, targets, anchors, hyp, nc, balance=None, autobalance=False, ssi=1):
    """Returns (loss * bs, items[lbox, lobj, lcls]) exactly as ComputeLoss.__call__ (loss.py:117-168).
    balance: the per-level objectness weights (a LIST: updated in place when autobalance, loss.py:158-164; ssi = index of the
    stride-16 level, loss.py:113); default the reference's constants for three levels."""
    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    tcls, tbox, indices, anch = build_targets(p, targets, anchors, hyp["anchor_t"])
    pw_cls, pw_obj = torch.tensor([hyp["cls_pw"]]), torch.tensor([hyp["obj_pw"]])
    gamma = float(hyp.get("fl_gamma", 0.0))
    if balance is None:
        balance = list(BALANCE)
    for i, pi in enumerate(p):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros_like(pi[..., 0])
        n = b.shape[0]
        if n:
            ps = pi[b, a, gj, gi]
            pxy = ps[:, :2].sigmoid() * 2.0 - 0.5
            pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anch[i]
            iou = ciou(torch.cat((pxy, pwh), 1).T, tbox[i])
            lbox = lbox + (1.0 - iou).mean()
            tobj[b, a, gj, gi] = iou.detach().clamp(0).type(tobj.dtype)
            if nc > 1:
                t = torch.zeros_like(ps[:, 5:])
                t[range(n), tcls[i]] = 1.0
                lcls = lcls + _bce(ps[:, 5:], t, pw_cls, gamma)
        obji = _bce(pi[..., 4], tobj, pw_obj, gamma)
        lobj = lobj + obji * balance[i]
        if autobalance:
            balance[i] = balance[i] * 0.9999 + 0.0001 / obji.detach().item()
    if autobalance:
        norm = balance[ssi]
        for i in range(len(balance)):
            balance[i] = balance[i] / norm
    lbox, lobj, lcls = lbox * hyp["box"], lobj * hyp["obj"], lcls * hyp["cls"]
    bs = p[0].shape[0]
    return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


def seg_loss(logits, target):
    return F.cross_entropy(logits, target, ignore_index=-1)


def step_loss(p, seg_logits, det_targets, seg_targets, anchors, nc, imgsz, world_size=1):
    """The scalar whose gradient one training micro-step accumulates (train.py:352-367, both backwards summed)."""
    h = scale_hyp(nc, imgsz, len(p))
    dl, items = det_loss(p, det_targets, anchors, h, nc)
    sl = seg_loss(seg_logits, seg_targets)
    total = dl * world_size * DETGAIN + sl * world_size * SEGGAIN
    return total, dl.detach(), items, sl.detach()
