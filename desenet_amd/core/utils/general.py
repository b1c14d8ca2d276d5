"""Mirror of the hot-path pieces of the reference's `core.utils.general`: make_divisible (:411-413), xywh2xyxy (:523-530),
box_iou (:626-648), scale_coords / clip_coords (:598-623) and non_max_suppression (:659-750).

`non_max_suppression` keeps the reference signature and return type (a list of (n_i, 6) tensors [xyxy, conf, cls]) but
runs as three HIP kernels (candidate keys -> per-image bitonic sort -> greedy suppression; csrc/detect_nms.hip) instead of
a Python loop over images around torchvision.ops.nms.  `labels` (auto-labelling apriori boxes, general.py:690-697) are appended
to the predictions as rows before the kernels run; merge-NMS is disabled in the reference (`merge = False`) and absent here.
"""
from __future__ import annotations

import math

import torch

from ... import hip_ops as ops


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """Sinusoidal ramp from y1 to y2 (general.py:421-423): the lr lambda of scripts/train.py:173."""
    return lambda x: ((1 - math.cos(x * math.pi / steps)) / 2) * (y2 - y1) + y1


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, shape):
    if isinstance(boxes, torch.Tensor):
        boxes[:, 0].clamp_(0, shape[1])
        boxes[:, 1].clamp_(0, shape[0])
        boxes[:, 2].clamp_(0, shape[1])
        boxes[:, 3].clamp_(0, shape[0])
    else:
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, shape[1])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


def box_iou(box1, box2):
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (area1[:, None] + area2 - inter)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300):
    """prediction: (bs, n, 5+nc) -> list of (n_i, 6) tensors [x1, y1, x2, y2, conf, cls] per image."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if labels and any(len(l) for l in labels):
        # apriori labels (general.py:690-697, auto-labelling): rows [box, conf 1, one-hot class] appended AFTER each image's own
        # rows (the candidate order the stable score sort preserves).  The kernels take one row count for the batch: images with
        # fewer labels get rows of objectness 0, which the confidence filter (general.py:668,687) drops.
        bs, n, no = prediction.shape
        lmax = max(len(l) for l in labels)
        extra = torch.zeros((bs, lmax, no), dtype=prediction.dtype, device=prediction.device)
        for xi, l in enumerate(labels):
            if len(l):
                l = torch.as_tensor(l, dtype=prediction.dtype, device=prediction.device)
                extra[xi, :len(l), :4] = l[:, 1:5]
                extra[xi, :len(l), 4] = 1.0
                extra[xi, torch.arange(len(l), device=prediction.device), l[:, 0].long() + 5] = 1.0
        prediction = torch.cat((prediction, extra), 1)
    out, cnt = ops.nms(prediction, conf_thres, iou_thres, multi_label, agnostic, classes, max_det)
    counts = cnt.cpu().tolist()      # the one host sync (the reference syncs per image on x.shape[0])
    return [out[i, :c] for i, c in enumerate(counts)]
