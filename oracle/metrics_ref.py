"""CPU restatement of the reference's evaluation arithmetic -- TEST INFRASTRUCTURE (only tests/ may import it).

Follows scripts/val.py:101-122 (process_batch), core/utils/metrics.py:247-269 (box_iou), :350-388 (batch_pix_accuracy,
batch_intersection_union).  Pinned by tests/golden/metrics.npz, which tools/gen_golden.py wrote by running the reference's own
functions in the build container."""
import numpy as np
import torch


def box_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """metrics.py:247-269."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    wh = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (a1[:, None] + a2 - inter)


def process_batch(detections: torch.Tensor, labels: torch.Tensor, iouv: torch.Tensor) -> torch.Tensor:
    """val.py:101-122."""
    correct = torch.zeros(detections.shape[0], iouv.shape[0], dtype=torch.bool)
    if detections.shape[0] == 0 or labels.shape[0] == 0:
        return correct
    iou = box_iou(labels[:, 1:], detections[:, :4])
    cand = torch.where((iou >= iouv[0]) & (labels[:, 0:1] == detections[:, 5]))
    if cand[0].shape[0]:
        m = torch.cat((torch.stack(cand, 1), iou[cand[0], cand[1]][:, None]), 1).numpy()
        if cand[0].shape[0] > 1:
            m = m[m[:, 2].argsort()[::-1]]
            m = m[np.unique(m[:, 1], return_index=True)[1]]
            m = m[np.unique(m[:, 0], return_index=True)[1]]
        mt = torch.from_numpy(np.ascontiguousarray(m)).float()
        correct[mt[:, 1].long()] = mt[:, 2:3] >= iouv
    return correct


def seg_counts(output: torch.Tensor, target: torch.Tensor, nclass: int):
    """metrics.py:350-388: (correct, labeled, inter, union) with numpy histograms over range (1, nclass)."""
    predict = torch.max(output, 1)[1].numpy().astype("int32")
    t = target.numpy().astype("int32")
    labeled = int(np.sum(t > 0))
    correct = int(np.sum((predict == t) * (t > 0)))
    nb = nclass - 1
    inter = np.histogram(predict * (predict == t), bins=nb, range=(1, nclass))[0]
    pred = np.histogram(predict, bins=nb, range=(1, nclass))[0]
    lab = np.histogram(t, bins=nb, range=(1, nclass))[0]
    return correct, labeled, inter, pred + lab - inter
