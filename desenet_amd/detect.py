"""The single-image chain of the reference's `scripts/detect.py` (BASELINE.json config 1), device end to end:

    im0 (HWC BGR uint8, what cv2.imread / LoadImages yields, mixed_datasets.py:209-223)
      -> letterbox(im0, imgsz, stride, auto) + HWC-BGR -> CHW-RGB           detect.py:134 / mixed_datasets.py:220-224
      -> `/ 255` (folded into Focus) -> fused eval model                      detect.py:147-158
      -> non_max_suppression(pred, conf, iou, classes, agnostic, max_det)    detect.py:190
      -> segoutput_to_target(seg, size=im0.shape[:2])                        detect.py:198 / plots.py:222-229
      -> scale_coords(img.shape[2:], det[:, :4], im0.shape).round()          detect.py:218

Everything between "im0 on the device" and "boxes + class map on the device" is HIP kernels (dsn_letterbox_u8, the network,
dsn_nms, dsn_seg_argmax_nearest); the box rescaling is the reference's host arithmetic on n <= max_det rows.  Drawing, file I/O,
the ONNX / TensorFlow branches and the second-stage classifier of detect.py are out of scope.
"""
from __future__ import annotations

import torch

from .core.utils.augmentations import letterbox
from .core.utils.general import non_max_suppression, scale_coords
from .core.utils.metrics import segoutput_to_target


@torch.no_grad()
def detect_image(model, im0, imgsz=640, conf_thres=0.25, iou_thres=0.45, max_det=1000, classes=None, agnostic_nms=False,
                 stride=32, auto=True):
    """One image through detect.py's loop body.  `model`: an eval-mode (normally `.fuse()`d, experimental.py:92) mirrored Model
    on the MI355X; `im0`: uint8 [H, W, 3] BGR (torch tensor or numpy array).  Returns (det [n, 6] = boxes in im0 pixels (rounded),
    conf, cls; seg [H, W] class map as float, plots.py:229; img: the uint8 [1, 3, h, w] network input, for callers that draw)."""
    dev = next(model.parameters()).device
    if not torch.is_tensor(im0):
        im0 = torch.from_numpy(im0)
    im0 = im0.to(dev)
    if model.training:
        raise ValueError("detect_image runs the eval-mode model: call model.eval() (and .fuse()) first")
    img, _, _ = letterbox(im0, imgsz, stride=stride, auto=auto, to_chw_rgb=True)
    img = img[None]                                                   # detect.py:150-151 (uint8: `/ 255` happens in Focus)
    (pred, _), seg_pred = model(img)
    det = non_max_suppression(pred, conf_thres, iou_thres, classes, agnostic_nms, max_det=max_det)[0]
    seg = segoutput_to_target(seg_pred, size=tuple(im0.shape[:2]))[0]
    if len(det):
        det = det.clone()
        det[:, :4] = scale_coords(img.shape[2:], det[:, :4], im0.shape).round()
    return det, seg, img
