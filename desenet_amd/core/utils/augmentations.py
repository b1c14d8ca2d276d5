"""The one piece of the reference's input pipeline that sits right in front of the network: `letterbox`
(core/utils/mixed_datasets.py:722-752, the same function as core/utils/datasets.py:618) and the HWC-BGR -> CHW-RGB conversion
the loader applies next (mixed_datasets.py:576).  The geometry (ratio, unpadded size, padding, border rounding) is the
reference's host arithmetic verbatim; the pixels are produced on the MI355X by one kernel (`dsn_letterbox_u8`).  The
random augmentations of the loader (mosaic, perspective, HSV) stay with the reference (out of scope)."""
from __future__ import annotations

import numpy as np
import torch

from ... import hip_ops as ops


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """-> (ratio (w, h), new_unpad (w, h), (dw, dh), (top, bottom, left, right)) exactly as mixed_datasets.py:724-750."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = (r, r)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = (new_shape[1] / shape[1], new_shape[0] / shape[0])
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return ratio, new_unpad, (dw, dh), (top, bottom, left, right)


def letterbox(img: torch.Tensor, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True,
              stride=32, to_chw_rgb=False):
    """Reference signature and return value (img, ratio, (dw, dh)); `img` is a uint8 [H, W, 3] tensor on the MI355X.
    to_chw_rgb=True additionally folds `img.transpose(2, 0, 1)[::-1]` in and returns the [3, H, W] network input."""
    shape = tuple(img.shape[:2])
    ratio, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry(shape, new_shape, auto, scaleFill, scaleup,
                                                                               stride)
    out_hw = (new_unpad[1] + top + bottom, new_unpad[0] + left + right)
    out = ops.letterbox_u8(img, out_hw, (new_unpad[1], new_unpad[0]), top, left, color, chw_reversed=to_chw_rgb)
    return out, ratio, (dw, dh)
