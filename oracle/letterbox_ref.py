"""CPU restatement of letterbox (core/utils/mixed_datasets.py:722-752) + the loader's transpose / channel flip (:576) --
TEST INFRASTRUCTURE (only tests/ may import it).

PARITY UNPINNED for the resize: `cv2.resize(..., INTER_LINEAR)` is OpenCV (requirements.txt: opencv-python>=4.1.2), a
third-party dependency that is neither under /root/reference nor installed in this image, and the reference holds no test
vectors for it.  `resize_linear_u8` restates OpenCV's published 8-bit algorithm (11-bit fixed-point coefficients,
imgproc/src/resize.cpp: HResizeLinear / VResizeLinear for uchar).  The geometry, the constant border
(cv2.copyMakeBorder, BORDER_CONSTANT) and the channel / layout conversion are plain integer operations restated with numpy."""
import numpy as np


def _taps(dst, src):
    scale = src / dst
    f = ((np.arange(dst) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= src - 1
    s = np.where(lo, 0, np.where(hi, src - 1, s))
    f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
    i1 = np.minimum(s + 1, src - 1)
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, i1, a0, a1


def resize_linear_u8(img, new_w, new_h):
    h0, w0 = img.shape[:2]
    if (h0, w0) == (new_h, new_w):
        return img.copy()
    y0, y1, b0, b1 = _taps(new_h, h0)
    x0, x1, a0, a1 = _taps(new_w, w0)
    s = img.astype(np.int64)
    rows0 = s[y0][:, x0] * a0[None, :, None] + s[y0][:, x1] * a1[None, :, None]
    rows1 = s[y1][:, x0] * a0[None, :, None] + s[y1][:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (rows0 >> 4)) >> 16) + ((b1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    shape = img.shape[:2]
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
    if shape[::-1] != new_unpad:
        img = resize_linear_u8(img, new_unpad[0], new_unpad[1])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    out[:] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, ratio, (dw, dh)


def to_network_input(img_hwc_bgr):
    """mixed_datasets.py:576-577: HWC BGR -> CHW RGB, contiguous."""
    return np.ascontiguousarray(img_hwc_bgr.transpose(2, 0, 1)[::-1])
