"""TEST INFRASTRUCTURE (CPU oracle; never imported by desenet_amd/): the single-image chain of the reference's
scripts/detect.py:134-218 restated with numpy / PyTorch-CPU ops -- letterbox (oracle.letterbox_ref), `/ 255`, the oracle's fused
eval forward (oracle.desenet_ref), non_max_suppression (oracle.nms_ref), segoutput_to_target (plots.py:222-229) and
scale_coords / clip_coords (general.py:598-623) followed by `.round()` (detect.py:218).

Pinning: the network, the NMS pre/post stages, segoutput_to_target and the box arithmetic follow functions the goldens of
tests/golden/{net,nms,metrics}.npz were generated from (tools/gen_golden.py imports the real reference).  Unpinned, as stated in
the headers of the two modules used here: cv2's INTER_LINEAR resize inside letterbox and torchvision's greedy NMS step."""
import numpy as np
import torch
import torch.nn.functional as F

from . import desenet_ref as R
from . import letterbox_ref, nms_ref


def scale_coords(img1_shape, coords, img0_shape):
    """general.py:598-611 (ratio_pad=None) + clip_coords :614-623, numpy, fp32 in place like the reference's tensor ops."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    c = torch.from_numpy(np.array(coords, dtype=np.float32, copy=True))
    c[:, [0, 2]] -= pad[0]
    c[:, [1, 3]] -= pad[1]
    c[:, :4] /= gain
    c[:, 0].clamp_(0, img0_shape[1])
    c[:, 1].clamp_(0, img0_shape[0])
    c[:, 2].clamp_(0, img0_shape[1])
    c[:, 3].clamp_(0, img0_shape[0])
    return c


def segoutput_to_target(output, size=None):
    """plots.py:222-229."""
    size = output[0][0].shape if size is None else size
    output = output.argmax(dim=1, keepdim=True).type(torch.float)
    return F.interpolate(output, size, mode="nearest").squeeze(dim=1)


def postprocess(pred, seg_pred, img_hw, im0_shape, conf_thres=0.25, iou_thres=0.45, max_det=1000):
    """detect.py:190-218 on given network outputs (torch CPU tensors): (det [n, 6] float32 numpy, seg [H, W] float tensor)."""
    det = nms_ref.non_max_suppression(pred.numpy(), conf_thres, iou_thres, max_det=max_det)[0]
    seg = segoutput_to_target(seg_pred, size=tuple(im0_shape[:2]))[0]
    det = np.array(det, dtype=np.float32, copy=True).reshape(-1, 6)
    if len(det):
        det[:, :4] = scale_coords(img_hw, det[:, :4], im0_shape).round().numpy()
    return det, seg


def detect_image(cfg, sd_folded, im0, imgsz=640, conf_thres=0.25, iou_thres=0.45, max_det=1000, stride=32, auto=True):
    """im0: uint8 [H, W, 3] BGR numpy array; sd_folded: oracle.desenet_ref.fold_bn(state_dict) (attempt_load fuses,
    experimental.py:92).  Returns (det, seg, img uint8 [1, 3, h, w], pred, seg_pred)."""
    lb, _, _ = letterbox_ref.letterbox(im0, imgsz, auto=auto, stride=stride)
    img = letterbox_ref.to_network_input(lb)[None]
    x = torch.from_numpy(img).float() / 255.0
    with torch.no_grad():
        (pred, _), seg_pred, _ = R.forward(cfg, sd_folded, x, fused=True)
    det, seg = postprocess(pred, seg_pred, img.shape[2:], im0.shape, conf_thres, iou_thres, max_det)
    return det, seg, img, pred, seg_pred
