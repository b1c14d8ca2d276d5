#!/usr/bin/env python3
"""tests/golden/loss_opts.npz: the reference's ComputeLoss (core/utils/loss.py:91-168) with the options the hot-path configuration
leaves off -- focal loss (fl_gamma > 0, loss.py:36-61,106-110) and autobalance (loss.py:113,158-164) -- run on seeded raw
predictions by IMPORTING THE REAL REFERENCE (same recipe and the same one-expression patch of loss.py:218 as tools/gen_golden.py).
Data only: inputs, losses, gradients, and the balance list after each of three consecutive calls.

    python tools/gen_golden_loss_opts.py
"""
import os
import sys
import unittest.mock as um

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
os.environ["RANK"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
for name in ["cv2", "torchvision", "torchvision.ops", "seaborn", "imgviz", "thop"]:
    sys.modules[name] = um.MagicMock()
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from core.models import common as _C  # noqa: E402,F401  (reference; import order as in tools/gen_golden.py: breaks a module cycle)
from core.models import yolo as _Y  # noqa: E402,F401
from core.utils import general as _G  # noqa: E402,F401
from core.utils import loss as L  # noqa: E402  (reference)

from desenet_amd.synth import synth_targets  # noqa: E402

ANCHORS = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float().view(3, 3, 2) \
    / torch.tensor([8., 16., 32.]).view(3, 1, 1)


def patch_loss():
    orig = torch.Tensor.clamp_

    def clamp_(self, lo=None, hi=None):
        if torch.is_tensor(hi) and not self.dtype.is_floating_point:
            hi = int(hi)
        if torch.is_tensor(lo) and not self.dtype.is_floating_point:
            lo = int(lo)
        return orig(self, lo, hi)

    torch.Tensor.clamp_ = clamp_


class Det:
    na, nc, nl, anchors = 3, 6, 3, ANCHORS
    stride = torch.tensor([8., 16., 32.])


class Model(torch.nn.Module):
    def __init__(self, hyp):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.model = [Det()]
        self.hyp = hyp


def main():
    patch_loss()
    hyp0 = yaml.safe_load(open(os.path.join(REF, "core/hyp/scratch.yaml")))
    s = {}
    for tag, bs, size, seed, gamma, auto, pw in [("focal", 2, 128, 31, 1.5, False, 1.0), ("focal_pw", 2, 128, 32, 2.0, False, 1.7),
                                                 ("auto", 2, 128, 33, 0.0, True, 1.0), ("focal_auto", 3, 64, 34, 1.5, True, 1.0)]:
        h = dict(hyp0)
        h["box"] *= 1.0
        h["cls"] *= 6 / 80.0
        h["obj"] *= (size / 640) ** 2
        h["fl_gamma"] = gamma
        h["cls_pw"], h["obj_pw"] = pw, pw
        h["label_smoothing"] = 0.0
        cl = L.ComputeLoss(Model(h), autobalance=auto)
        det_t, _ = synth_targets(bs, size, seed, boxes_per_image=12)
        s[f"{tag}/targets"] = det_t.numpy()
        s[f"{tag}/hyp"] = np.array([h["box"], h["obj"], h["cls"], h["cls_pw"], h["obj_pw"], h["anchor_t"], gamma, float(auto)], np.float64)
        for step in range(3):
            g = torch.Generator().manual_seed(seed * 10 + step)
            p = [(torch.randn(bs, 3, size // st, size // st, 11, generator=g) * 2).requires_grad_(True) for st in (8, 16, 32)]
            loss, items = cl(p, det_t)
            loss.sum().backward()
            s[f"{tag}/{step}/loss"] = loss.detach().numpy()
            s[f"{tag}/{step}/items"] = items.numpy()
            s[f"{tag}/{step}/balance"] = np.array([float(b) for b in cl.balance], np.float64)
            for i, t in enumerate(p):
                s[f"{tag}/{step}/p{i}"] = t.detach().numpy()
                s[f"{tag}/{step}/dp{i}"] = t.grad.numpy()
            print(tag, step, float(loss), items.tolist(), [float(b) for b in cl.balance])
    out = os.path.join(REPO, "tests", "golden", "loss_opts.npz")
    np.savez_compressed(out, **s)
    print(out, len(s), "arrays", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
