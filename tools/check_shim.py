#!/usr/bin/env python3
"""INTEGRATION.md 2, exercised against the REAL reference (build container only): with desenet_amd.shim installed the
reference's own import statements and loaders hand out the mirrored classes.
usage: python tools/check_shim.py <reference-format checkpoint.pt (tools/make_ref_checkpoint.py)>"""
import os
import sys
import unittest.mock as um

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
os.environ["RANK"] = "1"
os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"      # the reference calls torch.load(path) on pickled modules
sys.dont_write_bytecode = True
for name in ["cv2", "torchvision", "torchvision.ops", "seaborn", "imgviz", "thop"]:
    sys.modules[name] = um.MagicMock()
path = os.path.abspath(sys.argv[1])
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import torch  # noqa: E402

import desenet_amd.shim as shim  # noqa: E402
from desenet_amd.core.models import common as MC  # noqa: E402
from desenet_amd.core.models import yolo as MY  # noqa: E402

replaced = shim.install()
assert len(replaced) >= 20

# the scripts' own import lines (train.py:34,48,53; val.py:30; detect.py:27)
from core.models.yolo import Model  # noqa: E402
from core.utils.general import non_max_suppression  # noqa: E402
from core.utils.loss import ComputeLoss, SegmentationLosses  # noqa: E402
from core.utils.torch_utils import ModelEMA, intersect_dicts  # noqa: E402
import desenet_amd.core.utils.general as MG  # noqa: E402
import desenet_amd.core.utils.loss as ML  # noqa: E402

assert Model is MY.Model and non_max_suppression is MG.non_max_suppression
assert ComputeLoss is ML.ComputeLoss and SegmentationLosses is ML.SegmentationLosses
assert ModelEMA.__module__.startswith("desenet_amd.")

# train.py:125-131: checkpoint -> Model(cfg or ckpt['model'].yaml) -> intersect_dicts -> load_state_dict
ckpt = torch.load(path, map_location="cpu")
assert type(ckpt["model"]) is MY.Model, type(ckpt["model"])
model = Model(ckpt["model"].yaml, ch=3, nc=6)
csd = intersect_dicts(ckpt["model"].float().state_dict(), model.state_dict(), exclude=[])
assert len(csd) == len(model.state_dict())
model.load_state_dict(csd, strict=False)

# experimental.py:85-92 (detect.py:84, val.py:166): attempt_load -> .float().fuse().eval() on the unpickled object
from core.models.experimental import attempt_load  # noqa: E402
m = attempt_load(path, map_location="cpu")
assert type(m) is MY.Model and not m.training
assert type(m.model[0]) is MC.Focus and type(m.model[11]) is MC.Upsample and type(m.model[24]) is MY.SegMaskPSP
assert type(m.model[24].out[0].branch1) is MC._ConvBnAct
assert all(not hasattr(c, "bn") for c in m.modules() if type(c) is MC.Conv), "fuse() must have folded every Conv"
assert hasattr(m, "_cat_slot") and m.seg_index == 24 and m.names == [f"c{i}" for i in range(6)]
assert float(m.stride.max()) == 32.0
# no CPU compute path: the adopted model must refuse a CPU batch loudly, not fall back
try:
    m(torch.zeros(1, 3, 64, 64))
except RuntimeError as e:
    assert "no CPU fallback" in str(e)
else:
    raise AssertionError("CPU tensor was accepted")
print("shim OK: reference import lines and loaders resolve to desenet_amd")
