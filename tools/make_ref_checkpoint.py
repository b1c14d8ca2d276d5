#!/usr/bin/env python3
"""Write a checkpoint with the REAL reference (build container only) in its train.py format -- the fixture for
tests/test_checkpoint_cpu.py, generated on the fly (a pickled 31 MB model is not committed).
usage: python tools/make_ref_checkpoint.py <out.pt>"""
import os
import sys
import unittest.mock as um

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
os.environ["RANK"] = "1"
sys.dont_write_bytecode = True
for name in ["cv2", "torchvision", "torchvision.ops", "seaborn", "imgviz", "thop"]:
    sys.modules[name] = um.MagicMock()
out = os.path.abspath(sys.argv[1])
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)
from copy import deepcopy  # noqa: E402

import torch  # noqa: E402
import yaml  # noqa: E402
from core.models import yolo as Y  # noqa: E402  (reference)

from desenet_amd.synth import synthetic_checkpoint  # noqa: E402

d = yaml.safe_load(open(os.path.join(REF, "core/models/yolov5s_seg.yaml")))
d["se_nc"] = 2
d["head"][-2] = [[16, 19, 22], 1, "SegMaskPSP", ["se_nc", 3, 256, False]]
m = Y.Model(d, ch=3, nc=6)
sd = m.state_dict()
synthetic_checkpoint(sd)
m.load_state_dict(sd)
m.names = [f"c{i}" for i in range(6)]
ckpt = {"epoch": 7, "best_fitness": 0.25, "model": deepcopy(m).half(), "ema": deepcopy(m).half(), "updates": 123,
        "optimizer": None, "wandb_id": None}
torch.save(ckpt, out)
print("wrote", out)
