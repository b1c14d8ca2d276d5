#!/usr/bin/env python3
"""Load a checkpoint written by desenet_amd.checkpoint.save_reference_checkpoint with the REAL reference (build container
only, no desenet_amd import on this side except the hash-weight helper) the way its own loaders do (train.py:125-131,
experimental.py:91-92): the pickled 'model' / 'ema' must come back as the reference's classes, carry the expected weights, run
the reference's forward (un-fused and fused) and agree bit for bit with a natively built reference model.
usage: python tools/check_ref_loads_checkpoint.py <ckpt.pt>"""
import os
import sys
import unittest.mock as um

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
os.environ["RANK"] = "1"
sys.dont_write_bytecode = True
for name in ["cv2", "torchvision", "torchvision.ops", "seaborn", "imgviz", "thop"]:
    sys.modules[name] = um.MagicMock()
path = os.path.abspath(sys.argv[1])
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import torch  # noqa: E402
import yaml  # noqa: E402
from core.models import common as C  # noqa: E402  (reference)
from core.models import yolo as Y  # noqa: E402  (reference)

from desenet_amd.synth import synthetic_checkpoint  # noqa: E402

ckpt = torch.load(path, map_location="cpu", weights_only=False)
assert ckpt["epoch"] == 3 and ckpt["updates"] == 17 and abs(ckpt["best_fitness"] - 0.5) < 1e-12
for key in ("model", "ema"):
    m = ckpt[key]
    assert type(m) is Y.Model, type(m)
    assert type(m.model[0]) is C.Focus and type(m.model[2]) is C.C3 and type(m.model[24]) is Y.SegMaskPSP
    assert type(m.model[25]) is Y.Detect and type(m.model[11]) is torch.nn.Upsample
    assert next(m.parameters()).dtype == torch.float16
    assert isinstance(m.yaml, dict) and m.yaml["de_nc"] == 6
opt = ckpt["optimizer"]
assert opt is not None and len(opt["param_groups"]) == 3 and any("momentum_buffer" in s for s in opt["state"].values())

d = yaml.safe_load(open(os.path.join(REF, "core/models/yolov5s_seg.yaml")))
d["se_nc"] = 2
d["head"][-2] = [[16, 19, 22], 1, "SegMaskPSP", ["se_nc", 3, 256, False]]
native = Y.Model(d, ch=3, nc=6)
sd = native.state_dict()
synthetic_checkpoint(sd)
sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
native.load_state_dict(sd)

loaded = ckpt["model"].float()                      # train.py:128 `ckpt['model'].float().state_dict()`
lsd = loaded.state_dict()
assert list(lsd.keys()) == list(sd.keys())
for k in sd:
    if "num_batches_tracked" in k:
        continue
    assert torch.equal(lsd[k], sd[k]), k
x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(4))
native.eval()
loaded.eval()
with torch.no_grad():
    (pa, ra), sa = native(x)
    (pb, rb), sb = loaded(x)
assert torch.equal(pa, pb) and torch.equal(sa, sb) and all(torch.equal(a, b) for a, b in zip(ra, rb))
fused = ckpt["ema"].float().fuse().eval()           # experimental.py:92 `.float().fuse().eval()`
nf = native.fuse().eval()
with torch.no_grad():
    (pa, _), sa = nf(x)
    (pb, _), sb = fused(x)
assert torch.equal(pa, pb) and torch.equal(sa, sb)
print("reference loaded the checkpoint: classes, weights, forward and fused forward identical")
