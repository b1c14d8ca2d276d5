#!/usr/bin/env python3
"""Soak: 400 graph-replayed training steps of DeSeNet-s (batch 8, bf16, uint8 input, FusedSGD + EMA) on ONE fixed synthetic
batch: the loss must stay finite and fall (the net over-fits the batch), the EMA weights must stay finite."""
import sys, torch
sys.path.insert(0, ".")
import bench, desenet_amd
from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
from desenet_amd.core.utils.torch_utils import ModelEMA
from desenet_amd.graph import GraphedTrainStep
from desenet_amd.optim import FusedSGD
from desenet_amd.parallel import FlatGradients, sgd_param_groups
from desenet_amd.synth import synth_images, synth_targets

dev = torch.device("cuda", 0)
desenet_amd.set_compute_dtype(torch.bfloat16)
m = bench.build_model(dev).train()
m.hyp = scale_hyp(6, 640)
flat = FlatGradients(m.parameters())
opt = FusedSGD(sgd_param_groups(m), lr=0.01, momentum=0.937, nesterov=True)
cl, sl = ComputeLoss(m), SegmentationLosses()
x = (synth_images(8, 640, 3) * 255).round().to(torch.uint8).to(dev)
det_t, seg_t = synth_targets(8, 640, 3)
det_t, seg_t = det_t.to(dev), seg_t.to(dev)
ema = ModelEMA(m)


def lg(det, seg, det_labels, seg_labels):
    out, d_det = cl.forward_backward(det, det_labels, gain=DETGAIN)
    sout, d_seg = sl.forward_backward(seg, seg_labels)
    return (out, sout), d_det, d_seg


step = GraphedTrainStep(m, lg, flat, opt, x, ema=ema, det_targets=det_t, seg_targets=seg_t, max_targets=256)
losses = []
for i in range(400):
    out, sout = step(x, det_t, seg_t)
    if i % 50 == 0 or i == 399:
        losses.append(float(out[0] + sout[0] * SEGGAIN))
        print(f"step {i:4d} loss {losses[-1]:.4f}", flush=True)
assert all(l == l and abs(l) < 1e6 for l in losses), losses
assert losses[-1] < losses[0], losses
assert all(torch.isfinite(v).all() for v in ema.ema.state_dict().values() if v.dtype.is_floating_point)
assert all(torch.isfinite(p).all() for p in m.parameters())
print("soak OK: loss", losses[0], "->", losses[-1], "EMA updates", ema.updates)
