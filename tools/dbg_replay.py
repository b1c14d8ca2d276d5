#!/usr/bin/env python3
"""Debug helper: capture the training step of a small model and replay it several times with explicit synchronisation
points (prints which stage completed).  usage: dbg_replay.py [size] [batch] [dtype fp32|bf16] [replays] [alloc_between 0|1]"""
import sys, copy
import torch
sys.path.insert(0, ".")
import bench, desenet_amd
from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
from desenet_amd.graph import GraphedTrainStep
from desenet_amd.parallel import FlatGradients, sgd_param_groups
from desenet_amd.synth import synth_images, synth_targets

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "fp32"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
alloc = int(sys.argv[5]) if len(sys.argv) > 5 else 1
desenet_amd.set_compute_dtype(dt)
dev = torch.device("cuda", 0)
m = bench.build_model(dev).train()
m.hyp = scale_hyp(6, size)
flat = FlatGradients(m.parameters())
opt = torch.optim.SGD(sgd_param_groups(m), lr=0.01, momentum=0.937, nesterov=True)
cl, sl = ComputeLoss(m), SegmentationLosses()
x = synth_images(bs, size, 21).to(dev)
det_t, seg_t = synth_targets(bs, size, 21)
det_t, seg_t = det_t.to(dev), seg_t.to(dev)


def lg(det, seg):
    out, d_det = cl.forward_backward(det, det_t, gain=DETGAIN)
    sout, d_seg = sl.forward_backward(seg, seg_t)
    return out[0] + sout[0] * SEGGAIN, d_det, d_seg


import os
from desenet_amd.runtime import Tape
stage = os.environ.get("DBG_STAGE", "all")     # fwd | loss | bwd | all
if stage != "all":
    def body(self):
        self.flat.zero()
        tape = Tape()
        det, seg = self.model.fwd(self.x, tape)
        if stage == "fwd":
            return det[0].sum()
        loss, d_det, d_seg = self.loss_and_grads(det, seg)
        if stage == "loss":
            return loss
        tape.begin_backward()
        self.model.bwd(tape, (d_det, d_seg), need_dx=False)
        tape.join()
        return loss
    GraphedTrainStep._body = body
    class NoOpt:
        def step(self): pass
    opt = NoOpt()
step = GraphedTrainStep(m, lg, flat, opt, x, warmup=3)
torch.cuda.synchronize(); print("captured", flush=True)
keep = []
for i in range(reps):
    loss = step()
    torch.cuda.synchronize(); print(f"replay {i} ok loss {float(loss):.5f}", flush=True)
    if alloc:
        keep.append([p.detach().clone() for p in m.parameters()])
        torch.cuda.synchronize(); print(f"  clones {i} ok", flush=True)
print("done")
