import json, sys, torch
sys.path.insert(0, ".")
import bench, desenet_amd
from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
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
def lg(det, seg, dl, sg):
    out, d_det = cl.forward_backward(det, dl, gain=DETGAIN)
    sout, d_seg = sl.forward_backward(seg, sg)
    return (out, sout), d_det, d_seg
step = GraphedTrainStep(m, lg, flat, opt, x, det_targets=det_t, seg_targets=seg_t, max_targets=256)
losses = []
for i in range(12):
    out, sout = step(x, det_t, seg_t)
    losses.append(round(float(out[0] + sout[0] * SEGGAIN), 4))
print("LOSSES " + json.dumps(losses))
