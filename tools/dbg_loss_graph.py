#!/usr/bin/env python3
"""Debug helper: capture only the HIP loss calls in a graph and replay with allocations in between.
usage: dbg_loss_graph.py det|seg|both"""
import sys
import torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops
from desenet_amd.synth import synth_targets

which = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda", 0)
bs, size, nc = 2, 128, 6
det_t, seg_t = synth_targets(bs, size, 21)
det_t, seg_t = det_t.to(dev), seg_t.to(dev)
print("targets", det_t.shape, det_t.dtype, seg_t.shape, seg_t.dtype, flush=True)
torch.manual_seed(0)
p = [torch.randn(bs, 3, size // s, size // s, 5 + nc, device=dev) for s in (8, 16, 32)]
logits = torch.randn(bs, 2, size, size, device=dev)
anchors = [1.25, 1.625, 2.0, 3.75, 4.125, 2.875, 1.875, 3.8125, 3.875, 2.8125, 3.6875, 7.4375, 3.625, 2.8125, 4.875, 6.1875,
           11.65625, 10.1875]


def body():
    outs = []
    if which in ("det", "both"):
        out, dp = ops.det_loss(p, det_t, anchors, [4.0, 1.0, 0.4], 0.05, 1.0, 0.5, 1.0, 1.0, 4.0, 1.0, 0.0, nc, 1.0)
        outs.append(out[0] + sum(d.sum() for d in dp))
    if which in ("seg", "both"):
        sout, dl = ops.seg_ce(logits, seg_t, -1, True)
        outs.append(sout[0] + dl.sum())
    return sum(outs)


print("eager", float(body()), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = body()
torch.cuda.synchronize(); print("captured", flush=True)
keep = []
for i in range(3):
    g.replay()
    torch.cuda.synchronize(); print("replay", i, float(res), flush=True)
    keep.append([torch.empty(n, device=dev).normal_() for n in (10, 1000, 100000, 3000000, 64, 4096)])
    torch.cuda.synchronize(); print("  alloc ok", flush=True)
print("done")
