#!/usr/bin/env python3
"""us per launch of the fused PyramidPooling kernels at DeSeNet-s' shape (8 images, 128 -> 32 channels, 1/2/3/6 grids), graph replay."""
import sys, torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops

dev, dt = "cuda", torch.bfloat16
n, c, oc = 8, 128, 32
ks = [1, 2, 3, 6] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1].split(",")]
xs = [ops.new_act(n, c, k, k, dt, dev).normal_() for k in ks]
ws = [(torch.randn(oc, c, device=dev) * 0.1).to(dt) for _ in ks]
bns = [None if k == 1 else torch.nn.BatchNorm2d(oc).to(dev) for k in ks]
zs = [ops.new_act(n, oc, k, k, dt, dev) for k in ks]
ys = [ops.new_act(n, oc, k, k, dt, dev) for k in ks]
stats = [None if b is None else torch.empty(4, oc, device=dev) for b in bns]
dys = [ops.new_act(n, oc, k, k, dt, dev).normal_() for k in ks]
dxs = [ops.new_act(n, c, k, k, dt, dev) for k in ks]
dgs = [None if b is None else torch.zeros(oc, device=dev) for b in bns]
dbs = [None if b is None else torch.zeros(oc, device=dev) for b in bns]
dws = [torch.zeros(oc, c, device=dev) for _ in ks]


def fwd():
    ops.pp_stages_fwd(xs, ws, bns, zs, ys, stats, ops.ACT_SILU, 0.03, 1e-3)


def bwd():
    ops.pp_stages_bwd(xs, ws, zs, dys, dxs, stats, dgs, dbs, dws, ops.ACT_SILU, True)


for name, fn in (("fwd", fwd), ("bwd", bwd)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(f"pp_stages_{name} grids {ks}: {best:.2f} us per launch")
