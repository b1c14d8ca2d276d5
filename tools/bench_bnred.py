#!/usr/bin/env python3
"""Cost of forming the BatchNorm backward sums in the dgrad epilogue (dsn_conv2d_dgrad_bnred) against the stand-alone reduction:
per layer shape (DeSeNet-s, batch 8, bf16) the time of  dgrad,  dgrad + sums,  reduce launch  -- each as 20 launches inside one
hipGraph, best of 5 replays."""
import sys
import torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU

LAYERS = [  # k, ci (dx channels), co (dy channels), hw
    (1, 64, 64, 160), (3, 32, 32, 160), (1, 128, 128, 80), (1, 64, 64, 80), (3, 64, 64, 80), (1, 256, 256, 40), (1, 128, 128, 40),
    (3, 128, 128, 40), (1, 512, 512, 20), (1, 256, 256, 20), (3, 256, 256, 20), (1, 256, 128, 40), (1, 512, 256, 20),
]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


def main():
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    n = 8
    print(f"{'layer':28s} {'dgrad':>8s} {'dgrad+sums':>11s} {'reduce':>8s}   saved per layer (us)")
    tot = 0.0
    for k, ci, co, hw in LAYERS:
        w = ops.pack_weight_dgrad(torch.randn(co, ci, k, k, device="cuda") * 0.05, dt)
        dy = ops.new_act(n, co, hw, hw, dt, "cuda"); dy.normal_()
        dx = ops.new_act(n, ci, hw, hw, dt, "cuda")
        y = ops.new_act(n, ci, hw, hw, dt, "cuda"); y.normal_()
        st = torch.rand(4, ci, device="cuda") + 0.5
        p = ops.conv_params(k, 1, k // 2, 1)
        acc = torch.zeros(ops._lib.lib().dsn_bn_workspace_bytes(ci), dtype=torch.uint8, device="cuda")
        red = ops.bnred([(0, ci, y, st[0], st[1], st[2], st[3], ACT_SILU, acc, ci, 0)])
        t0 = timed(lambda: ops.conv2d_dgrad(dy, w, dx, p))
        t1 = timed(lambda: ops.conv2d_dgrad(dy, w, dx, p, red=red))
        t2 = timed(lambda: ops.bn_act_bwd_reduce(dx, y, st[0], st[1], st[2], st[3], ACT_SILU, acc))
        tot += t0 + t2 - t1
        print(f"{k}x{k} {co:4d} -> {ci:4d} @{hw:<4d}         {t0:8.1f} {t1:11.1f} {t2:8.1f}   {t0 + t2 - t1:6.1f}", flush=True)
    print(f"sum of savings over these layers: {tot:.1f} us")


if __name__ == "__main__":
    main()
