#!/usr/bin/env python3
"""Single-layer weight-gradient time per 3x3 layer shape (bf16), eager launches timed with events (kernels of 50-300 us: the
launch overhead does not matter).  Run under DSN_WGRAD_HALO=0 / 1 to compare the all-taps and halo-tile kernels."""
import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import hip_ops as ops

SHAPES = [  # n, ci, h, w, co, dil
    (8, 16, 320, 320, 32, 1), (8, 32, 160, 160, 32, 1), (8, 64, 80, 80, 64, 1), (8, 256, 80, 80, 128, 1), (8, 128, 80, 80, 128, 2),
    (8, 128, 80, 80, 128, 3), (4, 128, 160, 160, 128, 1), (4, 256, 80, 80, 256, 1), (4, 64, 320, 320, 64, 1),
]


def main():
    dt = torch.bfloat16
    desenet_amd.set_compute_dtype(dt)
    for n, ci, h, w, co, d in SHAPES:
        x = ops.new_act(n, ci, h, w, dt, "cuda"); x.normal_()
        dy = ops.new_act(n, co, h, w, dt, "cuda"); dy.normal_()
        g = torch.zeros(co, ci, 3, 3, device="cuda")
        p = ops.conv_params(3, 1, d, d)
        for _ in range(3):
            ops.conv2d_wgrad(x, dy, g, ci, p, oihw=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_wgrad(x, dy, g, ci, p, oihw=True)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        fl = 2.0 * n * h * w * co * ci * 9
        print(f"{n}x{ci}->{co} @{h} d{d}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
