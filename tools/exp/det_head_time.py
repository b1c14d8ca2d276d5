#!/usr/bin/env python3
"""us per launch of the fused Detect-head forward at DeSeNet-s' shape (8 images, 128/256/512 channels at 80/40/20, 3 x 11 outputs),
all levels and each level alone, graph replay (back-to-back launches: includes the ~2 us launch gap, L2-warm operands)."""
import sys, torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops

dev, dt = "cuda", torch.bfloat16
n, na, no = 8, 3, 11
lv = [(128, 80), (256, 40), (512, 20)]
xs = [ops.new_act(n, c, s, s, dt, dev).normal_() for c, s in lv]
ws = [(torch.randn(na * no, c, device=dev) * 0.05).to(dt).contiguous() for c, _ in lv]
bs = [torch.randn(na * no, device=dev) for _ in lv]
raws = [torch.empty(n, na, s, s, no, device=dev) for _, s in lv]
big = torch.empty(64 << 20, device=dev)         # 256 MB: evicts L2 + MALL between launches when asked


def run(sel, flush):
    def fn():
        if flush:
            big.zero_()
        ops.detect_head_fwd([xs[i] for i in sel], [ws[i] for i in sel], [bs[i] for i in sel], [raws[i] for i in sel], na, no)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


for flush in (False, True):
    base = 0.0
    if flush:
        def z():
            big.zero_()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            z(); g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    z()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            base = e0.elapsed_time(e1) / 20 * 1e3
    for sel in ([0, 1, 2], [0], [1], [2]):
        print(f"flush={flush} levels {sel}: {run(sel, flush) - base:.2f} us per launch", flush=True)
