#!/usr/bin/env python3
"""Micro-benchmarks of single kernels on DeSeNet-s layer shapes (batch 8, 640x640, bf16): wgrad / conv fwd / dgrad.
Times with torch.cuda events over many back-to-back launches (launch overhead amortised by a graph replay)."""
import sys, time
import torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops

LAYERS = [  # name, N, Ci, H, W, Co, k, s
    ("c3 cv1 128->64 k1 @80", 8, 128, 80, 80, 64, 1, 1),
    ("bneck 64->64 k1 @80", 8, 64, 80, 80, 64, 1, 1),
    ("c3 cv3 128->128 k1 @80", 8, 128, 80, 80, 128, 1, 1),
    ("bneck 128->128 k1 @40", 8, 128, 40, 40, 128, 1, 1),
    ("c3 cv1 256->128 k1 @40", 8, 256, 40, 40, 128, 1, 1),
    ("c3 cv1 512->256 k1 @20", 8, 512, 20, 20, 256, 1, 1),
    ("c3 cv3 512->512 k1 @20", 8, 512, 20, 20, 512, 1, 1),
    ("focus 16->32 k3 @320", 8, 16, 320, 320, 32, 3, 1),
    ("l1 32->64 k3s2 @320", 8, 32, 320, 320, 64, 3, 2),
    ("c3 cv1 64->32 k1 @160", 8, 64, 160, 160, 32, 1, 1),
    ("bneck 32->32 k3 @160", 8, 32, 160, 160, 32, 3, 1),
    ("c3 cv1 128->64 k1 @80", 8, 128, 80, 80, 64, 1, 1),
    ("bneck 64->64 k3 @80", 8, 64, 80, 80, 64, 3, 1),
    ("bneck 128->128 k3 @40", 8, 128, 40, 40, 128, 3, 1),
    ("c3 cv3 256->256 k1 @40", 8, 256, 40, 40, 256, 1, 1),
    ("l7 256->512 k3s2 @40", 8, 256, 40, 40, 512, 3, 2),
    ("bneck 256->256 k3 @20", 8, 256, 20, 20, 256, 3, 1),
    ("ffm 256->128 k3 @80", 8, 256, 80, 80, 128, 3, 1),
]


LAYERS_M = [  # DeSeNet-m (config 5): batch 4, 1280x1280 -- the layers that carry its FLOPs
    ("m bneck 128->128 k3 @160", 4, 128, 160, 160, 128, 3, 1),
    ("m bneck 256->256 k3 @80", 4, 256, 80, 80, 256, 3, 1),
    ("m bneck 512->512 k3 @40", 4, 512, 40, 40, 512, 3, 1),
    ("m bneck 64->64 k3 @320", 4, 64, 320, 320, 64, 3, 1),
    ("m ffm 512->256 k3 @160", 4, 512, 160, 160, 256, 3, 1),
    ("m l3 128->256 k3s2 @320", 4, 128, 320, 320, 256, 3, 2),
    ("m l5 256->512 k3s2 @160", 4, 256, 160, 160, 512, 3, 2),
    ("m c3 cv3 256->256 k1 @160", 4, 256, 160, 160, 256, 1, 1),
    ("m c3 pair 512->512 k1 @80", 4, 512, 80, 80, 512, 1, 1),
    ("m c3 cv1 256->256 k1 @80", 4, 256, 80, 80, 256, 1, 1),
    ("m c3 cv1 128->128 k1 @160", 4, 128, 160, 160, 128, 1, 1),
    ("m rfb 768->128 k1 @160", 4, 768, 160, 160, 128, 1, 1),
    ("m spp 2048->1024 k1 @40", 4, 2048, 40, 40, 1024, 1, 1),
]


LAYERS_PP = [  # conv_pp.hip: the same 400-patch grid (4 x 160 x 160) with K = 576 .. 4608 -> time = a + b K; one-round grids
    ("pp 64->128 k3 @160", 4, 64, 160, 160, 128, 3, 1),
    ("pp 128->128 k3 @160", 4, 128, 160, 160, 128, 3, 1),
    ("pp 256->128 k3 @160", 4, 256, 160, 160, 128, 3, 1),
    ("pp 512->128 k3 @160", 4, 512, 160, 160, 128, 3, 1),
    ("pp1 64->128 k3 @2x160", 2, 64, 160, 160, 128, 3, 1),
    ("pp1 128->128 k3 @2x160", 2, 128, 160, 160, 128, 3, 1),
    ("pp1 256->128 k3 @2x160", 2, 256, 160, 160, 128, 3, 1),
    ("pp1 512->128 k3 @2x160", 2, 512, 160, 160, 128, 3, 1),
    ("pp1 256->256 k3 @2x160", 2, 256, 160, 160, 256, 3, 1),
]

LAYERS_1X1 = [  # the 1x1 layers of config 5 (batch 4, 1280x1280) and the larger ones of config 3 (batch 8, 640x640)
    ("m 128->128 k1 @160", 4, 128, 160, 160, 128, 1, 1),
    ("m 256->256 k1 @80", 4, 256, 80, 80, 256, 1, 1),
    ("m 512->256 k1 @160", 4, 512, 160, 160, 256, 1, 1),
    ("m 256->512 k1 @160", 4, 256, 160, 160, 512, 1, 1),
    ("m 512->512 k1 @80", 4, 512, 80, 80, 512, 1, 1),
    ("m 1024->1024 k1 @40", 4, 1024, 40, 40, 1024, 1, 1),
    ("m 256->256 k1 @160", 4, 256, 160, 160, 256, 1, 1),
    ("m 768->256 k1 @160", 4, 768, 160, 160, 256, 1, 1),
    ("m 512->512 k1 @40", 4, 512, 40, 40, 512, 1, 1),
    ("m 1024->512 k1 @80", 4, 1024, 80, 80, 512, 1, 1),
    ("m 2048->1024 k1 @40", 4, 2048, 40, 40, 1024, 1, 1),
    ("m 512->256 k1 @80", 4, 512, 80, 80, 256, 1, 1),
    ("m 1024->512 k1 @40", 4, 1024, 40, 40, 512, 1, 1),
    ("s 256->256 k1 @40", 8, 256, 40, 40, 256, 1, 1),
    ("s 256->128 k1 @80", 8, 256, 80, 80, 128, 1, 1),
    ("s 384->128 k1 @80", 8, 384, 80, 80, 128, 1, 1),
    ("s 512->512 k1 @20", 8, 512, 20, 20, 512, 1, 1),
    ("s 128->128 k1 @80", 8, 128, 80, 80, 128, 1, 1),
]

_STREAM = None


def timeit(fn, iters=30):
    """us per call under hipGraph replay.  Warm-up and capture run on ONE side stream: hip_ops' workspaces are per stream and
    may not grow during a capture (the warm-up grows them to this layer's size first)."""
    global _STREAM
    if _STREAM is None:
        _STREAM = torch.cuda.Stream()
    _STREAM.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(_STREAM):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=_STREAM):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


BN_SHAPES = [(8, 32, 320, 320), (8, 64, 160, 160), (8, 32, 160, 160), (8, 64, 80, 80), (8, 128, 80, 80), (8, 128, 40, 40),
             (8, 256, 40, 40), (8, 256, 20, 20), (8, 512, 20, 20)]


def bench_bn():
    """BatchNorm + SiLU elementwise passes (training fwd apply, bwd reduce + apply) per activation shape, in-graph time."""
    dt = torch.bfloat16
    for n, c, h, w in BN_SHAPES:
        y = ops.new_act(n, c, h, w, dt, "cuda"); y.normal_()
        dz = ops.new_act(n, c, h, w, dt, "cuda"); dz.normal_()
        z = ops.new_act(n, c, h, w, dt, "cuda")
        dy = ops.new_act(n, c, h, w, dt, "cuda")
        g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
        rm = torch.zeros(c, device="cuda"); rv = torch.ones(c, device="cuda")
        sc, sh, mu, rs = ops.bn_stats(y, g, b, rm, rv, 0.03, 1e-3)
        dg = torch.zeros(c, device="cuda"); db = torch.zeros(c, device="cuda")
        mb = y.numel() * 2 / 1e6
        ops.bn_arena_begin(y.device)
        t_st = timeit(lambda: ops.bn_stats(y, g, b, rm, rv, 0.03, 1e-3))
        t_f = timeit(lambda: ops.bn_act_fwd(y, sc, sh, ops.ACT_SILU, None, z))
        ops.bn_arena_begin(y.device)
        t_b = timeit(lambda: ops.bn_act_bwd(dz, y, sc, sh, mu, rs, ops.ACT_SILU, dy, dg, db, accumulate=True), iters=20)
        print(f"{n}x{c}x{h}x{w} {mb:6.1f} MB | stats {t_st:6.1f} us | fwd {t_f:6.1f} us {2*mb/t_f*1e-3:5.2f} TB/s | "
              f"bwd(reduce+apply) {t_b:6.1f} us {5*mb/t_b*1e-3:5.2f} TB/s", flush=True)


def main():
    dt = torch.bfloat16
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    only = sys.argv[2] if len(sys.argv) > 2 else None
    if which == "bn":
        return bench_bn()
    import os
    for name, n, ci, h, w, co, k, s in {"m": LAYERS_M, "pp": LAYERS_PP, "1x1": LAYERS_1X1}.get(os.environ.get("DSN_BENCH_SET", ""), LAYERS):
        if only and only not in name:
            continue
        pad = k // 2
        ho, wo = ops.conv_out_hw(h, w, k, s, pad, 1)
        x = ops.new_act(n, ci, h, w, dt, "cuda"); x.normal_()
        dy = ops.new_act(n, co, ho, wo, dt, "cuda"); dy.normal_()
        wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wf, wd = ops.pack_weight_fwd(wt, dt), ops.pack_weight_dgrad(wt, dt)
        y = ops.new_act(n, co, ho, wo, dt, "cuda")
        dx = ops.new_act(n, ci, h, w, dt, "cuda")
        g = torch.zeros(co, ci, k, k, device="cuda")
        p = ops.conv_params(k, s, pad, 1)
        flops = 2.0 * n * ho * wo * co * ci * k * k
        byts = (x.numel() + dy.numel()) * 2
        res = {}
        if which in ("all", "fwd"):
            res["fwd"] = timeit(lambda: ops.conv2d_fwd(x, wf, None, None, y, p))
        if which == "fwdbn":     # conv with the BatchNorm partial sums in its epilogue (fp64 atomics into 8 replicas) vs plain
            import ctypes as C
            from desenet_amd import _lib
            L = _lib.lib()
            acc, nbytes = ops.bn_acc(co, "cuda")
            dxx, dyy = ops.desc(x), ops.desc(y)
            res["fwd"] = timeit(lambda: ops.conv2d_fwd(x, wf, None, None, y, p))
            res["fwd+bnacc"] = timeit(lambda: _lib.check(L.dsn_conv2d_fwd_bnacc(C.byref(dxx), wf.data_ptr(), C.byref(dyy), C.byref(p),
                                                                              acc.data_ptr(), nbytes, ops.stream_ptr()), "x"))
        if which in ("all", "dgrad"):
            res["dgrad"] = timeit(lambda: ops.conv2d_dgrad(dy, wd, dx, p))
        if which in ("all", "wgrad"):
            res["wgrad"] = timeit(lambda: ops.conv2d_wgrad(x, dy, g, ci, p, oihw=True))
        if which == "wgrad":   # per-kernel split (gather kernel vs slab reduce) from the library's event profiler
            ops.profile_enable(True)
            for _ in range(5):
                ops.conv2d_wgrad(x, dy, g, ci, p, oihw=True)
            torch.cuda.synchronize()
            prof = ops.profile_collect()
            ops.profile_enable(False)
            res.update({k_: v["ms"] / v["launches"] * 1e3 for k_, v in prof.items()})
        if which == "wgrad":
            print(f"{name:26s} " + "  ".join(f"{k_} {v:6.1f}us" for k_, v in res.items()), flush=True)
            continue
        print(f"{name:26s} GF {flops/1e9:6.2f} MB {byts/1e6:6.1f} | " +
              " | ".join(f"{k_} {v:7.1f} us {flops/v/1e6:6.1f} TF/s {byts/v/1e3:6.0f} GB/s" for k_, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
