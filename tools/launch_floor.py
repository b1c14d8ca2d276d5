#!/usr/bin/env python3
"""Per-kernel floor on this GPU: N back-to-back launches of a trivial kernel (one 256-thread block writing 4 bytes), eager on
a stream vs replayed from a hipGraph, and a medium elementwise kernel for scale.  DESIGN.md 3 quotes these numbers: a
DeSeNet-s training step at batch 8 is ~430 dependent kernels, so the floor bounds the step from below."""
import sys, time
import torch
sys.path.insert(0, ".")


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    n = 400
    x = torch.zeros(64, device="cuda")
    big = torch.zeros(8 * 128 * 40 * 40, device="cuda", dtype=torch.bfloat16)      # 3.3 MB: a typical 40x40 activation

    def tiny():
        for _ in range(n):
            x.add_(1.0)                   # one block, 64 elements

    def medium():
        for _ in range(n):
            big.add_(1.0)

    y = torch.zeros(64, device="cuda")
    z = torch.zeros(64, device="cuda", dtype=torch.int64)
    big2 = torch.zeros(16 << 20, device="cuda")         # 64 MB: evicts the 32 MB of L2 between tiny kernels

    def mixed():                                      # six DIFFERENT tiny kernels in rotation (instruction-cache cold starts)
        for i in range(n // 6):
            x.add_(1.0); y.mul_(1.5); z.add_(1); x.sub_(y); y.copy_(x); x.clamp_(min=0.0)

    def tiny_after_big():                             # a tiny kernel after a 128 MB streaming kernel, repeatedly
        for _ in range(n // 8):
            big2.add_(1.0); x.add_(1.0)

    def big_only():
        for _ in range(n // 8):
            big2.add_(1.0)

    from desenet_amd import hip_ops as ops
    a8 = ops.new_act(1, 8, 2, 2, torch.bfloat16, "cuda"); b8 = ops.new_act(1, 8, 2, 2, torch.bfloat16, "cuda")
    w8 = ops.pack_weight_fwd(torch.randn(8, 8, 1, 1, device="cuda"), torch.bfloat16)
    p11 = ops.conv_params(1, 1, 0, 1)
    sc = torch.ones(8, device="cuda"); sh = torch.zeros(8, device="cuda")

    def lib_copy():
        for _ in range(n):
            ops.copy(a8, b8)

    def lib_bnact():
        for _ in range(n):
            ops.bn_act_fwd(a8, sc, sh, ops.ACT_SILU, None, b8)

    def lib_conv():
        for _ in range(n):
            ops.conv2d_fwd(a8, w8, None, None, b8, p11)

    def lib_alt():                                    # the step's pattern: two DIFFERENT library kernels alternating, each reading the other's output
        for _ in range(n // 2):
            ops.conv2d_fwd(a8, w8, None, None, b8, p11)
            ops.bn_act_fwd(b8, sc, sh, ops.ACT_SILU, None, a8)

    # the same pattern at a real layer size (8 x 128 x 40 x 40, bf16: 3.3 MB per tensor): 1x1 conv -> BN + SiLU apply -> ...
    A = ops.new_act(8, 128, 40, 40, torch.bfloat16, "cuda"); A.normal_()
    B = ops.new_act(8, 128, 40, 40, torch.bfloat16, "cuda")
    W = ops.pack_weight_fwd(torch.randn(128, 128, 1, 1, device="cuda") * 0.05, torch.bfloat16)
    sc128 = torch.ones(128, device="cuda"); sh128 = torch.zeros(128, device="cuda")

    def layer_alt():
        for _ in range(n // 2):
            ops.conv2d_fwd(A, W, None, None, B, p11)
            ops.bn_act_fwd(B, sc128, sh128, ops.ACT_SILU, None, A)

    def layer_conv_only():
        for _ in range(n // 2):
            ops.conv2d_fwd(A, W, None, None, B, p11)
            ops.conv2d_fwd(B, W, None, None, A, p11)

    def layer_bn_only():
        for _ in range(n // 2):
            ops.bn_act_fwd(A, sc128, sh128, ops.ACT_SILU, None, B)
            ops.bn_act_fwd(B, sc128, sh128, ops.ACT_SILU, None, A)

    for name, body in (("conv <-> BN apply alternating, 1 block", lib_alt), ("1x1 conv <-> BN apply @8x128x40x40", layer_alt),
                       ("1x1 conv only @8x128x40x40", layer_conv_only), ("BN apply only @8x128x40x40", layer_bn_only),
                       ("library copy, 32 elements", lib_copy), ("library BN+SiLU apply, 32 elements", lib_bnact),
                       ("library igemm conv, 1 block", lib_conv), ("tiny (64 floats)", tiny), ("medium (3.3 MB bf16 read+write)", medium),
                       ("six distinct tiny kernels", mixed), ("128 MB stream + tiny (per pair x8)", tiny_after_big),
                       ("128 MB stream alone (x8)", big_only)):
        body(); torch.cuda.synchronize()
        t_eager = timed(body)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        t_graph = timed(g.replay)
        print(f"{name:34s} eager {t_eager / n * 1e3:6.2f} us/kernel   hipGraph replay {t_graph / n * 1e3:6.2f} us/kernel", flush=True)


if __name__ == "__main__":
    main()
