"""Fused BatchNorm-backward sums (dgrad epilogues) against the stand-alone reduction over the dz just written, over a sweep of
shapes and every data-gradient kernel family the default selection picks (tools: profiler labels)."""
import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
import os
if os.environ.get("DBG_WS3"):      # extras variants of the weights-stationary kernels on every eligible shape
    L.dsn_ws_mode(3, 3)
    L.dsn_pp_mode(0); L.dsn_pp1_mode(0)
def fold(a, c): return a.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)
bad = 0
shapes = []
for k in (1, 3):
    for (n, ci, co, h, w) in [(8, 64, 64, 80, 80), (8, 128, 64, 80, 80), (8, 64, 128, 80, 80), (8, 128, 128, 40, 40), (8, 256, 256, 20, 20), (8, 32, 32, 160, 160),
                              (4, 128, 128, 160, 160), (2, 64, 64, 80, 80), (16, 64, 64, 64, 64), (8, 64, 64, 100, 100), (3, 128, 128, 50, 50), (8, 256, 128, 40, 40),
                              (8, 128, 256, 40, 40), (8, 512, 512, 20, 20), (4, 256, 256, 80, 80), (4, 512, 256, 80, 80), (1, 64, 64, 160, 160), (5, 32, 64, 72, 72),
                              (8, 64, 32, 160, 160), (2, 128, 128, 80, 80), (6, 64, 64, 56, 56), (1, 64, 64, 80, 80), (1, 128, 128, 64, 64), (2, 256, 64, 48, 48),
                              (1, 192, 128, 96, 96), (4, 64, 64, 40, 40), (2, 64, 32, 100, 100), (1, 256, 256, 40, 40), (3, 192, 64, 64, 64), (1, 64, 64, 320, 320),
                              (16, 64, 64, 32, 32), (2, 128, 32, 128, 128), (1, 128, 64, 200, 200)]:
        shapes.append((k, n, ci, co, h, w))
for (k, n, ci, co, h, w) in shapes:
    torch.manual_seed(0)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wd = ops.pack_weight_dgrad(wt, dt)
    gd = ops.as_act(torch.randn(n, co, h, w, device="cuda").to(dt))
    yb = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt))
    res = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt))
    st = torch.stack([torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5, torch.randn(ci, device="cuda") * 0.1, torch.rand(ci, device="cuda") + 0.5])
    p = ops.conv_params(k, 1, k // 2, 1)
    for use_res in (False, True):
        dx = ops.new_act(n, ci, h, w, dt, "cuda")
        acc, _ = ops.bn_acc(ci, "cuda")
        red = ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ACT_SILU, acc, ci, 0)])
        ops.profile_enable(True)
        ops.conv2d_dgrad(gd, wd, dx, p, residual=res if use_res else None, red=red)
        torch.cuda.synchronize()
        lab = [kk for kk in ops.profile_collect()]
        ops.profile_enable(False)
        ws, _ = ops.bn_acc(ci, "cuda")
        ops.bn_act_bwd_reduce(dx, yb, st[0], st[1], st[2], st[3], ACT_SILU, ws)
        torch.cuda.synchronize()
        a, s = fold(acc, ci), fold(ws, ci)
        e = float((a - s).abs().max() / s.abs().max())
        flag = "BAD" if e > 1e-4 else "ok"
        bad += e > 1e-4
        print(f"{flag} k{k} {n}x{ci}->{co}@{h}x{w} res={int(use_res)} err {e:.2e}  {lab[0] if lab else ''}", flush=True)
print("bad:", bad)
