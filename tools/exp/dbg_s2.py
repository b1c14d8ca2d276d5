import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
def fold(a, c): return a.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)
for (n, ci, co, h, w) in [(8, 32, 64, 160, 160), (2, 32, 64, 320, 320), (16, 32, 64, 128, 128), (4, 32, 64, 160, 160), (8, 32, 64, 320, 320), (8, 64, 128, 160, 160)]:
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
    bank = ops.WeightBank([conv], [ci], dt, "cuda"); bank.pack()
    s2 = bank.dgrad_s2[0]
    ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
    gd = ops.as_act(torch.randn(n, co, ho, wo, device="cuda").to(dt))
    yb = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt))
    st = torch.stack([torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5, torch.randn(ci, device="cuda") * 0.1, torch.rand(ci, device="cuda") + 0.5])
    p = ops.conv_params(3, 2, 1, 1)
    res = {}
    for md in (2, 0):
        L.dsn_pp_mode(md)
        dx = ops.new_act(n, ci, h, w, dt, "cuda")
        acc, _ = ops.bn_acc(ci, "cuda")
        red = ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ACT_SILU, acc, ci, 0)])
        ops.conv2d_dgrad_s2(gd, s2, dx, p, red=red)
        ws, _ = ops.bn_acc(ci, "cuda")
        ops.bn_act_bwd_reduce(dx, yb, st[0], st[1], st[2], st[3], ACT_SILU, ws)
        torch.cuda.synchronize()
        res[md] = (dx, fold(acc, ci), fold(ws, ci))
    L.dsn_pp_mode(1)
    for md in (2, 0):
        dx, a, s = res[md]
        print((n, ci, co, h, w), "mode", md, "fused vs standalone:", float((a - s).abs().max() / s.abs().max()), "per-channel", ((a - s).abs().max(0).values / s.abs().max()).cpu().numpy().round(4)[:12])
    print("   dx diff", float((res[2][0].float() - res[0][0].float()).abs().max()), "standalone 2 vs 0", float((res[2][2] - res[0][2]).abs().max() / res[0][2].abs().max()))
