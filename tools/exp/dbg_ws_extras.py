import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
L.dsn_pp_mode(0); L.dsn_pp1_mode(0)
def fold(a, c): return a.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)
cases = [(3, 384, 152, 24, 58), (3, 384, 160, 24, 58), (3, 384, 192, 24, 58), (3, 384, 152, 32, 64), (1, 384, 152, 24, 58), (8, 384, 152, 24, 58),
         (3, 128, 152, 24, 58), (3, 384, 256, 24, 58), (2, 256, 192, 40, 40), (3, 384, 136, 24, 58), (3, 64, 192, 24, 58), (3, 384, 192, 25, 57)]
for (n, ci, co, h, w) in cases:
    for wsmode in (1, 2, 3):
        for use_res in (False, True):
            torch.manual_seed(0)
            L.dsn_ws_mode(wsmode, -1)
            wt = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
            wd = ops.pack_weight_dgrad(wt, dt)
            gd = ops.as_act(torch.randn(n, co, h, w, device="cuda").to(dt))
            yb = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt))
            res = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt)) if use_res else None
            st = torch.stack([torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5, torch.randn(ci, device="cuda") * 0.1, torch.rand(ci, device="cuda") + 0.5])
            errs = []
            for rep in range(3):
                dx = ops.new_act(n, ci, h, w, dt, "cuda")
                acc, _ = ops.bn_acc(ci, "cuda")
                ops.profile_enable(True)
                ops.conv2d_dgrad(gd, wd, dx, ops.conv_params(1, 1, 0, 1), residual=res, red=ops.bnred([(0, ci, yb, st[0], st[1], st[2], st[3], ACT_SILU, acc, ci, 0)]))
                torch.cuda.synchronize()
                lab = list(ops.profile_collect()); ops.profile_enable(False)
                ws, _ = ops.bn_acc(ci, "cuda")
                ops.bn_act_bwd_reduce(dx, yb, st[0], st[1], st[2], st[3], ACT_SILU, ws)
                torch.cuda.synchronize()
                a, s = fold(acc, ci), fold(ws, ci)
                errs.append(float((a - s).abs().max() / s.abs().max()))
            flag = "BAD" if max(errs) > 1e-4 else "ok "
            print(flag, (n, ci, co, h, w), "ws", wsmode, "res", int(use_res), ["%.1e" % e for e in errs], lab[0][:50] if lab else "", flush=True)
L.dsn_ws_mode(1, -1); L.dsn_pp_mode(1); L.dsn_pp1_mode(1)
