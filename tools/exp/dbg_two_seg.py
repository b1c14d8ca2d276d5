import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
def fold(a, c): return a.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)
for (n, ci, co, h, w) in [(3, 384, 152, 24, 58), (3, 384, 160, 24, 58), (2, 256, 192, 40, 40), (3, 384, 128, 24, 58), (3, 128, 152, 24, 58)]:
  for (ppm, wsm) in [(0, 1), (0, 2), (0, 3), (2, 1)]:
    for acts in [(ACT_SILU, ACT_NONE), (ACT_SILU, ACT_SILU), (ACT_NONE, ACT_NONE)]:
        torch.manual_seed(0)
        L.dsn_pp_mode(ppm); L.dsn_pp1_mode(ppm); L.dsn_ws_mode(wsm, -1)
        wt = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
        wd = ops.pack_weight_dgrad(wt, dt)
        gd = ops.as_act(torch.randn(n, co, h, w, device="cuda").to(dt))
        cut = (ci // 16) * 8
        segs, refs = [], []
        for j, (c0, c1) in enumerate([(0, cut), (cut, ci)]):
            c = c1 - c0
            yseg = ops.as_act(torch.randn(n, c, h, w, device="cuda").to(dt))
            st = torch.stack([torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") - 0.5, torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5])
            a, _ = ops.bn_acc(c, "cuda")
            segs.append((c0, c1, yseg, st[0], st[1], st[2], st[3], acts[j], a, c, 0))
            refs.append((yseg, st, a, c, acts[j], c0, c1))
        dx = ops.new_act(n, ci, h, w, dt, "cuda")
        ops.profile_enable(True)
        ops.conv2d_dgrad(gd, wd, dx, ops.conv_params(1, 1, 0, 1), red=ops.bnred(segs))
        torch.cuda.synchronize()
        lab = list(ops.profile_collect()); ops.profile_enable(False)
        errs = []
        for (yseg, st, a, c, act, c0, c1) in refs:
            ws, _ = ops.bn_acc(c, "cuda")
            ops.bn_act_bwd_reduce(dx[:, c0:c1], yseg, st[0], st[1], st[2], st[3], act, ws)
            torch.cuda.synchronize()
            want, got = fold(ws, c), fold(a, c)
            errs.append(float((got - want).abs().max() / want.abs().max()))
        print("BAD" if max(errs) > 1e-4 else "ok ", (n, ci, co, h, w), "pp", ppm, "ws", wsm, "acts", acts, ["%.1e" % e for e in errs], lab[0][:48] if lab else "", flush=True)
L.dsn_ws_mode(1, -1); L.dsn_pp_mode(1); L.dsn_pp1_mode(1)
