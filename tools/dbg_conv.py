import sys, torch
sys.path.insert(0, '.')
from desenet_amd import hip_ops as ops
import torch.nn.functional as F
torch.manual_seed(0)
def run(n, ci, h, w, co, k=1, ident=False, dtype=torch.float32):
    x = torch.randn(n, ci, h, w)
    wt = torch.randn(co, ci, k, k) * 0.3
    if ident:
        wt.zero_()
        for o in range(min(co, ci)): wt[o, o, k // 2, k // 2] = 1.0
    ref = F.conv2d(x.to(dtype).float(), wt.to(dtype).float(), None, 1, k // 2)
    xd = ops.new_act(n, ci, h, w, dtype, 'cuda'); xd.copy_(x)
    y = ops.new_act(n, co, h, w, dtype, 'cuda')
    ops.conv2d_fwd(xd, ops.pack_weight_fwd(wt.cuda(), dtype), None, None, y, ops.conv_params(k))
    d = (y.float().cpu() - ref).abs()
    print(f"n{n} ci{ci} {h}x{w} co{co} k{k} ident={ident} {dtype}: max err {d.max():.4g} (ref max {ref.abs().max():.3g})")
    if d.max() > 1e-2:
        bad = (d > 1e-2)
        print("  bad frac", bad.float().mean().item(), "bad by channel", bad.sum((0, 2, 3)).tolist()[:40])
        pix = bad.sum(1).flatten()
        print("  bad by pixel (first 64)", pix.tolist()[:64])
for dt in (torch.float32, torch.bfloat16):
    run(1, 16, 4, 4, 16, 1, True, dt)
    run(1, 16, 4, 4, 16, 1, False, dt)
    run(1, 32, 8, 8, 32, 1, False, dt)
    run(2, 16, 9, 7, 24, 1, False, dt)
    run(1, 64, 16, 16, 64, 1, False, dt)
    run(1, 16, 8, 8, 16, 3, False, dt)
