"""1x1 ping-pong kernel, strongly HBM-bound shapes (A operand 100-420 MB, cold), bit for bit against the kernels replaced."""
import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
torch.manual_seed(0)
tot = bad = 0
for (n, ci, h, w, co, m1) in [(4, 2048, 160, 160, 256, 3), (4, 1024, 160, 160, 128, 2), (4, 512, 320, 320, 128, 2), (4, 2048, 160, 160, 256, 2), (2, 1024, 320, 320, 256, 3)]:
    x = ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt))
    wt = torch.randn(co, ci, 1, 1, device="cuda") * 0.03
    wf = ops.pack_weight_fwd(wt, dt)
    p = ops.conv_params(1, 1, 0, 1)
    ya = ops.new_act(n, co, h, w, dt, "cuda"); yb = ops.new_act(n, co, h, w, dt, "cuda")
    L.dsn_pp1_mode(0)
    ops.conv2d_fwd(x, wf, None, None, yb, p)
    nb = 0
    for rep in range(25):
        L.dsn_pp1_mode(m1)
        ops.conv2d_fwd(x, wf, None, None, ya, p)
        torch.cuda.synchronize()
        tot += 1
        if not torch.equal(ya, yb):
            nb += 1
            d = (ya.float() - yb.float()).abs()
            print("  MISMATCH rep", rep, float(d.max()), int((d > 0).sum()), flush=True)
    bad += nb
    print(f"k1 {ci}->{co} @{n}x{h}x{w} (A {x.numel() * 2 / 1e6:.0f} MB), mode {m1}: {nb} of 25 launches differ", flush=True)
    del x, ya, yb
    torch.cuda.empty_cache()
L.dsn_pp1_mode(1)
print("launches", tot, "bad", bad)
