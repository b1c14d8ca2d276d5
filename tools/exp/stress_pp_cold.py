"""Stress: the ping-pong kernels on COLD operands (a pool of inputs far larger than L2 + MALL, rotated), every launch compared bit for
bit with the same launch on the kernels they replace.  Looks for late-landing LDS-DMA stages (counted vmcnt waits assume in-order
completion; tools/exp/oob_order.hip shows that a younger L2-hot / all-out-of-range DMA can retire before an older cold one)."""
import sys, torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
torch.manual_seed(0)
bad = tot = 0
for (k, n, ci, h, w, co, pool) in [(1, 4, 512, 80, 80, 512, 12), (1, 4, 512, 160, 160, 256, 6), (1, 4, 1024, 40, 40, 1024, 24), (3, 4, 256, 80, 80, 256, 16),
                                   (3, 4, 128, 160, 160, 128, 10), (3, 4, 512, 160, 160, 256, 5), (3, 8, 256, 80, 80, 128, 10)]:
    xs = [ops.as_act(torch.randn(n, ci, h, w, device="cuda").to(dt)) for _ in range(pool)]
    wts = [torch.randn(co, ci, k, k, device="cuda") * 0.05 for _ in range(4)]
    wfs = [ops.pack_weight_fwd(wt, dt) for wt in wts]
    p = ops.conv_params(k, 1, k // 2, 1)
    junk = torch.empty(512 * 1024 * 1024 // 2, dtype=dt, device="cuda")     # 512 MB: flushes L2 / MALL between rounds
    ya = ops.new_act(n, co, h, w, dt, "cuda"); yb = ops.new_act(n, co, h, w, dt, "cuda")
    nb = 0
    for rep in range(40):
        i, j = rep % pool, rep % 4
        L.dsn_pp_mode(0); L.dsn_pp1_mode(0)
        ops.conv2d_fwd(xs[i], wfs[j], None, None, yb, p)
        junk.fill_(rep)                       # evict
        L.dsn_pp_mode(2 if k == 3 else 1); L.dsn_pp1_mode(3 if co % 256 == 0 else 2)
        ops.conv2d_fwd(xs[i], wfs[j], None, None, ya, p)
        torch.cuda.synchronize()
        tot += 1
        if not torch.equal(ya, yb):
            nb += 1
            d = (ya.float() - yb.float()).abs()
            print("  MISMATCH rep", rep, "max diff", float(d.max()), "elements", int((d > 0).sum()), flush=True)
    bad += nb
    print(f"k{k} {ci}->{co} @{n}x{h}x{w}: {nb} of 40 launches differ", flush=True)
    del xs, junk
    torch.cuda.empty_cache()
L.dsn_pp_mode(1); L.dsn_pp1_mode(1)
print("launches", tot, "bad", bad)
