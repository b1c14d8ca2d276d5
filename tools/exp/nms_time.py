"""NMS stage times at config 2's shape (16 x 25200 rows, conf 0.25, nc 6) with every row a candidate and with ~2 % candidates."""
import sys, torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops
import numpy as np
for frac in (1.0, 0.02):
    rng = np.random.RandomState(0)
    p = np.zeros((16, 25200, 11), np.float32)
    p[..., 0:2] = rng.uniform(0, 640, (16, 25200, 2)); p[..., 2:4] = rng.uniform(4, 200, (16, 25200, 2))
    p[..., 4] = np.where(rng.uniform(0, 1, (16, 25200)) < frac, 0.9, 0.01); p[..., 5:] = rng.uniform(0.5, 1, (16, 25200, 6))
    x = torch.from_numpy(p).cuda()
    for _ in range(3): ops.nms(x, 0.25, 0.45, max_det=1000)
    ops.profile_enable(True)
    for _ in range(10): ops.nms(x, 0.25, 0.45, max_det=1000)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    ops.profile_enable(False)
    t0.record()
    for _ in range(20): ops.nms(x, 0.25, 0.45, max_det=1000)
    t1.record(); torch.cuda.synchronize()
    print(f"candidates {frac:.0%}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us per batch of 16", flush=True)
