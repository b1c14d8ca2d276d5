"""conv3x3.hip, the halo-tile 3x3 kernel, in its LARGE instantiation (8 x 16 pixels x 128 channels, taken by config 5's maps):
DSN_HALO is read once per process, so the forced-large run happens in a child process and is compared with ATen there."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from desenet_amd import hip_ops as ops
def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(shape, generator=g)
dt = torch.bfloat16
for (n, ci, h, w, co) in [(2, 128, 16, 32, 128), (1, 64, 8, 16, 256), (1, 192, 24, 16, 160)]:
    x, wt = rnd((n, ci, h, w), 1), rnd((co, ci, 3, 3), 2, -0.2, 0.2)
    xq, wq = x.to(dt).float().requires_grad_(True), wt.to(dt).float()
    y = F.conv2d(xq, wq, None, 1, 1)
    gy = rnd(tuple(y.shape), 3)
    y.backward(gy.to(dt).float())
    xd = ops.new_act(n, ci, h, w, dt, "cuda"); xd.copy_(x)
    gd = ops.new_act(n, co, h, w, dt, "cuda"); gd.copy_(gy)
    yd = ops.conv2d_fwd(xd, ops.pack_weight_fwd(wt.cuda(), dt), None, None, ops.new_act(n, co, h, w, dt, "cuda"), ops.conv_params(3))
    dx = ops.conv2d_dgrad(gd, ops.pack_weight_dgrad(wt.cuda(), dt), ops.new_act(n, ci, h, w, dt, "cuda"), ops.conv_params(3))
    e1 = float((yd.float().cpu() - y.detach()).abs().max() / y.detach().abs().max())
    e2 = float((dx.float().cpu() - xq.grad).abs().max() / xq.grad.abs().max())
    assert e1 < 2e-2 and e2 < 2e-2, (n, ci, h, w, co, e1, e2)
print("halo large OK")
"""


def test_large_halo_tile_forward_and_dgrad_vs_aten():
    env = dict(os.environ, DSN_HALO="2")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "halo large OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
