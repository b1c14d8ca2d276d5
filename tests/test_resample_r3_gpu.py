"""Round-3 rewrites of three glue kernels against ATen-CPU (fp32 1e-3 / bf16 2e-2 as tests/test_kernels_gpu.py; indices and max
values exact): the SPP max pools as a cascade of (value, index) pairs (resample.hip: maxpool_cascade_kernel), the separable
bilinear(align_corners=True) backward (bilinear_bwd_rows_kernel) and the row-form adaptive-average-pool backward
(adaptive_avgpool_bwd_rows_kernel) -- at the seg head's real shapes and at ragged ones."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import TOL, q, rnd, to_dev
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from desenet_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(8, 256, 20, 20), (2, 16, 13, 17), (2, 24, 40, 40)])
@pytest.mark.parametrize("levels", [0, 5])
def test_maxpool_cascade_values_and_first_max_indices(ops, shape, dtype, levels):
    """5 / 9 / 13 pools as pool_5 applied 1x / 2x / 3x over (value, flat index) pairs: outputs equal ATen's bit for bit and the
    stored arg-max is ATen's FIRST maximum in row-major window order -- also with heavy ties (inputs quantised to `levels` values)."""
    n, c, h, w = shape
    x = rnd(shape, 91)
    if levels:
        x = torch.round(x * levels) / levels
    xq = q(x, dtype)
    ks = [5, 9, 13]
    xd = to_dev(ops, x, dtype)
    ys = [ops.new_act(n, c, h, w, dtype, "cuda") for _ in ks]
    idxs = [torch.empty((n, h, w, c), dtype=torch.int32, device="cuda") for _ in ks]
    ops.maxpool_s1_multi(xd, ys, ks, idxs)
    for y, ix, k in zip(ys, idxs, ks):
        ref, rix = F.max_pool2d(xq, k, 1, k // 2, return_indices=True)
        assert torch.equal(y.float().cpu(), ref), f"values k={k}"
        assert torch.equal(ix.cpu().permute(0, 3, 1, 2).long(), rix), f"first-max indices k={k}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [((8, 128, 20, 20), (80, 80)), ((8, 128, 40, 40), (80, 80)), ((2, 16, 10, 12), (20, 24)),
                                  ((2, 8, 9, 7), (33, 20)), ((1, 40, 12, 12), (12, 12))])
@pytest.mark.parametrize("accumulate", [False, True])
def test_bilinear_backward_separable_rows(ops, case, dtype, accumulate):
    shape, out = case
    x = rnd(shape, 92)
    xq = q(x, dtype).requires_grad_(True)
    ref = F.interpolate(xq, out, mode="bilinear", align_corners=True)
    gy = rnd(tuple(ref.shape), 93)
    ref.backward(q(gy, dtype))
    base = rnd(shape, 94)
    dx = to_dev(ops, base, dtype) if accumulate else ops.new_act(*shape, dtype, "cuda")
    ops.bilinear_ac_bwd(to_dev(ops, gy, dtype), dx, accumulate=accumulate)
    want = xq.grad + (q(base, dtype) if accumulate else 0)
    assert_close(dx.float().cpu(), want, 2 * TOL[dtype] if accumulate else TOL[dtype], f"bilinear bwd {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ks", [((8, 128, 80, 80), [1, 2, 3, 6]), ((4, 16, 33, 47), [2, 3, 6]), ((8, 32, 80, 80), [1])])
@pytest.mark.parametrize("accumulate", [False, True])
def test_adaptive_avgpool_backward_rows(ops, shape, ks, dtype, accumulate):
    n, c, h, w = shape
    x = rnd(shape, 95)
    xq = q(x, dtype).requires_grad_(True)
    gys = [rnd((n, c, k, k), 96 + k) for k in ks]
    sum(((F.adaptive_avg_pool2d(xq, k) * q(g, dtype)).sum() for k, g in zip(ks, gys))).backward()
    base = rnd(shape, 97)
    dx = to_dev(ops, base, dtype) if accumulate else ops.new_act(*shape, dtype, "cuda")
    ops.adaptive_avgpool_bwd_multi([to_dev(ops, g, dtype) for g in gys], dx, accumulate=accumulate)
    want = xq.grad + (q(base, dtype) if accumulate else 0)
    assert_close(dx.float().cpu(), want, 2 * TOL[dtype] if accumulate else TOL[dtype], f"adaptive pool bwd {shape} {ks}")
