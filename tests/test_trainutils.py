"""Training-step helpers around the hot path: ModelEMA (one HIP launch), the warm-up block of
scripts/train.py:332-340, one_cycle, the uint8 -> float /255 input conversion folded into Focus, bias gradients.
Goldens (tests/golden/trainutils.npz) come from the reference's own ModelEMA / one_cycle (tools/gen_golden.py trainutils)."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.util import golden


def test_one_cycle_vs_reference_golden():
    from desenet_amd.core.utils.general import one_cycle
    g = golden("trainutils")
    lf = one_cycle(1, 0.2, 300)
    assert np.array_equal(np.array([lf(float(x)) for x in g["one_cycle/x"]]), g["one_cycle/y"])


def test_warmup_schedule_matches_train_py_block():
    """train.py:332-340 restated by hand: np.interp ramps for lr (bias group falls from warmup_bias_lr) and momentum."""
    from desenet_amd.core.utils.general import one_cycle
    from desenet_amd.optim import warmup_schedule
    hyp = dict(lr0=0.01, lrf=0.2, momentum=0.937, warmup_bias_lr=0.1, warmup_momentum=0.8)
    lf = one_cycle(1, hyp["lrf"], 300)
    ps = [nn.Parameter(torch.zeros(1)) for _ in range(3)]
    opt = torch.optim.SGD([ps[0]], lr=hyp["lr0"], momentum=hyp["momentum"], nesterov=True)
    opt.add_param_group({"params": [ps[1]], "weight_decay": 5e-4})
    opt.add_param_group({"params": [ps[2]]})
    for g in opt.param_groups:
        g["initial_lr"] = g["lr"]
    nw = 1000
    for ni, epoch in ((0, 0), (1, 0), (250, 0), (999, 1), (1000, 1)):
        acc = warmup_schedule(opt, ni, nw, hyp, lf, epoch, nbs=64, batch_size=16)
        f = ni / nw
        for j, g in enumerate(opt.param_groups):
            lo = hyp["warmup_bias_lr"] if j == 2 else 0.0
            assert g["lr"] == pytest.approx(lo + f * (hyp["lr0"] * lf(epoch) - lo), rel=1e-12, abs=1e-15)
            assert g["momentum"] == pytest.approx(0.8 + f * (0.937 - 0.8), rel=1e-12)
        assert acc == max(1, round(1 + f * (64 / 16 - 1)))
    before = [dict(lr=g["lr"], momentum=g["momentum"]) for g in opt.param_groups]
    assert warmup_schedule(opt, 1001, nw, hyp, lf, 1, nbs=64, batch_size=16) == 4        # past warm-up: untouched
    assert before == [dict(lr=g["lr"], momentum=g["momentum"]) for g in opt.param_groups]


def _tiny():
    return nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5), nn.Conv2d(5, 7, 1, bias=True))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,updates0", [("cold", 0), ("warm", 5000)])
def test_model_ema_bit_exact_vs_reference(tag, updates0):
    from desenet_amd.core.utils.torch_utils import ModelEMA
    g = golden("trainutils")
    net = _tiny().cuda()
    net.load_state_dict({k: torch.from_numpy(g[f"ema_{tag}/init/{k}"]) for k in net.state_dict()})
    ema = ModelEMA(net, updates=updates0)
    assert not ema.ema.training and not any(p.requires_grad for p in ema.ema.parameters())
    for step in range(3):
        net.load_state_dict({k: torch.from_numpy(g[f"ema_{tag}/model{step}/{k}"]) for k in net.state_dict()})
        ema.update(net)
    assert ema.updates == int(g[f"ema_{tag}/updates"])
    for k, v in ema.ema.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), g[f"ema_{tag}/final/{k}"]), k        # incl. the untouched int64 counter
    ema.update_attr(net, include=["training"])
    assert ema.ema.training == net.training


@pytest.mark.gpu
def test_model_ema_of_the_mirrored_model_in_a_graph_step():
    """deepcopy of the mirrored Model works, and EMA inside GraphedTrainStep == EMA applied eagerly after each step."""
    import desenet_amd
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.core.utils.torch_utils import ModelEMA
    from tests.util import load_cfg
    desenet_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = Model(load_cfg(), ch=3, nc=6).cuda().train()
    ema = ModelEMA(m)
    before = {k: v.clone() for k, v in ema.ema.state_dict().items()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01)
    ema.update(m)
    d = 0.9999 * (1 - math.exp(-1 / 2000))
    for k, v in ema.ema.state_dict().items():
        if v.dtype.is_floating_point:
            want = before[k] * d
            want += (1.0 - d) * m.state_dict()[k]
            assert torch.equal(v, want), k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_focus_u8_folds_the_255_division(dtype):
    from desenet_amd import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (2, 3, 64, 96), generator=g, dtype=torch.uint8).cuda()
    cp = 12 if dtype == torch.float32 else 16
    a = ops.focus_s2d(u8, ops.new_act(2, cp, 32, 48, dtype, u8.device))
    # the yard-stick is the CPU division (IEEE, what the oracle and every golden use); ATen's GPU kernel multiplies by a
    # rounded 1/255 instead and differs from it in the last bit for some pixel values
    b = ops.focus_s2d((u8.cpu().float() / 255.0).cuda(), ops.new_act(2, cp, 32, 48, dtype, u8.device))
    assert torch.equal(a, b)
    odd = torch.randint(0, 256, (1, 5, 8, 10), generator=g, dtype=torch.uint8).cuda()          # generic (non-vector) path
    a = ops.focus_s2d(odd, ops.new_act(1, 20, 4, 5, torch.float32, odd.device))
    b = ops.focus_s2d((odd.cpu().float() / 255.0).cuda(), ops.new_act(1, 20, 4, 5, torch.float32, odd.device))
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_uint8_batch_through_the_whole_net():
    import desenet_amd
    from desenet_amd.core.models.yolo import Model
    from tests.util import load_cfg
    desenet_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = Model(load_cfg(), ch=3, nc=6).cuda().eval()
    u8 = torch.randint(0, 256, (1, 3, 128, 128), dtype=torch.uint8).cuda()
    with torch.no_grad():
        (p0, _), s0 = m(u8)
        (p1, _), s1 = m((u8.cpu().float() / 255.0).cuda())
    assert torch.equal(p0, p1) and torch.equal(s0, s1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,c,ldc", [(torch.float32, 33, 33), (torch.bfloat16, 40, 40), (torch.bfloat16, 33, 40),
                                         (torch.bfloat16, 2, 2)])
def test_channel_sum_bias_gradient(dtype, c, ldc):
    from desenet_amd import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    t = ops.new_act(3, ldc, 20, 24, dtype, "cuda")
    t.copy_(torch.randn(3, ldc, 20, 24, generator=g))
    view = t[:, :c]
    want = view.float().sum((0, 2, 3))
    got = ops.channel_sum(view)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-4)
    slot = torch.full((c,), 2.0, device="cuda")
    ops.channel_sum(view, out=slot, accumulate=True)
    assert torch.allclose(slot, want + 2.0, rtol=1e-5, atol=1e-4)
    assert torch.equal(ops.channel_sum(view), got)               # the accumulators were handed back zeroed
