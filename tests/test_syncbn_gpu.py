"""SyncBatchNorm (scripts/train.py:218-220) across two ranks sharing the test GPU (gloo rendezvous, CUDA tensors): a
Conv+BN+SiLU -> Conv+BN+SiLU stack on the two halves of a batch with synchronised statistics must reproduce the single-process
run on the whole batch -- outputs, input gradients, running statistics, and (after the SUM all-reduce of the gradients the
data-parallel step performs) every parameter gradient.  fp32: 1e-4 on activations / input gradients, 2e-5 of the tensor's
largest entry on parameter gradients (the fp64 accumulators make the statistics exact; what differs is the summation order
of the weight gradients)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp


def _build(seed=3):
    from desenet_amd.core.models.common import Conv
    torch.manual_seed(seed)
    net = torch.nn.Sequential(Conv(8, 16, 3, 1), Conv(16, 8, 1, 1))
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    return net


def _run(net, x):
    x = x.clone().requires_grad_(True)
    z = net(x)
    (z.float() * torch.linspace(0.5, 1.5, z.numel(), device=z.device).view_as(z)).sum().backward()
    return z.detach(), x.grad.detach()


def _worker(rank, world, port, path):
    import torch.distributed as dist
    import desenet_amd
    from desenet_amd.parallel import convert_sync_batchnorm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        desenet_amd.set_compute_dtype(torch.float32)
        torch.cuda.set_device(0)
        g = torch.Generator().manual_seed(7)
        xfull = torch.randn(4, 8, 12, 10, generator=g)
        net = convert_sync_batchnorm(_build()).cuda().train()
        x = xfull[2 * rank:2 * rank + 2].cuda()
        # the loss weights of _run index the FULL batch: give each rank its half of them
        x = x.clone().requires_grad_(True)
        z = net(x)
        wfull = torch.linspace(0.5, 1.5, 4 * z[0].numel(), device="cuda").view(4, *z.shape[1:])
        (z.float() * wfull[2 * rank:2 * rank + 2]).sum().backward()
        grads = [p.grad.clone() for p in net.parameters()]
        for gr in grads:
            dist.all_reduce(gr)                      # the data-parallel step's SUM over ranks
        torch.save(dict(z=z.detach().cpu(), dx=x.grad.cpu(), grads=[t.cpu() for t in grads],
                        rm=[m.running_mean.cpu() for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)],
                        rv=[m.running_var.cpu() for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]),
                   os.path.join(path, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_syncbn_two_ranks_equal_one_rank_on_the_whole_batch():
    import desenet_amd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r = [torch.load(os.path.join(d, f"r{i}.pt")) for i in range(2)]
    desenet_amd.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(7)
    xfull = torch.randn(4, 8, 12, 10, generator=g).cuda()
    net = _build().cuda().train()
    z, dx = _run(net, xfull)
    tol = dict(rtol=1e-4, atol=1e-4)
    assert torch.allclose(torch.cat([r[0]["z"], r[1]["z"]]), z.cpu(), **tol)
    assert torch.allclose(torch.cat([r[0]["dx"], r[1]["dx"]]), dx.cpu(), **tol)
    for i, p in enumerate(net.parameters()):
        ref = p.grad.cpu()         # summation order differs (two half-batch sums vs one): tolerance relative to the tensor's scale
        assert (r[0]["grads"][i] - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-6, i
        assert torch.equal(r[0]["grads"][i], r[1]["grads"][i])
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for i, m in enumerate(bns):
        assert torch.allclose(r[0]["rm"][i], m.running_mean.cpu(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(r[0]["rv"][i], m.running_var.cpu(), rtol=1e-5, atol=1e-6)      # unbiased over the GLOBAL count
