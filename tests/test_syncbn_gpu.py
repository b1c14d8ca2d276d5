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


def _graph_worker(rank, port, path):
    import copy
    import torch.distributed as dist
    import desenet_amd
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.graph import GraphedTrainStep
    from desenet_amd.optim import FusedSGD
    from desenet_amd.parallel import FlatGradients, convert_sync_batchnorm, sgd_param_groups
    from desenet_amd.synth import synth_images, synth_targets, synthetic_checkpoint
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        desenet_amd.set_compute_dtype(torch.float32)
        base = Model("desenet_s.yaml", ch=3, nc=6)
        sd = base.state_dict()
        synthetic_checkpoint(sd)
        base.load_state_dict(sd)
        x = synth_images(2, 128, 31).cuda()
        det_t, seg_t = synth_targets(2, 128, 31)
        det_t, seg_t = det_t.cuda(), seg_t.cuda()
        out = {}
        for tag in ("plain", "sync"):
            m = copy.deepcopy(base).cuda().train()
            m.hyp = scale_hyp(6, 128)
            if tag == "sync":
                convert_sync_batchnorm(m, force_collectives=True)
            flat = FlatGradients(m.parameters())
            opt = FusedSGD(sgd_param_groups(m), lr=0.01, momentum=0.937, nesterov=True)
            cl, sl = ComputeLoss(m), SegmentationLosses()

            def lg(det, seg, dl, sg):
                o, d_det = cl.forward_backward(det, dl, gain=DETGAIN)
                so, d_seg = sl.forward_backward(seg, sg, gain=SEGGAIN)
                return (o, so), d_det, d_seg
            step = GraphedTrainStep(m, lg, flat, opt, x, det_targets=det_t, seg_targets=seg_t, max_targets=32)
            for _ in range(2):
                step(x, det_t, seg_t)
            torch.cuda.synchronize()
            out[tag] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        torch.save(out, os.path.join(path, "sd.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_syncbn_collectives_inside_the_captured_step():
    """GraphedTrainStep with SyncBatchNorm: the per-layer all-reduces of the BatchNorm accumulators are captured into the step's
    hipGraph (RCCL).  One GPU, one rank, collectives forced on: two replayed optimizer steps must leave the model exactly where the
    same steps without SyncBatchNorm leave it (a one-rank SUM is the identity) -- what is exercised is that ~135 captured RCCL
    all-reduces per step replay correctly between the library's kernels.  The two-rank arithmetic is the eager test above."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_graph_worker, args=(port, d), nprocs=1, join=True)
        sd = torch.load(os.path.join(d, "sd.pt"))
    for k, v in sd["plain"].items():
        if v.dtype.is_floating_point:
            # (the synchronised path runs the two-launch conv + BatchNorm kernels, the plain one the deferred-statistics pair:
            #  same arithmetic in another summation order)
            assert torch.allclose(sd["sync"][k], v, rtol=2e-3, atol=1e-4), k
