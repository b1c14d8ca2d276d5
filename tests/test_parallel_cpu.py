"""CPU / gloo, world_size 2: the data-parallel exchange step (desenet_amd.parallel) -- one flat gradient buffer, one
all-reduce(SUM) -- equals the single-process sum of per-rank gradients (the reference's DDP semantics after its
`loss *= WORLD_SIZE`, scripts/train.py:356-358), and parameter broadcast makes replicas identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from desenet_amd.parallel import FlatGradients, broadcast_parameters, sgd_param_groups


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.SiLU(),
                               torch.nn.Conv2d(8, 4, 1))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.rand(2, 3, 8, 8, generator=g), torch.rand(2, 4, 8, 8, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _toy(seed=rank)            # different init per rank: broadcast must fix that
    broadcast_parameters(model)
    flat = FlatGradients(model.parameters())
    opt = torch.optim.SGD(sgd_param_groups(model), lr=0.1, momentum=0.9, nesterov=True)
    for _ in range(2):                  # two steps: zero() must keep the .grad views attached
        flat.zero()
        x, t = _data(rank)
        ((model(x) - t) ** 2).sum().backward()
        flat.all_reduce()
        opt.step()
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    dist.destroy_process_group()


def test_flat_allreduce_equals_sum_of_rank_gradients(tmp_path):
    port, out = _free_port(), str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single-process reference: same two steps with the SUM of both ranks' gradients
    torch.set_num_threads(1)
    model = _toy(seed=0)
    opt = torch.optim.SGD(sgd_param_groups(model), lr=0.1, momentum=0.9, nesterov=True)
    replicas = [_toy(seed=0), _toy(seed=0)]     # per-rank BN batch statistics -> per-rank replicas, shared weights
    for _ in range(2):
        grads = None
        for r, rep in enumerate(replicas):
            rep.load_state_dict(model.state_dict())
            rep.zero_grad()
            x, t = _data(r)
            ((rep(x) - t) ** 2).sum().backward()
            g = [p.grad.clone() for p in rep.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(model.parameters(), grads):
            p.grad = g
        opt.step()
    for (k, v) in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            continue                     # BN statistics stay per-rank (no SyncBN by default, train.py:218-220)
        assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-6), k


def test_param_groups_match_reference_split():
    m = _toy(0)
    g_bn, g_w, g_b = sgd_param_groups(m)
    assert len(g_bn["params"]) == 1 and g_bn["weight_decay"] == 0.0
    assert len(g_w["params"]) == 2 and g_w["weight_decay"] == 5e-4
    assert len(g_b["params"]) == 3 and g_b["weight_decay"] == 0.0
    n = sum(p.numel() for g in (g_bn, g_w, g_b) for p in g["params"])
    assert n == sum(p.numel() for p in m.parameters())
