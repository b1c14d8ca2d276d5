"""Data parallelism for the training step: one process per GPU, ONE flat gradient buffer, ONE RCCL all-reduce per step.

The reference wraps the model in DistributedDataParallel (scripts/train.py:255: 25 MB buckets, per-parameter hooks) and
multiplies the losses by WORLD_SIZE to cancel DDP's averaging (train.py:356-358); the net effect is
    grad = SUM over ranks of d(detgain * det_loss_r + seggain * seg_loss_r) / d(theta).
Here every parameter's `.grad` is a VIEW into one contiguous fp32 buffer (31 MB for DeSeNet-s), autograd accumulates
into the views in place, and `all_reduce()` issues a single `dist.all_reduce(SUM)` on the buffer (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  xGMI is point-to-point (7 links x ~153 GB/s): a 31 MB ring all-reduce at 8
ranks is ~0.35 ms, i.e. latency-, not bandwidth-bound, so one large collective beats bucketing.
The reference's literal DDP path (two backward() calls per forward + grad-less parameters) does not run on current
PyTorch; this defines the mathematically equivalent single reduction.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradients:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat[off:off + k].view(p.shape)
            off += k

    def zero(self):
        """One memset instead of one per parameter; keeps the .grad views attached."""
        if self.flat.is_cuda:
            from . import hip_ops
            hip_ops.zero_(self.flat)              # a library launch (no ATen fill kernel / memset node inside a captured step)
        else:
            self.flat.zero_()
        for p in self.params:          # optimizers / zero_grad(set_to_none=True) may have detached a view
            if p.grad is None or p.grad.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
                raise RuntimeError("a parameter's .grad was replaced; use FlatGradients.zero() instead of zero_grad()")

    def all_reduce(self):
        """SUM over ranks (see module docstring).  No-op for a single process."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def norm(self) -> torch.Tensor:
        return self.flat.norm()


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Initial replica sync (what DDP does when wrapping): parameters and buffers from rank `src`."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def sgd_param_groups(model: torch.nn.Module, weight_decay: float = 5e-4):
    """The reference's three optimizer groups (scripts/train.py:151-166): BN weights (no decay), other weights (decay),
    biases (no decay)."""
    g_bn, g_w, g_b = [], [], []
    for m in model.modules():
        if hasattr(m, "bias") and isinstance(m.bias, torch.nn.Parameter):
            g_b.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d):
            g_bn.append(m.weight)
        elif hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter):
            g_w.append(m.weight)
    return [dict(params=g_bn, weight_decay=0.0), dict(params=g_w, weight_decay=weight_decay),
            dict(params=g_b, weight_decay=0.0)]


def convert_sync_batchnorm(module: torch.nn.Module, process_group=None, force_collectives: bool = False) -> torch.nn.Module:
    """`torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` for the mirrored model (scripts/train.py:218-220): every
    BatchNorm2d computes its training statistics over the GLOBAL batch.  The modules keep their type and `state_dict` (same
    keys as SyncBatchNorm); the exchange is one all-reduce (SUM) of the layer's fp64 sum / sum-of-squares accumulators in the
    forward pass and one of its two backward sums -- the same two collectives per layer as torch's implementation, on raw sums
    instead of (mean, invstd, count) triples.  Per-rank batches must be equal (they are under the reference's DDP split,
    train.py:223).  Returns `module`.  Under desenet_amd.graph.GraphedTrainStep the collectives are CAPTURED into the step's
    hipGraph (backend "nccl" = RCCL only: its all-reduce is a kernel on the capturing stream; gloo's is a host operation and the
    step refuses it).  force_collectives: issue the exchanges even with a single rank (tests of the captured path on one GPU)."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.__dict__["_dsn_sync"] = process_group if process_group is not None else True
            if force_collectives:
                m.__dict__["_dsn_sync_force"] = True
    return module
