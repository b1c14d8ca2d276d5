"""SGD with momentum / Nesterov / weight decay as ONE HIP launch per step (`dsn_sgd_step`).

Drop-in for the reference's `optim.SGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)` with its three
parameter groups (scripts/train.py:159-166): same constructor arguments, same `param_groups` / `state_dict` layout
(`momentum_buffer` per parameter), same update as torch.optim.SGD.  torch's foreach implementation issues ~23 launches per
step for DeSeNet-s (0.22 ms at batch 8); this one walks every parameter with one kernel driven by a device-side table.

Hyper-parameters live in a small DEVICE tensor the kernel reads, so a captured hipGraph follows `lr` / `momentum` changes
made by a scheduler: `step()` re-uploads them whenever the Python-side values changed (outside capture), and
`sync_hyper()` does the same explicitly between graph replays.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .hip_ops import stream_ptr


def warmup_schedule(optimizer, ni: int, nw: int, hyp: dict, lf, epoch: int, nbs: int = 64, batch_size: int = 64) -> int:
    """The warm-up block of scripts/train.py:332-340, verbatim arithmetic: for integrated batch index ni <= nw, group 2
    (biases) falls from hyp['warmup_bias_lr'] to lr0*lf(epoch) while the other groups rise from 0, momentum ramps from
    hyp['warmup_momentum'] to hyp['momentum'].  Returns `accumulate` (train.py:337).  With FusedSGD the new values reach the
    device on the next step()/sync_hyper() -- a captured graph follows them."""
    import numpy as np
    accumulate = max(round(nbs / batch_size), 1)
    if ni <= nw:
        xi = [0, nw]
        accumulate = max(1, np.interp(ni, xi, [1, nbs / batch_size]).round())
        for j, x in enumerate(optimizer.param_groups):
            x["lr"] = np.interp(ni, xi, [hyp["warmup_bias_lr"] if j == 2 else 0.0, x["initial_lr"] * lf(epoch)])
            if "momentum" in x:
                x["momentum"] = np.interp(ni, xi, [hyp["warmup_momentum"], hyp["momentum"]])
    return int(accumulate)


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid SGD hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self._table = None
        self._uploaded = None
        # `first` (kernel: buf = g' instead of mom*buf + (1-damp)*g') is only ever needed with dampening != 0: a momentum
        # buffer created as zeros gives mom*0 + (1-0)*g' = g', torch.optim.SGD's first step, without any flag -- so buffers
        # restored by load_state_dict (resume: train.py `optimizer.load_state_dict(ckpt['optimizer'])`) and buffers created
        # later (add_param_group) each do the right thing per parameter.  Decided per table build in _build().
        self._first = False

    # ---- device-side tables ---------------------------------------------------------------------------------------------
    def _build(self):
        L = _lib.lib()
        chunk = L.dsn_sgd_chunk()
        entries = []
        dev = None
        fresh = existing = 0
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() \
                        or not p.grad.is_contiguous() or not p.is_cuda:
                    raise TypeError("FusedSGD updates contiguous fp32 CUDA parameters with fp32 gradients")
                dev = p.device
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    fresh += 1
                else:
                    existing += 1
                    b = st["momentum_buffer"]
                    if b.dtype != torch.float32 or not b.is_contiguous() or b.device != p.device:
                        st["momentum_buffer"] = b.to(device=p.device, dtype=torch.float32).contiguous()
                entries.append((p, p.grad, st["momentum_buffer"], gi))
        if not entries:
            return None
        self._first = False
        if any(float(g["dampening"]) != 0.0 and float(g["momentum"]) != 0.0 for g in self.param_groups):
            if fresh and existing:
                raise NotImplementedError("FusedSGD with dampening != 0: new and restored momentum buffers in one optimizer")
            self._first = bool(fresh)
        descs = (_lib.dsn_sgd_desc * len(entries))()
        first = 0
        for i, (p, g, b, gi) in enumerate(entries):
            descs[i] = _lib.dsn_sgd_desc(p.data_ptr(), g.data_ptr(), b.data_ptr(), p.numel(), gi, first)
            first += (p.numel() + chunk - 1) // chunk
        key = tuple((p.data_ptr(), g.data_ptr(), b.data_ptr()) for p, g, b, _ in entries)
        return dict(key=key, n=len(entries), chunks=first, entries=entries,
                    descs=torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev),
                    hyper=torch.zeros(8 * len(self.param_groups), dtype=torch.float32, device=dev))

    def _hyper_values(self):
        vals = []
        for g in self.param_groups:
            vals += [float(g["lr"]), float(g["momentum"]), float(g["dampening"]), float(g["weight_decay"]),
                     1.0 if g["nesterov"] else 0.0, 1.0 if self._first else 0.0, 0.0, 0.0]
        return vals

    def sync_hyper(self):
        """Upload lr / momentum / ... if they changed (call between graph replays when a scheduler moved them)."""
        if self._table is None:
            return
        vals = self._hyper_values()
        if vals != self._uploaded:
            self._table["hyper"].copy_(torch.tensor(vals, dtype=torch.float32))
            self._uploaded = vals

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_current_stream_capturing()
        if self._table is None or (not capturing and self._table["key"] != tuple(
                (p.data_ptr(), p.grad.data_ptr(), self.state[p]["momentum_buffer"].data_ptr())
                for p, _, _, _ in self._table["entries"] if p.grad is not None)):
            if capturing and self._table is None:
                raise RuntimeError("FusedSGD: take one eager step before capturing a graph (tables are built on first use)")
            self._table = self._build()
            self._uploaded = None
        if self._table is None:
            return loss
        if not capturing:
            self.sync_hyper()
        t = self._table
        _lib.check(_lib.lib().dsn_sgd_step(t["descs"].data_ptr(), t["n"], t["chunks"], t["hyper"].data_ptr(), stream_ptr()),
                   "sgd_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that everything keyed on
        # them (packed-weight caches, saved-tensor checks) sees the update
        torch.autograd.graph.increment_version([p for p, _, _, _ in t["entries"]])
        if self._first and not capturing:
            self._first = False        # momentum buffers now hold g': later steps blend (uploaded before the next launch)
        return loss

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict, then rebuild the device table around the restored momentum buffers (they
        are blended into, never overwritten, by the next step -- as torch.optim.SGD does for parameters that have a buffer)."""
        super().load_state_dict(state_dict)
        self._table = None
        self._uploaded = None
        self._first = False

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._table = None
        self._uploaded = None
