"""hipGraph replay of the training step: the ~1300 kernel launches of one DeSeNet-s forward/backward are enqueued by ONE
hipGraphLaunch instead of ~1300 Python->ctypes calls (host launch cost exceeded device time by 1.4x in eager mode).

    G1 = [zero flat grads | pack weights | forward]          captured once, replayed per step
         eager: losses on the (static) outputs -> d(raws), d(seg)      (PyTorch ops with data-dependent shapes; they become
                                                                         HIP kernels and join the graph in a later round)
    G2 = [backward through the model]                        captured once
         eager: flat gradient all-reduce (RCCL), only when world_size > 1
    G3 = [optimizer step]

Everything the graphs touch lives in one private memory pool (tape buffers written by G1 are read by G2), inputs are
copied into static tensors.  No tracing compiler is involved: capture records exactly the launches the eager path makes.
"""
from __future__ import annotations

from typing import Callable

import torch

from .runtime import Tape, flatten, unflatten


class GraphedTrainStep:
    def __init__(self, model, loss_fn: Callable, flat, optimizer, example_input: torch.Tensor, warmup: int = 3):
        """loss_fn(det_out, seg_out) -> scalar loss tensor (eager).  `flat` is a parallel.FlatGradients."""
        self.model, self.loss_fn, self.flat, self.opt = model, loss_fn, flat, optimizer
        self.x = example_input.clone()
        dev = self.x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # populate caches / workspaces / optimizer state outside capture
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        self.pool = torch.cuda.graph_pool_handle()
        self.g1, self.g2, self.g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.tape = Tape()
        with torch.no_grad():
            with torch.cuda.graph(self.g1, pool=self.pool):
                self.flat.zero()
                self.outs = self.model.fwd(self.x, self.tape)
        outs_flat, self.out_spec = flatten(self.outs)
        self.d_outs = [torch.zeros_like(o) for o in outs_flat]
        with torch.no_grad():
            with torch.cuda.graph(self.g2, pool=self.pool):
                self.tape.begin_backward()
                self.model.bwd(self.tape, unflatten(self.out_spec, iter(self.d_outs)), need_dx=False)
                for p, g in self.tape.grads.items():     # parameters without a pre-attached .grad slot (none with FlatGradients)
                    p.grad.add_(g) if p.grad is not None else setattr(p, "grad", g)
            with torch.cuda.graph(self.g3, pool=self.pool):
                self.opt.step()
        torch.cuda.synchronize(dev)

    def _eager_step(self):
        self.flat.zero()
        det, seg = self.model(self.x)
        self.loss_fn(det, seg).backward()
        self.flat.all_reduce()
        self.opt.step()

    def __call__(self, x: torch.Tensor = None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.g1.replay()
        outs_flat, _ = flatten(self.outs)
        leaves = [o.detach().requires_grad_(True) for o in outs_flat]
        det, seg = unflatten(self.out_spec, iter(leaves))
        loss = self.loss_fn(det, seg)
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        for d, g in zip(self.d_outs, grads):
            d.zero_() if g is None else d.copy_(g)
        self.g2.replay()
        self.flat.all_reduce()
        self.g3.replay()
        return loss
