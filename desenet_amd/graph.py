"""hipGraph replay of the training step: the ~1500 kernel launches of one DeSeNet-s step (weight pack, forward, losses,
backward, optimizer) are enqueued by ONE hipGraphLaunch instead of ~1500 Python->ctypes calls (host launch cost exceeded
device time by 1.4x in eager mode).  No tracing compiler is involved: capture records exactly the launches the eager path
makes; every buffer the graph touches lives in its private memory pool and inputs are copied into static tensors.

    world_size == 1 : G = [zero flat grads | pack weights | forward | det + seg loss (HIP) | backward | optimizer step]
    world_size  > 1 : G_a = [zero | pack | forward | losses | backward]  ->  eager RCCL all-reduce of the flat gradient
                      buffer  ->  G_b = [optimizer step]
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist

from .runtime import Tape


class GraphedTrainStep:
    def __init__(self, model, loss_and_grads: Callable, flat, optimizer, example_input: torch.Tensor, warmup: int = 3,
                 ema=None):
        """loss_and_grads(det_out, seg_out) -> (loss tensor, d_det, d_seg): HIP kernels only (capturable).
        ema: optional desenet_amd ModelEMA, updated right after the optimizer step (scripts/train.py:374-375) inside the graph."""
        self.model, self.loss_and_grads, self.flat, self.opt, self.ema = model, loss_and_grads, flat, optimizer, ema
        if any(m.__dict__.get("_dsn_sync") is not None for m in model.modules()):
            raise NotImplementedError("GraphedTrainStep with SyncBatchNorm: the per-layer collectives run eagerly; use eager steps")
        self.x = example_input.clone()
        dev = self.x.device
        self.multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # populate caches / workspaces / optimizer state outside capture
                self._body()
                self.flat.all_reduce()
                self.opt.step()
                if self.ema is not None:     # the warm-up steps are real training steps: the EMA follows them
                    self.ema.update(model)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        self.pool = torch.cuda.graph_pool_handle()
        self.ga, self.gb = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if self.multi else None)
        with torch.no_grad():
            # thread_local: other threads (the RCCL watchdog polls events) must not invalidate the capture
            with torch.cuda.graph(self.ga, pool=self.pool, capture_error_mode="thread_local"):
                self.loss = self._body()
                if not self.multi:
                    self.opt.step()
                    if self.ema is not None:
                        self.ema.update(self.model)
            if self.multi:
                with torch.cuda.graph(self.gb, pool=self.pool, capture_error_mode="thread_local"):
                    self.opt.step()
                    if self.ema is not None:
                        self.ema.update(self.model)
        torch.cuda.synchronize(dev)

    def _body(self):
        self.flat.zero()
        tape = Tape()
        det, seg = self.model.fwd(self.x, tape)
        loss, d_det, d_seg = self.loss_and_grads(det, seg)
        tape.begin_backward()
        self.model.bwd(tape, (d_det, d_seg), need_dx=False)
        tape.join()                              # weight gradients run on a side stream (parallel graph branch)
        for p, g in tape.grads.items():          # parameters without a pre-attached .grad slot (none with FlatGradients)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return loss

    def __call__(self, x: torch.Tensor = None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        if self.ema is not None:
            self.ema.tick()                      # updates += 1, decay for this step -> device
        self.ga.replay()
        if self.multi:
            self.flat.all_reduce()
            self.gb.replay()
        return self.loss


class GraphedInference:
    """hipGraph replay of the eval-mode forward pass for a fixed input shape (detect.py / val.py run batches of one shape):
    the ~230 launches of a fused DeSeNet-s forward become one `hipGraphLaunch`.  Returns the model's usual eval output
    `((pred, raws), seg)` -- views of buffers that the next call overwrites.  NMS stays outside (its output sizes are data)."""

    def __init__(self, model, example_input: torch.Tensor, warmup: int = 2):
        if model.training:
            raise ValueError("GraphedInference captures the eval-mode forward: call model.eval() (and .fuse()) first")
        self.model = model
        self.x = example_input.clone()
        dev = self.x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                  # packed-weight caches, workspaces
                model(self.x)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.g, capture_error_mode="thread_local"):
            self.out = model(self.x)
        torch.cuda.synchronize(dev)

    def __call__(self, x: torch.Tensor = None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.g.replay()
        return self.out
