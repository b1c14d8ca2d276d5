"""hipGraph replay of the training step: the launches of one DeSeNet-s step (weight pack, forward, losses, backward,
optimizer) are enqueued by ONE hipGraphLaunch instead of several hundred Python->ctypes calls (host launch cost exceeded
device time by 1.4x in eager mode).  No tracing compiler is involved: capture records exactly the launches the eager path
makes; every buffer the graph touches lives in its private memory pool and the batch -- images AND labels -- is copied into
static tensors before each replay.

    world_size == 1, accumulate == 1 : G = [zero flat grads | pack | forward | det + seg loss (HIP) | backward | SGD | EMA]
    accumulate  > 1                  : G0 = [zero | pack | forward | losses | backward]  (first micro-batch of a window)
                                       G1 = [       pack | forward | losses | backward]  (the others: gradients add up)
                                       G_opt = [SGD | EMA] after the last one            (scripts/train.py:370-376)
    world_size  > 1                  : the backward pass is captured in two halves around `split_layer`:
                                       GA = [zero | pack | forward | losses | backward of layers >= split | their weight grads]
                                       -> all-reduce (RCCL, asynchronous, its own stream) of the flat-buffer tail that
                                          holds those layers' gradients
                                       GB = [backward of layers < split | their weight gradients]   (overlaps the collective)
                                       -> all-reduce of the head of the buffer -> wait for both -> G_opt
Labels: `det_targets` [n, 6] rows (image, class, x, y, w, h) go into a fixed-capacity [max_targets, 6] buffer whose unused rows
are zero (w = h = 0 fails the anchor-ratio test of build_targets, loss.py:192-195, so padded rows match nothing and the
candidate order of the real rows is unchanged); `seg_targets` [N, H, W] int64 are copied as they are.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import hip_ops as ops
from .runtime import Tape


class GraphedTrainStep:
    def __init__(self, model, loss_and_grads: Callable, flat, optimizer, example_input: torch.Tensor, warmup: int = 3,
                 ema=None, det_targets: Optional[torch.Tensor] = None, seg_targets: Optional[torch.Tensor] = None,
                 max_targets: int = 0, accumulate: int = 1, restore_after_warmup: bool = True,
                 split_layer: Optional[int] = None, fuse_seg_loss: bool = True):
        """loss_and_grads(det_out, seg_out, det_targets, seg_targets) -> (loss tensor, d_det, d_seg): HIP kernels only
        (capturable); it must read the labels from the tensors it is handed (the step's static buffers), not from a closure.
        Legacy form: without det_targets / seg_targets the callable is invoked as loss_and_grads(det_out, seg_out) and whatever
        labels it closes over are frozen into the graph (benchmarks with one fixed batch only).
        accumulate: micro-batches per optimizer step (train.py:146,337 `accumulate = round(nbs / batch_size)`).
        restore_after_warmup: the `warmup` eager steps that populate caches / workspaces / optimizer state are undone
        afterwards (weights, BatchNorm buffers, momentum buffers, EMA), so that building the step does not train.
        ema: optional desenet_amd ModelEMA, updated right after the optimizer step (train.py:374-375) inside the graph.
        fuse_seg_loss: the seg head hands the loss its 1/8-resolution logits and SegmentationLosses.forward_backward fuses the x8
        bilinear up-sampling (yolo.py:183) with the cross entropy, forward and backward -- the N x 2 x 640 x 640 fp32 logits and
        their gradient are never written.  Needs a loss_and_grads built on SegmentationLosses.forward_backward.
        split_layer (world_size > 1): first top-level layer index of the second all-reduce chunk (default: the layer at which
        about half of the parameters lie behind; 0 disables the overlap)."""
        self.model, self.loss_and_grads, self.flat, self.opt, self.ema = model, loss_and_grads, flat, optimizer, ema
        if any(m.__dict__.get("_dsn_sync") is not None for m in model.modules()):
            # SyncBatchNorm (train.py:217-220): every BatchNorm exchanges its fp64 accumulators once per direction.  Those
            # all-reduces are recorded INTO the captured graphs like any other launch -- possible with RCCL, whose collectives are
            # kernels enqueued on the capturing stream; a host-side backend (gloo) cannot be captured.
            if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
                raise NotImplementedError("GraphedTrainStep with SyncBatchNorm needs the nccl (RCCL) backend: the per-layer "
                                          "all-reduces are captured into the hipGraph; with gloo use eager steps")
        if accumulate < 1:
            raise ValueError("accumulate must be >= 1")
        self.accumulate = int(accumulate)
        self.micro = 0
        self.fuse_seg_loss = bool(fuse_seg_loss)
        self.x = example_input.clone()
        dev = self.x.device
        self.det_t = self.seg_t = None
        if (det_targets is None) != (seg_targets is None):
            raise ValueError("pass both det_targets and seg_targets (or neither: legacy frozen-label form)")
        if det_targets is not None:
            cap = max(int(max_targets), int(det_targets.shape[0]), 1)
            self.det_t = torch.zeros((cap, 6), dtype=torch.float32, device=dev)
            self.seg_t = seg_targets.to(device=dev, dtype=torch.int64).clone()
            self._set_labels(det_targets, None)
        self.multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.split = self._pick_split(split_layer) if self.multi else 0
        if self.split > 0 and any(m.__dict__.get("_dsn_sync") is not None for m in model.modules()):
            # SyncBatchNorm's per-layer all-reduces are CAPTURED into the backward graphs; the overlapped gradient all-reduce is an
            # eager collective on the same communicator.  One communicator must not carry a captured and an eager operation in
            # flight at once (no device-side order between them): run the step unsplit, gradient all-reduce after the whole backward.
            self.split = 0

        snap = self._snapshot() if restore_after_warmup else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # populate caches / workspaces / optimizer state outside capture
                self.flat.zero()
                self._forward_backward()
                self.flat.all_reduce()
                self.opt.step()
                if self.ema is not None:
                    self.ema.update(model)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if snap is not None:
            self._restore(snap)
            torch.cuda.synchronize(dev)

        self.pool = torch.cuda.graph_pool_handle()
        G = torch.cuda.CUDAGraph
        cap_kw = dict(pool=self.pool, capture_error_mode="thread_local")   # other threads (the RCCL watchdog polls events)
        single = (not self.multi) and self.accumulate == 1
        self.g_first = self.g_next = self.g_tail = self.g_opt = None
        with torch.no_grad():
            if single:
                self.g_first = G()
                with torch.cuda.graph(self.g_first, **cap_kw):
                    self.flat.zero()
                    self.loss = self._forward_backward()
                    self._optimizer()
            else:
                self.g_first = G()
                tape = None
                with torch.cuda.graph(self.g_first, **cap_kw):
                    self.flat.zero()
                    self.loss, tape = self._forward_backward(stop_at=self.split)
                # BatchNorm accumulator slots: the forward pass of each micro-batch graph clears the arena and hands slots out from
                # its start; the second half of that micro-batch's backward must CONTINUE from where its first half stopped (not from
                # wherever the previously captured graph left the cursor), or its slots would lie beyond what the clear covers.
                cur_first = ops.bn_arena_cursor(dev)
                if self.accumulate > 1:
                    self.g_next = G()
                    with torch.cuda.graph(self.g_next, **cap_kw):
                        self.loss_next, tape_n = self._forward_backward(stop_at=self.split)
                    cur_next = ops.bn_arena_cursor(dev)
                if self.split > 0:
                    self.g_tail = G()
                    ops.bn_arena_cursor(dev, cur_first)
                    with torch.cuda.graph(self.g_tail, **cap_kw):
                        self._backward_rest(tape)
                    if self.accumulate > 1:
                        self.g_tail_next = G()
                        ops.bn_arena_cursor(dev, cur_next)
                        with torch.cuda.graph(self.g_tail_next, **cap_kw):
                            self._backward_rest(tape_n)
                del tape
                self.g_opt = G()
                with torch.cuda.graph(self.g_opt, **cap_kw):
                    self._optimizer()
        torch.cuda.synchronize(dev)

    # ---- pieces of a step ---------------------------------------------------------------------------------------------------
    def _losses(self, det, seg):
        if self.det_t is not None:
            return self.loss_and_grads(det, seg, self.det_t, self.seg_t)
        return self.loss_and_grads(det, seg)

    def _forward_backward(self, stop_at: Optional[int] = None):
        """forward + losses + backward (down to top-level layer `stop_at`, exclusive of the layers below it, when given).
        Returns the loss, or (loss, tape) when stop_at is given."""
        tape = Tape()
        heads = [m for m in self.model.modules() if type(m).__name__ == "SegMaskPSP"] if self.fuse_seg_loss else []
        for m in heads:
            m.__dict__["_dsn_lowres_out"] = True
        try:
            det, seg = self.model.fwd(self.x, tape)
        finally:
            for m in heads:
                m.__dict__.pop("_dsn_lowres_out", None)
        loss, d_det, d_seg = self._losses(det, seg)
        tape.begin_backward()
        if stop_at is None:
            self.model.bwd(tape, (d_det, d_seg), need_dx=False)
            self._finish_backward(tape)
            return loss
        tape.bwd_state = self.model.bwd_begin((d_det, d_seg))
        self.model.bwd_layers(tape, tape.bwd_state, len(self.model.model) - 1, stop_at)
        self._finish_backward(tape)
        return loss, tape

    def _backward_rest(self, tape):
        self.model.bwd_layers(tape, tape.bwd_state, self.split - 1, 0)
        self._finish_backward(tape)

    @staticmethod
    def _finish_backward(tape):
        tape.join()                              # queued weight gradients of the layers walked so far: three launches
        for p, g in tape.grads.items():          # parameters without a pre-attached .grad slot (none with FlatGradients)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        tape.grads = {}

    def _optimizer(self):
        self.opt.step()
        if self.ema is not None:
            self.ema.update(self.model)

    # ---- multi-rank: where to cut the backward pass ---------------------------------------------------------------------------
    def _pick_split(self, split_layer):
        layers = list(self.model.model)
        if split_layer is not None:
            return max(0, min(int(split_layer), len(layers) - 1))
        counts = [sum(p.numel() for p in m.parameters() if p.requires_grad) for m in layers]
        total, run = sum(counts), 0
        for i in range(len(layers) - 1, 0, -1):
            run += counts[i]
            if run * 2 >= total:
                return i
        return 0

    def _flat_ranges(self):
        """(tail, head): flat-buffer element ranges of the parameters of layers >= split / < split.  FlatGradients lays the
        parameters out in `model.parameters()` order, i.e. by top-level layer index."""
        cache = getattr(self, "_ranges", None)
        if cache is None:
            first_tail = None
            ids = {id(p) for m in list(self.model.model)[self.split:] for p in m.parameters()}
            off = 0
            for p in self.flat.params:
                if id(p) in ids:
                    if first_tail is None:
                        first_tail = off
                elif first_tail is not None:
                    raise RuntimeError("flat gradient buffer is not ordered by layer: cannot split the all-reduce")
                off += p.numel()
            first_tail = off if first_tail is None else first_tail
            cache = self._ranges = ((first_tail, off), (0, first_tail))
        return cache

    # ---- warm-up undo ---------------------------------------------------------------------------------------------------------
    def _snapshot(self):
        snap = {"model": [t.detach().clone() for t in self.model.state_dict().values()],
                "had_opt_state": {id(p): ("momentum_buffer" in self.opt.state.get(p, {})
                                          and self.opt.state[p]["momentum_buffer"] is not None)
                                  for g in self.opt.param_groups for p in g["params"]},
                "opt": {id(p): self.opt.state[p]["momentum_buffer"].detach().clone()
                        for g in self.opt.param_groups for p in g["params"]
                        if "momentum_buffer" in self.opt.state.get(p, {}) and self.opt.state[p]["momentum_buffer"] is not None}}
        if self.ema is not None:
            snap["ema"] = [t.detach().clone() for t in self.ema.ema.state_dict().values()]
            snap["ema_updates"] = self.ema.updates
        return snap

    def _restore(self, snap):
        with torch.no_grad():
            for t, s in zip(self.model.state_dict().values(), snap["model"]):
                t.copy_(s)
            for g in self.opt.param_groups:
                for p in g["params"]:
                    st = self.opt.state.get(p, {})
                    b = st.get("momentum_buffer")
                    if b is None:
                        continue
                    if snap["had_opt_state"].get(id(p)):
                        b.copy_(snap["opt"][id(p)])
                    else:
                        b.zero_()            # created by the warm-up: a zero buffer is torch.optim.SGD's "no buffer yet"
            if self.ema is not None:
                for t, s in zip(self.ema.ema.state_dict().values(), snap["ema"]):
                    t.copy_(s)
                self.ema.updates = snap["ema_updates"]
            torch.autograd.graph.increment_version([p for p in self.model.parameters()])

    # ---- per-step API -----------------------------------------------------------------------------------------------------------
    def _stage(self, x, det_targets, seg_targets):
        """The batch -> the static buffers the graphs read, in ONE launch (dsn_copy_multi: image, label rows zero-padded to the
        buffer's capacity, masks) when everything is already on the device in the buffers' dtypes; per-tensor copies otherwise."""
        pairs, slow = [], []
        if x is not None and x.data_ptr() != self.x.data_ptr():
            pairs.append((self.x, x))
        if det_targets is not None or seg_targets is not None:
            if self.det_t is None:
                raise RuntimeError("this step was built without static label buffers (legacy form): pass det_targets / "
                                   "seg_targets to the constructor")
            if det_targets is not None:
                n = int(det_targets.shape[0])
                if n > self.det_t.shape[0]:
                    raise ValueError(f"{n} label rows exceed this step's capacity ({self.det_t.shape[0]}); build it with a larger max_targets")
                pairs.append((self.det_t, det_targets if n else None))
            if seg_targets is not None:
                pairs.append((self.seg_t, seg_targets))
        for dst, src in pairs:
            # images and masks must fill their buffers exactly: a smaller final batch (the reference's loader does not drop_last)
            # would otherwise train on zero-padded images with all-background masks; only the label ROWS may be fewer
            if src is not None and dst is not self.det_t and tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"batch shape {tuple(src.shape)} does not match the captured step's {tuple(dst.shape)}: run a "
                                 "partial batch through an eager step (or build a second GraphedTrainStep for it)")
            if src is not None and (src.device != dst.device or src.dtype != dst.dtype or not src.is_contiguous()
                                    or src.numel() > dst.numel() or (src.data_ptr() | dst.data_ptr()) % 4
                                    or (src.numel() * src.element_size()) % 4):
                slow.append((dst, src))
        fast = [p for p in pairs if not any(p[0] is d for d, _ in slow)]
        if fast:
            ops.copy_multi(fast)
        for dst, src in slow:
            if dst is self.det_t:
                dst.zero_()
                dst[:src.shape[0]].copy_(src.to(dst.dtype), non_blocking=True)
            else:
                dst.copy_(src, non_blocking=True)

    def _set_labels(self, det_targets, seg_targets):
        self._stage(None, det_targets, seg_targets)

    def __call__(self, x: torch.Tensor = None, det_targets: torch.Tensor = None, seg_targets: torch.Tensor = None):
        """One micro-batch: copies the batch into the static buffers, replays forward/backward, and -- on the last micro-batch
        of an accumulation window -- all-reduces and steps the optimizer.  Returns the (device) loss of this micro-batch."""
        self._stage(x, det_targets, seg_targets)
        first = self.micro == 0
        last = self.micro == self.accumulate - 1
        if self.g_opt is None:                   # single rank, no accumulation: one replay is the whole step
            if self.ema is not None:
                self.ema.tick()                  # updates += 1, decay for this step -> device
            self.g_first.replay()
            return self.loss
        (self.g_first if first else self.g_next).replay()
        loss = self.loss if first else self.loss_next
        work = None
        if self.split > 0:
            if self.multi and last:
                (lo, hi), _ = self._flat_ranges()
                if hi > lo:
                    work = dist.all_reduce(self.flat.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
            (self.g_tail if first else self.g_tail_next).replay()
        if last:
            if self.multi:
                if self.split > 0:
                    _, (lo, hi) = self._flat_ranges()
                    if hi > lo:
                        dist.all_reduce(self.flat.flat[lo:hi], op=dist.ReduceOp.SUM)
                    if work is not None:
                        work.wait()
                else:
                    self.flat.all_reduce()
            if self.ema is not None:
                self.ema.tick()
            self.g_opt.replay()
        self.micro = 0 if last else self.micro + 1
        return loss


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
