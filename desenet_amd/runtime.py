"""Host-side runtime shared by the mirrored modules: compute dtype, the backward tape, the autograd bridge.

Every mirrored module (desenet_amd/core/models/*) implements

    fwd(x, tape=None, out=None)      enqueue HIP kernels; `tape` (None in inference) records what backward needs;
                                     `out` is an optional pre-allocated NHWC view to write into (concat-by-construction)
    bwd(tape, dy, dx=None, acc=False, need_dx=True)
                                     enqueue the backward kernels; writes (or, with acc, adds) the input gradient into
                                     `dx` when given; parameter gradients go to tape.grads

and composite modules call their children's fwd/bwd directly, so ONE autograd node (HipFunction) covers whatever
module the caller invoked -- the whole Model in training (scripts/train.py:352 `model(imgs)`), or a single block in a
unit test.  PyTorch's autograd only sees that node: no per-op Python dispatch, no ATen kernels on the hot path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

_compute_dtype = torch.float32


def set_compute_dtype(dtype):
    """fp32 (config 2: inference parity) or bf16 storage with fp32 accumulation (configs 3-5)."""
    global _compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dtype


def compute_dtype():
    """bf16 inside any torch.autocast region (the reference trains under amp.autocast, train.py:351), else the default."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return _compute_dtype


# ---- side stream for weight gradients --------------------------------------------------------------------------------------
# dW = wgrad(x, dy) feeds nothing but the optimizer, so it is taken off the critical path of backward: it runs on a second
# HIP stream, forked after dy is ready and joined once at the end of the backward pass.  Under hipGraph capture this
# becomes a parallel branch of the graph (the layers are small: concurrent kernels fill CUs that would idle otherwise).
import os as _os
_no_queue = bool(int(_os.environ.get("DSN_NO_WGRAD_QUEUE", "0")))     # debugging switch: launch every wgrad immediately
_side_streams = {}
_use_side_stream = False    # measured: no gain under hipGraph replay on ROCm 7.2 (branches are not overlapped); opt-in


def set_wgrad_side_stream(on: bool):
    global _use_side_stream
    _use_side_stream = bool(on)


def wgrad_stream(device):
    if not _use_side_stream:
        return None
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


class LazyRec:
    """One BatchNorm module whose saved statistics are still to be written: `acc` holds the per-channel fp64 sums the convolution's
    epilogue produced (channels ch0 .. ch0 + hi of an acc_c-channel accumulator); `stats` ([4, n]: scale, shift, mean, rstd,
    columns o0 ..) is filled by ONE dsn_bn_finalize_multi launch at the end of the forward pass, which also updates the running
    statistics -- the backward pass reads the arrays."""
    __slots__ = ("sp", "lo", "hi", "acc", "acc_c", "ch0", "count", "bn", "act", "stats", "o0", "keep")


class Tape:
    """LIFO of forward records with a replayable cursor (two backward() calls on one forward, train.py:366-367), plus the
    registry of BatchNorm-block outputs of this forward pass (keyed by MEMORY: storage + channel range, so channel slices and
    zero-copy concats resolve by themselves) and the BatchNorm modules whose statistics the end-of-forward launch finalises."""

    def __init__(self):
        self.stack: List[object] = []
        self.cursor = 0
        self.grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
        self.lazy_pending: List[LazyRec] = []
        self.bn_out: Dict[int, list] = {}

    # ---- producers of BatchNorm outputs, by memory (conv_impl: backward sums in the consumer's dgrad epilogue) -----------------
    def bn_register(self, out, rec):
        """`out` (a view) holds the output of the BatchNorm block recorded in `rec` (channel k of the block at channel k of the
        view).  The entry keeps `out` alive so that its memory cannot be handed to another tensor of this pass."""
        rng = self._range(out)
        if rng is not None:
            sp, _, lo, c = rng
            self.bn_out.setdefault(sp, []).append((lo, lo + c, rec, out))

    def bn_producers(self, x):
        """[(c0, c1, rec, k0)]: channels [c0, c1) of the view x are channels k0.. of the BatchNorm block `rec`; None if none are."""
        if not self.bn_out or not torch.is_tensor(x) or x.dim() != 4:
            return None
        rng = self._range(x)
        if rng is None:
            return None
        sp, _, lo, c = rng
        out = []
        for plo, phi, rec, _ in self.bn_out.get(sp, ()):
            a, b = max(lo, plo), min(lo + c, phi)
            if a < b:
                out.append((a - lo, b - lo, rec, a - plo))
        return out or None

    # ---- memory ranges / pending BatchNorm finalisation ----------------------------------------------------------------------------------------
    @staticmethod
    def _range(t):
        ops = _ops()
        ldc = ops._nhwc_ldc(t)
        if ldc is None:
            return None
        return t.untyped_storage().data_ptr(), ldc, (t.storage_offset() % ldc if ldc else 0), t.shape[1]

    def lazy_pending_only(self, acc, acc_c, ch0, n, count, bn, stats, o0=0):
        """A BatchNorm whose output was written by dsn_lazy_materialize: its saved statistics and running averages are still to be
        written by the end-of-forward finalisation."""
        r = LazyRec()
        r.sp, r.lo, r.hi, r.acc, r.acc_c, r.ch0, r.count, r.bn, r.act, r.stats, r.o0, r.keep = \
            None, 0, n, acc, acc_c, ch0, float(count), bn, 0, stats, o0, None
        self.lazy_pending.append(r)
        return r

    def finalize_forward(self):
        """End of the forward pass: ONE launch (per 40 modules) writes every pending BatchNorm's saved statistics and updates its
        running averages (torch_utils.py:164-165)."""
        if self.lazy_pending:
            _ops().bn_finalize_multi(self.lazy_pending)
            self.lazy_pending = []

    def push(self, rec):
        self.stack.append(rec)

    def begin_backward(self):
        self.cursor = len(self.stack)
        self.grads = {}
        self.forked = set()
        self.wq = None

    def wgrad_queue(self, device):
        """Queue that collects this backward pass's weight-gradient jobs (launched together by join()); None when the
        side-stream variant is selected instead."""
        if _use_side_stream or _no_queue:
            return None
        q = getattr(self, "wq", None)
        if q is None:
            q = self.wq = _ops().WgradQueue(device)
        return q

    def fork(self, device):
        """Side stream that may start once everything enqueued so far on the current stream is done (or None)."""
        side = wgrad_stream(device)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(device))
            self.forked.add(side)
        return side

    def join(self):
        """End of a backward pass: launch the queued weight gradients; the current stream waits for every side stream used
        since begin_backward()."""
        q = getattr(self, "wq", None)
        if q is not None:
            q.flush()
        for side in getattr(self, "forked", ()):
            torch.cuda.current_stream(side.device).wait_stream(side)
        self.forked = set()

    def pop(self):
        self.cursor -= 1
        return self.stack[self.cursor]

    def add_grad(self, param, g: torch.Tensor):
        if param is None or not param.requires_grad:
            return
        if param in self.grads:
            self.grads[param].add_(g)          # same parameter used twice in one graph: rare, not on the hot path
        else:
            self.grads[param] = g


def _flatten(x, out):
    if torch.is_tensor(x):
        out.append(x)
        return "t"
    return [_flatten(e, out) for e in x], type(x)


def flatten(x):
    out: List[torch.Tensor] = []
    spec = _flatten(x, out)
    return out, spec


def unflatten(spec, it):
    if spec == "t":
        return next(it)
    items, typ = spec
    vals = [unflatten(s, it) for s in items]
    return tuple(vals) if typ is tuple else vals


def _ops():
    from . import hip_ops
    return hip_ops


class HipFunction(torch.autograd.Function):
    """One autograd node around module.fwd / module.bwd."""

    @staticmethod
    def forward(ctx, module, n_in, in_spec, holder, *tensors):
        inputs = unflatten(in_spec, iter(tensors[:n_in]))
        tape = Tape()
        with torch.no_grad():
            out = module.fwd(inputs, tape)
            tape.finalize_forward()
        outs, out_spec = flatten(out)
        holder.append(out_spec)
        ctx.module, ctx.tape, ctx.n_in, ctx.out_spec = module, tape, n_in, out_spec
        ctx.params = tensors[n_in:]
        ctx.in_needs = [t.requires_grad for t in tensors[:n_in]]
        ctx.out_meta = [(o.shape, o.dtype, o.device) for o in outs]
        # outputs that are NHWC-backed activations: their incoming gradients are brought to the same storage/dtype
        ctx.out_is_act = [o.dim() == 4 and _ops()._nhwc_ldc(o) is not None and o.stride(1) == 1 for o in outs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        grads = [g if g is not None else torch.zeros(s, dtype=d, device=dev)
                 for g, (s, d, dev) in zip(grads, ctx.out_meta)]
        grads = [_ops().as_act(g.to(d)) if is_act else g
                 for g, is_act, (_, d, _) in zip(grads, ctx.out_is_act, ctx.out_meta)]
        dy = unflatten(ctx.out_spec, iter(grads))
        tape = ctx.tape
        tape.begin_backward()
        with torch.no_grad():
            dx = ctx.module.bwd(tape, dy, need_dx=any(ctx.in_needs))
            tape.join()
        dxs: List[Optional[torch.Tensor]] = []
        if dx is not None:
            dxs, _ = flatten(dx)
        dxs = list(dxs) + [None] * (ctx.n_in - len(dxs))
        dxs = [d if need else None for d, need in zip(dxs, ctx.in_needs)]
        pg = [tape.grads.get(p) if isinstance(p, torch.nn.Parameter) else None for p in ctx.params]
        return (None, None, None, None, *dxs, *pg)


def run_module(module, x):
    """nn.Module.forward of every mirrored module: through autograd when gradients are wanted, else plain fwd."""
    ins, in_spec = flatten(x)
    params = [p for p in module.parameters() if p.requires_grad]
    want = torch.is_grad_enabled() and (bool(params) or any(t.requires_grad for t in ins))
    if not want:
        with torch.no_grad():
            return module.fwd(x, None)
    holder: list = []
    outs = HipFunction.apply(module, len(ins), in_spec, holder, *ins, *params)
    return unflatten(holder[0], iter(outs))
