"""Conv2d (+BatchNorm2d) (+activation) block on the HIP kernels: the work-horse behind Conv, the raw Conv2d+BN+SiLU triples
of RFB2, FFM's attention 1x1s, Detect.m and the seg classifier.

Numerical contract (what the reference computes):
  eval  : act(conv(x, W) * g/sqrt(var+eps) + (b - mean*g/sqrt(var+eps)))  -- ONE launch: the BN affine is folded into the
          packed weights + epilogue bias (identical math for a fused and an un-fused model; torch_utils.py:196-216).
  Q1    : an un-fused `Conv` on a 1x1 map skips BN altogether (common.py:53) -- `q1=True` -- while a fused one applies it.
  train : y = conv(x, W) ; batch statistics (eps 1e-3, momentum .03; torch_utils.py:164-165) ; z = act(BN(y)) [+ shortcut]
          backward: dy from (dz, y) ; dW = wgrad(x, dy) ; dx = dgrad(dy, W).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib as _lib_mod
from . import hip_ops as ops
from .hip_ops import ACT_NONE, ACT_SIGMOID, ACT_SILU

BN_EPS, BN_MOMENTUM = 1e-3, 0.03


def act_code(act) -> int:
    if act is None or isinstance(act, nn.Identity):
        return ACT_NONE
    if isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, nn.Sigmoid):
        return ACT_SIGMOID
    raise NotImplementedError(f"activation {type(act).__name__} has no HIP epilogue (SiLU, Sigmoid, Identity only)")


def _conv_geom(conv: nn.Conv2d):
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    if conv.groups != 1 or k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d[0] != d[1] or conv.padding_mode != "zeros":
        raise NotImplementedError("HIP conv supports square kernels, groups=1, zero padding (all DeSeNet-s convs)")
    return k[0], s[0], p[0], d[0]


def _ver(*ts):
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)


def _cache(conv):
    c = conv.__dict__.get("_dsn_cache")
    if c is None:
        c = conv.__dict__["_dsn_cache"] = {}
    return c


def packed_fwd(conv: nn.Conv2d, dtype, ci_pad: Optional[int] = None, bn: Optional[nn.BatchNorm2d] = None):
    """(w_packed [Co][KH][KW][ci_pad], bias fp32 [Co] | None); BN running stats folded in when `bn` is given (eval)."""
    key = ("fwd", dtype, ci_pad, bn is not None)
    ver = _ver(conv.weight, conv.bias, *((bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else ()))
    hit = _cache(conv).get(key)
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    scale = bias = None
    if bn is not None:   # host-side per-channel vectors (computed once per weight version)
        g = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
        b = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_var)
        scale = (g.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)).contiguous()
        bias = b.detach().float() - bn.running_mean.float() * scale
        if conv.bias is not None:
            bias = bias + conv.bias.detach().float() * scale
        bias = bias.contiguous()
    elif conv.bias is not None:
        bias = conv.bias.detach().float().contiguous()
    w = ops.pack_weight_fwd(conv.weight, dtype, scale, ci_pad)
    _cache(conv)[key] = (ver, w, bias)
    return w, bias


def packed_dgrad(conv: nn.Conv2d, dtype, co_pad: Optional[int] = None):
    """[Ci][KH][KW][co_pad] (co_pad > Co: zero-padded K axis for a dy with zero-padded rows; see Detect.bwd)."""
    co_pad = co_pad if co_pad is not None else conv.out_channels
    key = ("dgrad", dtype) if co_pad == conv.out_channels else ("dgrad", dtype, co_pad)
    ver = _ver(conv.weight)
    hit = _cache(conv).get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    w = ops.pack_weight_dgrad(conv.weight, dtype, co_pad)
    _cache(conv)[key] = (ver, w)
    return w


def out_shape(conv: nn.Conv2d, x):
    k, s, p, d = _conv_geom(conv)
    ho, wo = ops.conv_out_hw(x.shape[2], x.shape[3], k, s, p, d)
    return x.shape[0], conv.out_channels, ho, wo


def _bn_sync(bn):
    """(process group, world size) when this BatchNorm was converted by parallel.convert_sync_batchnorm and more than one rank
    is running, else None."""
    tag = bn.__dict__.get("_dsn_sync")
    if tag is None:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = None if tag is True else tag
    world = dist.get_world_size(group)
    # (a single rank needs no exchange; `_dsn_sync_force` keeps the collectives in for tests of the code path)
    return (group, world) if (world > 1 or bn.__dict__.get("_dsn_sync_force")) else None


import os as _os

# BatchNorm in training: the convolution's epilogue adds per-channel sums into fp64 accumulators, ONE elementwise launch
# (dsn_lazy_materialize) folds them in its prologue and writes z = act(bn(y)) (+ shortcut), and the saved statistics / running
# averages of all modules are written by one finalisation launch at the end of the forward pass (runtime.Tape.finalize_forward).
# (Rounds 2-3 also carried CONSUMER-side forms -- act(bn(.)) applied in the next convolution's operand loader, or once per block
#  inside the LDS-DMA kernels -- behind DSN_LAZY_TAPS / DSN_LAZY_Z.  Both measured slower, 5.01 -> 5.47 / 6.66 ms and 4.77 -> 5.04 ms
#  per step, and were deleted in round 4 together with their kernel instantiations.)


def _defer_ok(bn, out, co, bias, sync) -> bool:
    """Can this BatchNorm's output stay raw / be produced through the deferred path (16-byte channel vectors everywhere)?"""
    return bias is None and sync is None and co <= 1024 and co % 8 == 0 and ops._vec16(out)


def conv_block_fwd(x, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], act: int, training: bool, tape=None, out=None,
                   residual=None, q1: bool = False, ci_pad: Optional[int] = None):
    """x: NHWC-backed activation (channels may be zero-padded up to ci_pad).  Returns z (written into `out` if given)."""
    k, s, p, d = _conv_geom(conv)
    n, co, ho, wo = out_shape(conv, x)
    dtype = x.dtype
    if out is None:
        out = ops.new_act(n, co, ho, wo, dtype, x.device)
    skip_bn = bn is None or (q1 and x.shape[2] * x.shape[3] == 1)
    train_bn = training and not skip_bn
    if tape is None and not train_bn:
        # inference: one fused launch
        w, bias = packed_fwd(conv, dtype, ci_pad, None if skip_bn else bn)
        ops.conv2d_fwd(x, w, bias, residual, out, ops.conv_params(k, s, p, d, act))
        return out
    # training (or eval-mode forward that must stay differentiable)
    w, bias = packed_fwd(conv, dtype, ci_pad, None)
    plain = skip_bn and act == ACT_NONE and residual is None
    sync = _bn_sync(bn) if train_bn else None
    rec = dict(conv=conv, bn=None if skip_bn else bn, act=act, x=x, x_in=x, ci_pad=ci_pad, geom=(k, s, p, d), plain=plain)
    if train_bn and tape is not None and _defer_ok(bn, out, co, bias, sync):
        # conv (+ BatchNorm sums in its epilogue), then ONE elementwise launch writes z (+ shortcut).  Saved statistics / running
        # averages: end-of-forward finalisation.
        y = ops.new_act(n, co, ho, wo, dtype, x.device)
        acc, _ = ops.conv2d_fwd_acc(x, w, y, ops.conv_params(k, s, p, d, ACT_NONE))
        stats = torch.empty((4, co), dtype=torch.float32, device=x.device)
        count = n * ho * wo
        ops.lazy_materialize(y, _self_lazy(acc, co, 0, 0, co, count, bn, act), out, residual, None)
        tape.lazy_pending_only(acc, co, 0, co, count, bn, stats)
        if bn.num_batches_tracked is not None and not bn.__dict__.get("_dsn_shared_counter"):
            bn.num_batches_tracked.add_(1)    # (a Model increments all of its counters with one launch per step)
        rec.update(y=y, scale=stats[0], shift=stats[1], mean=stats[2], rstd=stats[3], frozen=False, sync=None)
        tape.push(rec)
        tape.bn_register(out, rec)
        return out
    y = out if plain else ops.new_act(n, co, ho, wo, dtype, x.device)
    rec["y"] = y
    stats = None
    if train_bn and bias is None and co <= 1024:
        # BatchNorm statistics come out of the conv epilogue (fp32 accumulators) and are folded in the prologue of the
        # BN + act kernel: conv -> BN -> act is two launches, y is read once
        stats = ops.conv2d_fwd_bnstats(x, w, y, ops.conv_params(k, s, p, d, ACT_NONE), bn.weight, bn.bias, bn.running_mean,
                                       bn.running_var, bn.momentum if bn.momentum is not None else BN_MOMENTUM, bn.eps,
                                       act, residual, out, sync=sync)
    else:
        if sync is not None:
            raise NotImplementedError("SyncBatchNorm after a biased or > 1024-channel convolution (not in DeSeNet)")
        ops.conv2d_fwd(x, w, bias, None, y, ops.conv_params(k, s, p, d, ACT_NONE))
    if plain:
        pass
    elif skip_bn:
        ops.bn_act_fwd(y, None, None, act, residual, out)
    elif train_bn:
        if stats is not None:
            scale, shift, mean, rstd = stats        # z already written by the fused pair
        else:
            scale, shift, mean, rstd = ops.bn_stats(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                    bn.momentum if bn.momentum is not None else BN_MOMENTUM, bn.eps)
            ops.bn_act_fwd(y, scale, shift, act, residual, out)
        if bn.num_batches_tracked is not None and not bn.__dict__.get("_dsn_shared_counter"):
            bn.num_batches_tracked.add_(1)    # (a Model increments all of its counters with one launch per step)
        rec.update(scale=scale, shift=shift, mean=mean, rstd=rstd, frozen=False, sync=sync)
        if tape is not None:
            tape.bn_register(out, rec)
    else:
        # eval-mode BN kept differentiable (running statistics are constants): z = act(y*scale + shift)
        g = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(bn.running_var)
        b = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(bn.running_var)
        scale = (g / torch.sqrt(bn.running_var.float() + bn.eps)).contiguous()
        shift = (b - bn.running_mean.float() * scale).contiguous()
        ops.bn_act_fwd(y, scale, shift, act, residual, out)
        rec.update(scale=scale, shift=shift, frozen=True)
    if tape is not None:
        tape.push(rec)
    return out


def _self_lazy(acc, acc_c, ch0, c0, c1, count, bn, act, second=None):
    """dsn_lazy_in describing a conv's OWN fresh output: channels [c0, c1) <- accumulator channels ch0.. of `bn`; `second` =
    (split, bn2): channels >= split belong to a second BatchNorm module (the merged C3 pair)."""
    from . import _lib
    lz = _lib.dsn_lazy_in()
    parts = [(c0, c1, ch0, bn)] if second is None else [(c0, second[0], ch0, bn), (second[0], c1, ch0 + second[0] - c0, second[1])]
    lz.nseg = len(parts)
    for i, (a, b, ch, m) in enumerate(parts):
        sg = lz.seg[i]
        sg.c0, sg.c1, sg.ch0, sg.p0, sg.acc_c, sg.act = a, b, ch, 0, acc_c, act
        sg.acc = acc.data_ptr()
        sg.gamma = m.weight.data_ptr() if m.weight is not None else None
        sg.beta = m.bias.data_ptr() if m.bias is not None else None
        sg.count, sg.eps = float(count), float(m.eps)
    return lz


def pair_ready(blk_a, blk_b, x, tape, dtype):
    """Can blk_a | blk_b (Conv modules: 1x1 conv + BN + act on the SAME input) run as one merged convolution right now?
    Needs the model's WeightBank to have packed them back to back for the current weights, training-mode BatchNorm on a map
    larger than 1x1, gradient slots to accumulate into, and identical BN hyper-parameters / sync state."""
    if tape is None or not (blk_a.training and blk_b.training) or blk_a.fused or blk_b.fused:
        return None
    ca, cb, ba, bb = blk_a.conv, blk_b.conv, blk_a.bn, blk_b.bn
    hit = _cache(ca).get(("pair", dtype))
    if hit is None or hit[1] is not cb or hit[0] != _ver(ca.weight, cb.weight) or x.shape[2] * x.shape[3] == 1:
        return None
    if ba.eps != bb.eps or ba.momentum != bb.momentum or type(blk_a.act) is not type(blk_b.act) or _bn_sync(ba) != _bn_sync(bb):
        return None
    if (ba.running_mean is None) != (bb.running_mean is None) or ca.out_channels + cb.out_channels > 1024:
        return None
    for p_ in (ca.weight, cb.weight, ba.weight, ba.bias, bb.weight, bb.bias):
        if _grad_slot(p_) is None:
            return None
    return hit


def pair_block_fwd(x, blk_a, blk_b, hit, tape, out):
    """z[:, :coA] = blk_a(x), z[:, coA:] = blk_b(x) as ONE convolution (`out`: coA+coB channels) and ONE elementwise launch over two
    BatchNorm modules that share an accumulator."""
    ca, cb, ba, bb = blk_a.conv, blk_b.conv, blk_a.bn, blk_b.bn
    _, _, wf, _ = hit
    n, _, h, w = x.shape
    coa = ca.out_channels
    co = coa + cb.out_channels
    act = act_code(blk_a.act)
    sync = _bn_sync(ba)
    x_in = x
    if _defer_ok(ba, out, co, None, sync) and coa % 8 == 0:
        y = ops.new_act(n, co, h, w, x.dtype, x.device)
        acc, _ = ops.conv2d_fwd_acc(x, wf, y, ops.conv_params(1, 1, 0, 1, ACT_NONE))
        stats = torch.empty((4, co), dtype=torch.float32, device=x.device)
        count = n * h * w
        ops.lazy_materialize(y, _self_lazy(acc, co, 0, 0, co, count, ba, act, second=(coa, bb)), out)
        tape.lazy_pending_only(acc, co, 0, coa, count, ba, stats, o0=0)
        tape.lazy_pending_only(acc, co, coa, co - coa, count, bb, stats, o0=coa)
        scale, shift, mean, rstd = stats[0], stats[1], stats[2], stats[3]
        sync = None
    else:
        y = ops.new_act(n, co, h, w, x.dtype, x.device)
        mom = ba.momentum if ba.momentum is not None else BN_MOMENTUM
        scale, shift, mean, rstd = ops.conv2d_fwd_bnstats(
            x, wf, y, ops.conv_params(1, 1, 0, 1, ACT_NONE), ba.weight, ba.bias, ba.running_mean, ba.running_var, mom, ba.eps, act,
            None, out, sync=sync, second=(coa, bb.weight, bb.bias, bb.running_mean, bb.running_var, None, None))
    for bn in (ba, bb):
        if bn.num_batches_tracked is not None and not bn.__dict__.get("_dsn_shared_counter"):
            bn.num_batches_tracked.add_(1)
    rec = dict(pair=(blk_a, blk_b), x=x, x_in=x_in, y=y, act=act, scale=scale, shift=shift, mean=mean, rstd=rstd, sync=sync)
    tape.push(rec)
    tape.bn_register(out, rec)           # (both modules: channel k of `out` is channel k of the merged BatchNorm pass)
    return out


_BNRED = int(_os.environ.get("DSN_BNRED", "2"))


def _bnred_plan(tape, x_in, dx, residual, rng=None):
    """The input-gradient launch about to write dx completes dz for the BatchNorm block(s) whose output the view x_in is (the
    caller vouches for that: fuse_up): (dsn_bnred, marks) so that their backward sums ride in its epilogue, or (None, ()) when the
    blocks / layouts cannot take it (more than two blocks, SyncBatchNorm, channel ranges that are not whole 16-byte vectors)."""
    if not _BNRED or tape is None or x_in is None:
        return None, ()
    hits = tape.bn_producers(x_in)
    if hits and isinstance(rng, tuple):  # only channels [lo, hi) of dx are complete (RFB2: ConvLinear's input is [x0 | x1 | x2 | x3]
        lo, hi = rng                     # and x0, x1 still receive the gradient of their other consumer)
        hits = [h for h in hits if h[0] >= lo and h[1] <= hi]
    if not hits or len(hits) > 2:       # DSN_BNRED_MAXSEG
        return None, ()
    vec = 4 if dx.dtype == torch.float32 else 8
    if not ops._vec16(dx) or dx.shape[1] % vec or (residual is not None and not ops._vec16(residual)):
        return None, ()
    segs, marks = [], []
    for c0, c1, prec, k0 in hits:
        y = prec.get("y")
        if prec.get("sync") is not None or prec.get("frozen") or y is None or "scale" not in prec or y.dtype != dx.dtype:
            return None, ()
        ctot, k1 = y.shape[1], k0 + (c1 - c0)
        if c0 % vec or c1 % vec or k0 % vec or ctot % vec or not ops._vec16(y) or tuple(y.shape[2:]) != tuple(dx.shape[2:]):
            return None, ()
        if any(a < k1 and k0 < b for a, b in prec.get("bnred_cov", ())):
            return None, ()                 # (somebody already summed these channels: this launch is not what completes them)
        acc = prec.get("bnred_acc")
        if acc is None:
            acc = ops.bn_acc(ctot, y.device)[0]
        segs.append((c0, c1, y[:, k0:k1], prec["scale"][k0:k1], prec["shift"][k0:k1], prec["mean"][k0:k1], prec["rstd"][k0:k1],
                     prec["act"], acc, ctot, k0))
        marks.append((prec, acc, k0, k1))
    return ops.bnred(segs), marks


def _bnred_commit(marks):
    for prec, acc, k0, k1 in marks:
        prec["bnred_acc"] = acc
        prec.setdefault("bnred_cov", []).append((k0, k1))


def _bnred_pre(rec):
    """(accumulator, covered channel ranges) left by the consumers' dgrad launches of THIS backward pass, consumed here (a second
    backward over the same tape starts from scratch)."""
    acc = rec.pop("bnred_acc", None)
    cov = rec.pop("bnred_cov", None)
    return (acc, cov) if acc is not None else None


def pair_block_bwd(tape, dz, dx=None, acc: bool = False, need_dx: bool = True, fuse_up: bool = False):
    """Backward of pair_block_fwd: one BN/act backward over the merged tensor, two queued weight-gradient jobs, one dgrad.
    fuse_up: see conv_block_bwd."""
    rec = tape.pop()
    blk_a, blk_b = rec["pair"]
    ca, cb, ba, bb = blk_a.conv, blk_b.conv, blk_a.bn, blk_b.bn
    x, y, act = rec["x"], rec["y"], rec["act"]
    dtype = x.dtype
    hit = _cache(ca).get(("pair", dtype))
    if hit is None or hit[0] != _ver(ca.weight, cb.weight):
        raise RuntimeError("merged C3 pair: the packed weights changed between forward and backward")
    coa = ca.out_channels
    dy = ops.new_act(*y.shape, dtype, y.device)
    ops.bn_act_bwd(dz, y, rec["scale"], rec["shift"], rec["mean"], rec["rstd"], act, dy, _grad_slot(ba.weight),
                   _grad_slot(ba.bias), accumulate=True, sync=rec["sync"],
                   second=(coa, None, None, None, None, _grad_slot(bb.weight), _grad_slot(bb.bias)), pre=_bnred_pre(rec))
    q = tape.wgrad_queue(x.device)
    for conv, sl in ((ca, dy[:, :coa]), (cb, dy[:, coa:])):
        if conv.weight.requires_grad:
            _wgrad(tape, x, sl, _grad_slot(conv.weight), conv.in_channels, ops.conv_params(1, 1, 0, 1, accumulate=True), q)
    if not need_dx:
        return None
    if dx is None:
        dx = ops.new_act(*x.shape, dtype, x.device)
        acc = False
    red, marks = _bnred_plan(tape, rec.get("x_in"), dx, None) if fuse_up else (None, ())
    _dgrad_red(dy, hit[3], dx, ops.conv_params(1, 1, 0, 1, accumulate=acc), None, red, marks)
    return dx


def _dgrad_red(dy, w, dx, params, residual, red, marks, s2=False):
    """Input gradient with the BatchNorm backward sums of `red` in its epilogue; a launch the fused entry point does not take
    (DSN_EUNSUPPORTED: dy not in 16-byte channel vectors, ...) is re-issued in its plain form and the sums are left to the
    producers' own reduction (include/desenet_hip.h: dsn_conv2d_dgrad_bnred)."""
    fn = ops.conv2d_dgrad_s2 if s2 else ops.conv2d_dgrad
    kw = {} if s2 else {"residual": residual}
    if red is not None:
        try:
            fn(dy, w, dx, params, red=red, **kw)
            _bnred_commit(marks)
            return
        except _lib_mod.Unsupported:
            pass
    fn(dy, w, dx, params, **kw)


def _wgrad(tape, x, dy, g, ci, params, queue):
    """dW (+)= wgrad(x, dy) into the OIHW gradient g (queued: planned now, launched with every other layer's at the end of the pass)."""
    ops.conv2d_wgrad(x, dy, g, ci, params, oihw=True, queue=queue)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _grad_slot(p):
    """The parameter's existing fp32 contiguous .grad (kernels then accumulate into it directly), else None."""
    if p is None or not p.requires_grad:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    return g


def conv_block_bwd(tape, dz, dx=None, acc: bool = False, need_dx: bool = True, residual=None, fuse_up: bool = False):
    """Backward of conv_block_fwd.  dz: gradient of the block output (the caller routes dz to a shortcut itself).
    residual: a tensor of dx's shape added to the input gradient in the dgrad epilogue (a shortcut's gradient).
    fuse_up: the caller vouches that the dx this call writes (with `residual` and, if acc, what dx already holds) is the COMPLETE
    gradient of the block's input -- no other contribution follows.  Where that input is the output of BatchNorm block(s) of this
    tape, their backward sums are then formed in the dgrad epilogue (dsn_conv2d_dgrad_bnred) and their own backward skips its
    reduction launch.  A (lo, hi) tuple instead of True restricts the claim to those channels of the block's input.
    A dz whose rows are padded with ZEROS up to a multiple of the vector width (ops.new_act(ldc_align=...), padding cleared by
    its producer -- Detect.bwd does this for its 33-channel heads) takes the 16-byte paths: the weight gradient ignores the
    padding lanes, the input gradient runs over the padded K axis with zero-padded weights."""
    rec = tape.pop()
    conv, bn, act, x, y = rec["conv"], rec["bn"], rec["act"], rec["x"], rec["y"]
    k, s, p, d = rec["geom"]
    dtype = x.dtype
    if rec["plain"]:
        dy = dz
    elif bn is None:
        dy = ops.act_bwd(dz, y, act, ops.new_act(*y.shape, dtype, y.device)) if act != ACT_NONE else dz
    elif rec["frozen"]:
        raise NotImplementedError("backward through eval-mode BatchNorm is not part of the DeSeNet training path")
    else:
        dy = ops.new_act(*y.shape, dtype, y.device)
        # gradient accumulation straight into existing .grad storage (FlatGradients views): no temporaries, no add kernels
        gw, gb = _grad_slot(bn.weight), _grad_slot(bn.bias)
        direct = gw is not None and gb is not None
        dg = gw if direct else torch.empty_like(rec["scale"])
        db = gb if direct else torch.empty_like(rec["scale"])
        ops.bn_act_bwd(dz, y, rec["scale"], rec["shift"], rec["mean"], rec["rstd"], act, dy, dg, db, accumulate=direct,
                       sync=rec.get("sync"), pre=_bnred_pre(rec))
        if not direct:
            tape.add_grad(bn.weight, dg)
            tape.add_grad(bn.bias, db)
    side = tape.fork(x.device) if (conv.weight.requires_grad or conv.bias is not None) else None
    ctx_mgr = torch.cuda.stream(side) if side is not None else _NullCtx()
    with ctx_mgr:
        if conv.weight.requires_grad:
            co, ci, kh, kw = conv.weight.shape
            slot = _grad_slot(conv.weight)
            g = slot if slot is not None else torch.empty((co, ci, kh, kw), dtype=torch.float32, device=x.device)
            _wgrad(tape, x, dy, g, ci, ops.conv_params(k, s, p, d, accumulate=slot is not None),
                   tape.wgrad_queue(x.device) if (side is None and conv.weight not in tape.grads) else None)
            if slot is None:
                tape.add_grad(conv.weight, g)
        if conv.bias is not None and conv.bias.requires_grad and not getattr(dz, "_dsn_bias_done", False):
            slot = _grad_slot(conv.bias)
            if slot is not None:
                ops.channel_sum(dy, out=slot, accumulate=True)
            else:
                tape.add_grad(conv.bias, ops.channel_sum(dy))
    if side is not None:
        dy.record_stream(side)     # dy is freed when this function returns; the side stream may still be reading it
        x.record_stream(side)
    if not need_dx:
        return None
    if dx is None:
        dx = ops.new_act(*x.shape, dtype, x.device)
        acc = False
    vec = 4 if dtype == torch.float32 else 8
    ldc = ops._nhwc_ldc(dy)
    if rec["plain"] and dy.shape[1] % vec != 0 and ldc is not None and ldc % vec == 0 and ldc - dy.shape[1] < vec \
            and getattr(dy, "_dsn_zero_padded", False):
        ops.conv2d_dgrad(ops.padded_view(dy), packed_dgrad(conv, dtype, ldc), dx, ops.conv_params(k, s, p, d, accumulate=acc))
    else:
        s2 = _cache(conv).get(("dgrad_s2", dtype)) if (k, s, p, d) == (3, 2, 1, 1) else None
        if s2 is not None and s2[0] == _ver(conv.weight) and dy.shape[1] % vec == 0 and dx.shape[1] % vec == 0:
            # stride-2 3x3: one 2x2 stride-1 conv over dy + depth-to-space store (weights packed by the model's WeightBank)
            # (measured A/B on DeSeNet-s, 3 x 100 steps each: 4.830 ms without any fused sums, 4.725 with the stride-1 launches only,
            #  4.707 with the two stride-2 stems as well -- DSN_BNRED=0 / 1 / 2)
            red, marks = _bnred_plan(tape, rec.get("x_in"), dx, None, fuse_up) if (fuse_up and residual is None and _BNRED >= 2) else (None, ())
            _dgrad_red(dy, s2[1], dx, ops.conv_params(k, s, p, d, accumulate=acc), None, red, marks, s2=True)
        else:
            red, marks = _bnred_plan(tape, rec.get("x_in"), dx, residual, fuse_up) if fuse_up else (None, ())
            _dgrad_red(dy, packed_dgrad(conv, dtype), dx, ops.conv_params(k, s, p, d, accumulate=acc), residual, red, marks)
            residual = None
    if residual is not None:                       # (paths without a fused epilogue add: not taken by Bottleneck's 1x1)
        ops.copy(residual, dx, accumulate=True)
    return dx
