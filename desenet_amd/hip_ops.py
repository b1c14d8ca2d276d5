"""Thin torch-tensor wrappers over the C ABI (include/desenet_hip.h).

Tensors are logical NCHW `torch.Tensor`s whose MEMORY is NHWC with an arbitrary pixel stride `ldc >= C`
(`new_act` allocates them; `x[:, c0:c1]` of such a tensor is a zero-copy channel slice the kernels can read or write in
place -- that is how concat buffers are built without torch.cat).  PyTorch supplies device memory and the current HIP
stream; every FLOP and byte moved on the hot path is done by libdesenet_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SIGMOID, ACT_SILU, DSN_BF16, DSN_F32, dsn_conv_params, dsn_tensor  # noqa: F401

_DT = {torch.float32: DSN_F32, torch.bfloat16: DSN_BF16}


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("desenet_amd kernels run on an MI355X only: got a CPU tensor (there is no CPU fallback; "
                           "the CPU oracle lives in oracle/ and is test infrastructure)")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def new_act(n: int, c: int, h: int, w: int, dtype, device, zero: bool = False, ldc_align: int = 1) -> torch.Tensor:
    """Logical NCHW tensor backed by a fresh dense NHWC buffer.  ldc_align > 1 rounds the pixel stride up (row padding after
    the c channels, e.g. 33 -> 40) so that 16-byte channel vectors stay aligned; the padding lanes are uninitialised."""
    ldc = (c + ldc_align - 1) // ldc_align * ldc_align
    buf = torch.empty((n, h, w, ldc), dtype=dtype, device=device)
    if zero:
        zero_(buf) if buf.is_cuda else buf.zero_()
    return buf.permute(0, 3, 1, 2)[:, :c] if ldc != c else buf.permute(0, 3, 1, 2)


def padded_view(t: torch.Tensor) -> torch.Tensor:
    """The [N, ldc, H, W] view over ALL channel lanes of a row-padded activation (`new_act(..., ldc_align=k)`)."""
    ldc = _nhwc_ldc(t)
    n, c, h, w = t.shape
    return t if ldc == c else t.as_strided((n, ldc, h, w), t.stride())


def _nhwc_ldc(t: torch.Tensor) -> Optional[int]:
    """Pixel stride if `t` (logical NCHW) is NHWC-dense in N,H,W with unit channel stride, else None."""
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        return None
    if w > 1:
        ldc = sw
    elif h > 1:
        ldc = sh
    elif n > 1:
        ldc = sn
    else:
        ldc = c
    if ldc < c:
        return None
    if (w > 1 and sw != ldc) or (h > 1 and sh != w * ldc) or (n > 1 and sn != h * w * ldc):
        return None
    return ldc


def as_act(t: torch.Tensor) -> torch.Tensor:
    """Return `t` itself when its memory is already NHWC(+slice); otherwise one channels-last copy (network boundary)."""
    if t.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got shape {tuple(t.shape)}")
    _require_gpu(t)
    if _nhwc_ldc(t) is not None:
        return t
    n, c, h, w = t.shape
    out = new_act(n, c, h, w, t.dtype, t.device)
    out.copy_(t)
    return out


def desc(t: torch.Tensor, raw: bool = False) -> dsn_tensor:      # (raw: kept for call sites that name a pre-BatchNorm operand)
    _require_gpu(t)
    ldc = _nhwc_ldc(t)
    if ldc is None:
        raise ValueError(f"tensor with shape {tuple(t.shape)} strides {t.stride()} is not an NHWC view")
    if t.dtype not in _DT:
        raise TypeError(f"unsupported dtype {t.dtype} (fp32 and bf16 only)")
    n, c, h, w = t.shape
    return dsn_tensor(t.data_ptr(), _DT[t.dtype], n, h, w, c, ldc)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _ref(d: Optional[dsn_tensor]):
    return None if d is None else C.byref(d)


# ------------------------------------------------------------------------------------------------ convolution
def conv_params(k, stride=1, pad=None, dil=1, act=ACT_NONE, accumulate=False) -> dsn_conv_params:
    kh, kw = (k, k) if isinstance(k, int) else k
    if pad is None:
        pad = kh // 2
    return dsn_conv_params(kh, kw, stride, pad, dil, act, int(accumulate))


def conv_out_hw(h, w, k, stride, pad, dil):
    return ((h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1)


def conv2d_fwd(x, w_packed, bias, residual, y, p: dsn_conv_params):
    L = _lib.lib()
    dx, dy = desc(x), desc(y)
    dr = desc(residual) if residual is not None else None
    _lib.check(L.dsn_conv2d_fwd(C.byref(dx), w_packed.data_ptr(), _p(bias), _ref(dr), C.byref(dy), C.byref(p),
                                stream_ptr()), "conv2d_fwd")
    return y


def conv2d_fwd_acc(x, w_packed, y, p: dsn_conv_params):
    """Training forward of a BN'd convolution WITHOUT the BN + act pass: y = conv(x) and the per-channel fp64 sums of y in a fresh
    accumulator slot (dsn_conv2d_fwd_bnacc).  Returns (acc tensor, acc bytes)."""
    L = _lib.lib()
    dx, dy = desc(x), desc(y)
    acc, nbytes = bn_acc(y.shape[1], y.device)
    _lib.check(L.dsn_conv2d_fwd_bnacc(C.byref(dx), w_packed.data_ptr(), C.byref(dy), C.byref(p), acc.data_ptr(), nbytes,
                                      stream_ptr()), "conv2d_fwd_bnacc")
    return acc, nbytes


def lazy_materialize(x, lz, z, residual=None, lres=None):
    """z = act(bn(x)) per deferred segment of lz [+ residual, itself possibly deferred (lres)]: one elementwise launch."""
    dx, dz = desc(x, raw=True), desc(z)
    dr = desc(residual, raw=True) if residual is not None else None
    _lib.check(_lib.lib().dsn_lazy_materialize(C.byref(dx), C.byref(lz) if lz is not None else None, _ref(dr),
                                               C.byref(lres) if lres is not None else None, C.byref(dz), stream_ptr()),
               "lazy_materialize")
    return z


def bn_finalize_multi(recs):
    """Saved statistics + running averages of every pending BatchNorm (runtime.LazyRec) in one launch per 40 modules."""
    n = len(recs)
    arr = (_lib.dsn_bn_final * n)()
    for i, r in enumerate(recs):
        bn, e, k = r.bn, arr[i], r.hi - r.lo
        e.acc, e.acc_c, e.ch0, e.n, e.count = r.acc.data_ptr(), r.acc_c, r.ch0, k, r.count
        e.gamma, e.beta = _p(bn.weight), _p(bn.bias)
        e.running_mean, e.running_var = _p(bn.running_mean), _p(bn.running_var)
        st, es = r.stats, 4 * r.o0
        e.scale, e.shift, e.mean, e.rstd = (st[0].data_ptr() + es, st[1].data_ptr() + es, st[2].data_ptr() + es,
                                            st[3].data_ptr() + es)
        e.momentum = float(bn.momentum if bn.momentum is not None else 0.03)
        e.eps = float(bn.eps)
    _lib.check(_lib.lib().dsn_bn_finalize_multi(arr, n, stream_ptr()), "bn_finalize_multi")


def _sync_sum(acc_bytes_tensor, nbytes, group):
    """SyncBatchNorm's exchange: SUM the fp64 accumulators over the ranks of `group` (in place, on the current stream)."""
    import torch.distributed as dist
    dist.all_reduce(acc_bytes_tensor[:nbytes].view(torch.float64), op=dist.ReduceOp.SUM, group=group)


def _bn_split(second):
    """dsn_bn_split from (split_c, gamma2, beta2, running_mean2, running_var2, dgamma2, dbeta2) (None entries allowed)."""
    if second is None:
        return None, None
    sc, g2, b2, rm2, rv2, dg2, db2 = second
    st = _lib.dsn_bn_split(int(sc), 0, _p(g2), _p(b2), _p(rm2), _p(rv2), _p(dg2), _p(db2))
    return st, C.byref(st)


def conv2d_fwd_bnstats(x, w_packed, y, p: dsn_conv_params, gamma, beta, running_mean, running_var, momentum, eps, act,
                       residual, z, sync=None, second=None):
    """Training forward of conv -> BatchNorm -> act (+ shortcut) in TWO launches: the conv epilogue adds the per-channel
    sums into fp64 accumulators, the elementwise kernel folds them in its prologue (no statistics pass, no finalize
    launch).  Writes z; returns (scale, shift, mean, rstd) fp32 [C] for the backward pass."""
    L = _lib.lib()
    dx, dy, dz = desc(x), desc(y), desc(z)
    dr = desc(residual) if residual is not None else None
    c = y.shape[1]
    acc, nbytes = bn_acc(c, y.device)
    _lib.check(L.dsn_conv2d_fwd_bnacc(C.byref(dx), w_packed.data_ptr(), C.byref(dy), C.byref(p), acc.data_ptr(), nbytes,
                                      stream_ptr()), "conv2d_fwd_bnacc")
    count = 0.0
    if sync is not None:                # (group, world size): statistics over the global batch (equal per-rank batches)
        _sync_sum(acc, nbytes, sync[0])
        count = float(y.shape[0] * y.shape[2] * y.shape[3]) * sync[1]
    out = torch.empty((4, c), dtype=torch.float32, device=y.device)
    _lib.check(L.dsn_bn_act_fwd_acc(C.byref(dy), acc.data_ptr(), nbytes, count, _p(gamma), _p(beta), _p(running_mean),
                                    _p(running_var), momentum, eps, out[0].data_ptr(), out[1].data_ptr(),
                                    out[2].data_ptr(), out[3].data_ptr(), act, _ref(dr), C.byref(dz),
                                    _bn_split(second)[1], stream_ptr()),
               "bn_act_fwd_acc")
    return out[0], out[1], out[2], out[3]


def bnred(segments):
    """dsn_bnred from [(c0, c1, y, scale, shift, mean, rstd, act, acc, acc_c, ch0)]: the BatchNorm blocks whose dz an input-gradient
    launch completes (y: the block's raw conv output restricted to the segment's channels; scale .. rstd: the segment's slices of
    the saved statistics; acc: the block's backward accumulator, ch0 = accumulator channel of the segment's first channel)."""
    r = _lib.dsn_bnred()
    r.nseg = len(segments)
    for i, (c0, c1, y, scale, shift, mean, rstd, act, acc, acc_c, ch0) in enumerate(segments):
        s = r.seg[i]
        dy_ = desc(y, raw=True)
        s.c0, s.c1, s.ch0, s.acc_c, s.act = c0, c1, ch0, acc_c, act
        s.y, s.yld = dy_.ptr, dy_.ldc
        s.scale, s.shift, s.mean, s.rstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        s.acc = acc.data_ptr()
    return r


def conv2d_dgrad_s2(dy, w_s2, dx, p: dsn_conv_params, red=None):
    """Input gradient of a 3x3 / stride-2 / pad-1 conv through the 2x2 + depth-to-space form (weights: WeightBank.dgrad_s2).
    red: dsn_bnred -- the launch completes dz of those blocks and forms their BatchNorm backward sums in its epilogue."""
    L = _lib.lib()
    a, b = desc(dy), desc(dx)
    if red is not None:
        _lib.check(L.dsn_conv2d_dgrad_s2_bnred(C.byref(a), w_s2.data_ptr(), C.byref(b), C.byref(p), C.byref(red), stream_ptr()),
                   "conv2d_dgrad_s2_bnred")
        return dx
    _lib.check(L.dsn_conv2d_dgrad_s2(C.byref(a), w_s2.data_ptr(), C.byref(b), C.byref(p), stream_ptr()), "conv2d_dgrad_s2")
    return dx


def conv2d_dgrad(dy, w_packed_dgrad, dx, p: dsn_conv_params, residual=None, red=None):
    """dx (+)= conv_transpose(dy, w) (+ residual: a shortcut's gradient, added in the epilogue).  red: see conv2d_dgrad_s2."""
    L = _lib.lib()
    a, b = desc(dy), desc(dx)
    if red is not None:
        r = desc(residual) if residual is not None else None
        _lib.check(L.dsn_conv2d_dgrad_bnred(C.byref(a), w_packed_dgrad.data_ptr(), C.byref(b), C.byref(p), _ref(r), C.byref(red),
                                            stream_ptr()), "conv2d_dgrad_bnred")
        return dx
    if residual is not None:
        r = desc(residual)
        _lib.check(L.dsn_conv2d_dgrad_res(C.byref(a), w_packed_dgrad.data_ptr(), C.byref(b), C.byref(p), C.byref(r),
                                          stream_ptr()), "conv2d_dgrad_res")
        return dx
    _lib.check(L.dsn_conv2d_dgrad(C.byref(a), w_packed_dgrad.data_ptr(), C.byref(b), C.byref(p), stream_ptr()),
               "conv2d_dgrad")
    return dx


_scratch = {}
# Workspaces whose address a captured hipGraph has baked in: a later, larger eager call REPLACES the module-level buffer, and
# the superseded one must then stay alive (and un-reused by the caching allocator) for as long as the graph may replay --
# it is parked here for the life of the process.  (Replays and eager launches share a stream, so the graph keeps using its
# own, old workspace in stream order while eager work moves on to the new one.)
_graph_pinned = set()      # id() of buffers handed out during a capture
_retired = []


def _note_capture(buf):
    if torch.cuda.is_current_stream_capturing():
        _graph_pinned.add(id(buf))
    return buf


def _retire(buf):
    if buf is not None and id(buf) in _graph_pinned:
        _retired.append(buf)


def scratch(nbytes: int, device) -> torch.Tensor:
    """A persistent, growing workspace per (device, stream): kernels on one stream run in order, so every workspace user
    on that stream can share it (no allocator round-trip per call, stable pointers for hipGraph capture)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing() and buf is not None:
            raise RuntimeError("hip_ops.scratch: the workspace would have to grow during hipGraph capture; run the same "
                               "step eagerly once before capturing")
        _retire(buf)
        buf = _scratch[key] = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
    return _note_capture(buf)


def conv2d_wgrad(x, dy, dw, ci, p: dsn_conv_params, oihw: bool = False, queue: "Optional[WgradQueue]" = None):
    """oihw=False: dw packed [Co][KH][KW][ci] (ci >= x channels).  oihw=True: dw is the OIHW fp32 gradient itself
    (ci = real input channels <= x channels); p.accumulate adds into it.
    queue: defer the launch -- the job is planned now and runs with every other queued layer at queue.flush()."""
    L = _lib.lib()
    a, b = desc(x), desc(dy)
    nbytes = L.dsn_conv2d_wgrad_workspace_bytes(C.byref(a), C.byref(b), C.byref(p), ci)
    if queue is not None:
        slab = queue.slab(nbytes) if nbytes else None
        rc = L.dsn_conv2d_wgrad_plan(C.byref(a), C.byref(b), dw.data_ptr(), ci, int(oihw), C.byref(p), _p(slab), nbytes,
                                     queue.next_job_ptr())
        if rc == 0:
            queue.commit(x, dy, dw, slab)
            return dw
        if rc != DSN_EUNSUPPORTED:
            _lib.check(rc, "conv2d_wgrad_plan")
    ws = scratch(nbytes, x.device) if nbytes else None
    _lib.check(L.dsn_conv2d_wgrad(C.byref(a), C.byref(b), dw.data_ptr(), ci, int(oihw), C.byref(p), _p(ws), nbytes,
                                  stream_ptr()), "conv2d_wgrad")
    return dw


DSN_EUNSUPPORTED = -2


class KernelUnsupported(RuntimeError):
    """A fused kernel does not take this shape: the caller runs the unfused launches instead."""


_wgrad_arena = {}          # device -> persistent slab buffer (bump-allocated within a backward pass)
_wgrad_pinned = {}         # device -> {"ring": [...], "events": [...], "next": int, "capture": [...]}


class WgradQueue:
    """Weight-gradient jobs of one backward pass: planned layer by layer, launched together by flush() (three launches for
    the whole network instead of two per layer).  Slabs live in a persistent per-device arena, the job table travels through
    a pinned host buffer (one small H2D copy per pass; under hipGraph capture it becomes a memcpy node of the graph)."""
    CAP = 512
    RING, CAPTURE_POOL = 4, 32

    def __init__(self, device):
        self.device = torch.device(device)
        self.job_bytes = int(_lib.lib().dsn_wgrad_job_bytes())
        self.n = 0
        self.keep = []
        self.offset = 0
        self.extra = 0
        self.host = None

    # -- pinned job table -------------------------------------------------------------------------------------------------
    def _pool(self):
        pool = _wgrad_pinned.get(self.device)
        if pool is None:
            mk = lambda: torch.empty(self.CAP * self.job_bytes, dtype=torch.uint8, pin_memory=True)
            pool = _wgrad_pinned[self.device] = {"ring": [mk() for _ in range(self.RING)], "events": [None] * self.RING,
                                                 "next": 0, "capture": [mk() for _ in range(self.CAPTURE_POOL)]}
        return pool

    def _acquire_host(self):
        pool = self._pool()
        if torch.cuda.is_current_stream_capturing():
            if not pool["capture"]:
                raise RuntimeError("WgradQueue: out of pinned job tables for hipGraph capture")
            self.slot = None
            t = pool["capture"].pop()
            pool.setdefault("owned", []).append(t)     # read by the graph's memcpy node at EVERY replay: never freed or reused
            return t
        i = pool["next"]
        pool["next"] = (i + 1) % self.RING
        ev = pool["events"][i]
        if ev is not None:
            ev.synchronize()                     # the copy that last read this buffer has executed (normally long ago)
        self.slot = i
        return pool["ring"][i]

    def next_job_ptr(self):
        if self.n >= self.CAP:
            raise RuntimeError(f"WgradQueue: more than {self.CAP} layers queued")
        if self.host is None:
            self.host = self._acquire_host()
        return self.host.data_ptr() + self.n * self.job_bytes

    # -- slab arena ---------------------------------------------------------------------------------------------------------
    def slab(self, nbytes):
        n = (nbytes + 255) // 256 * 256
        arena = _wgrad_arena.get(self.device)
        if arena is not None and self.offset + n <= arena.numel():
            _note_capture(arena)
            t = arena[self.offset:self.offset + n]
            self.offset += n
            return t
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("WgradQueue: the slab arena would have to grow during hipGraph capture; run the same step "
                               "eagerly once before capturing")
        self.extra += n                          # does not fit: a one-off buffer now, a larger arena from the next pass on
        t = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.keep.append(t)
        return t

    def commit(self, x, dy, dw, slab):
        self.keep.append((x, dy, dw))
        self.n += 1

    def flush(self):
        """Launch every queued job (the grouped gather launches + the slab reduction)."""
        keep = self.keep
        if self.n:
            L = _lib.lib()
            launch = (C.c_double * 10)()
            _lib.check(L.dsn_conv2d_wgrad_plan_finish(self.host.data_ptr(), self.n, launch), "conv2d_wgrad_plan_finish")
            nb = self.n * self.job_bytes
            dev = torch.empty(nb, dtype=torch.uint8, device=self.device)
            dev.copy_(self.host[:nb], non_blocking=True)
            if self.slot is not None:
                ev = torch.cuda.Event()
                ev.record()
                self._pool()["events"][self.slot] = ev
            _lib.check(L.dsn_conv2d_wgrad_run(dev.data_ptr(), self.n, launch, stream_ptr()), "conv2d_wgrad_run")
        if self.extra and not torch.cuda.is_current_stream_capturing():
            need = self.offset + self.extra      # everything is enqueued on this stream: safe to replace the arena now
            _retire(_wgrad_arena.get(self.device))     # (a captured graph keeps replaying into the arena it was captured with)
            _wgrad_arena[self.device] = torch.empty(int(need * 1.1) + (1 << 20), dtype=torch.uint8, device=self.device)
        self.n, self.keep, self.offset, self.extra, self.host = 0, [], 0, 0, None
        return keep


class WeightBank:
    """All conv weights of a model packed by ONE kernel launch per optimizer step (forward + dgrad layouts)."""

    def __init__(self, convs, ci_pads, dtype, device, need_dgrad=True, co_pads=None, pairs=()):
        """pairs: (convA, convB) 1x1 convolutions of the SAME input that also run as one merged convolution with output
        channels [A | B] (C3's cv2 | cv1): their forward layouts are placed back to back (`pair_fwd[A]` = the merged
        [coA+coB][1][1][ci] matrix) and their dgrad layouts interleave into one [ci][1][1][coA+coB] matrix (`pair_dgrad[A]`)."""
        L = _lib.lib()
        es = 2 if dtype == torch.bfloat16 else 4
        convs, ci_pads = list(convs), list(ci_pads)
        co_pads = list(co_pads) if co_pads is not None else [c.out_channels for c in convs]
        pairs = [(a, b) for a, b in pairs
                 if a.kernel_size == (1, 1) and b.kernel_size == (1, 1) and a.in_channels == b.in_channels
                 and a.bias is None and b.bias is None and a.stride == (1, 1) and b.stride == (1, 1)
                 and ci_pads[convs.index(a)] == a.in_channels and ci_pads[convs.index(b)] == b.in_channels
                 and (a.in_channels * a.out_channels) % 64 == 0]
        second = {id(b) for _, b in pairs}
        follow = {id(a): b for a, b in pairs}
        order = []
        for c in convs:                               # A immediately followed by B, everything else in module order
            if id(c) in second:
                continue
            order.append(c)
            if id(c) in follow:
                order.append(follow[id(c)])
        perm = [convs.index(c) for c in order]
        convs, ci_pads, co_pads = order, [ci_pads[i] for i in perm], [co_pads[i] for i in perm]
        for a, b in pairs:
            co_pads[convs.index(a)] = co_pads[convs.index(b)] = a.out_channels + b.out_channels
        self.ci_pads = ci_pads
        self.dtype, self.convs = dtype, list(convs)
        sizes_f = [c.out_channels * c.kernel_size[0] * c.kernel_size[1] * cp for c, cp in zip(self.convs, ci_pads)]
        self.co_pads = co_pads
        sizes_d = [0 if id(c) in second else c.weight.numel() // c.out_channels * cop for c, cop in zip(self.convs, co_pads)]
        al = lambda n: (n + 63) // 64 * 64
        self.fwd_buf = torch.zeros(sum(al(n) for n in sizes_f), dtype=dtype, device=device)    # ci_pad lanes stay zero
        self.dg_buf = torch.zeros(sum(al(n) for n in sizes_d), dtype=dtype, device=device) if need_dgrad else None    # co_pad lanes stay zero
        descs = (_lib.dsn_pack_desc * len(self.convs))()
        work, self.fwd, self.dgrad, self.dgrad_s2 = [], [], [], []
        vec = 16 // es
        is_s2 = [need_dgrad and c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.dilation == (1, 1)
                 and c.in_channels % vec == 0 and c.out_channels % vec == 0 for c in self.convs]
        sizes_2 = [16 * c.in_channels * c.out_channels if f else 0 for c, f in zip(self.convs, is_s2)]
        self.s2_buf = torch.zeros(max(1, sum(al(n) for n in sizes_2)), dtype=dtype, device=device)   # unused blocks stay zero
        of = od = o2 = 0
        for i, (c, cp) in enumerate(zip(self.convs, ci_pads)):
            co, ci, kh, kw = c.weight.shape
            fv = self.fwd_buf[of:of + sizes_f[i]].view(co, kh, kw, cp)
            if id(c) in second:       # B of a pair: its dgrad columns live inside A's merged matrix, after A's coA columns
                prev = self.convs[i - 1]
                dv = None
                dptr = self.dgrad[i - 1].data_ptr() + prev.out_channels * es if need_dgrad else None
            else:
                dv = self.dg_buf[od:od + sizes_d[i]].view(ci, kh, kw, co_pads[i]) if need_dgrad else None
                dptr = dv.data_ptr() if need_dgrad else None
            sv = self.s2_buf[o2:o2 + sizes_2[i]].view(4 * ci, 2, 2, co) if is_s2[i] else None
            self.fwd.append(fv)
            self.dgrad.append(dv)
            self.dgrad_s2.append(sv)
            w = c.weight
            assert w.dtype == torch.float32 and w.is_contiguous()
            descs[i] = _lib.dsn_pack_desc(w.data_ptr(), fv.data_ptr(), dptr,
                                          sv.data_ptr() if sv is not None else None, co, ci, kh, kw, cp, co_pads[i])
            work += [(i, t) for t in range(L.dsn_pack_tiles(co, ci, kh, kw))]
            of += al(sizes_f[i])
            od += al(sizes_d[i])
            o2 += al(sizes_2[i])
        raw = bytes(descs)
        self.descs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.work_dev = torch.tensor(work, dtype=torch.int32).to(device)
        self.n_work = len(work)
        self.ptrs = tuple(c.weight.data_ptr() for c in self.convs)
        self.pair_fwd, self.pair_dgrad = {}, {}
        for a, b in pairs:
            i = self.convs.index(a)
            fa, fb = self.fwd[i], self.fwd[i + 1]
            assert fb.data_ptr() == fa.data_ptr() + fa.numel() * es, "pair forward layouts must be back to back"
            n = fa.numel() + fb.numel()
            start = (fa.data_ptr() - self.fwd_buf.data_ptr()) // es
            self.pair_fwd[a] = (b, self.fwd_buf[start:start + n].view(a.out_channels + b.out_channels, 1, 1, a.in_channels))
            self.pair_dgrad[a] = self.dgrad[i] if need_dgrad else None

    def valid_for(self, dtype):
        return dtype == self.dtype and self.ptrs == tuple(c.weight.data_ptr() for c in self.convs)

    def pack(self):
        _lib.check(_lib.lib().dsn_pack_weights_multi(self.descs_dev.data_ptr(), self.work_dev.data_ptr(), self.n_work,
                                                     _DT[self.dtype], stream_ptr()), "pack_weights_multi")


def pack_weight_fwd(w_oihw: torch.Tensor, dtype, scale: Optional[torch.Tensor] = None, ci_pad: Optional[int] = None):
    co, ci, kh, kw = w_oihw.shape
    ci_pad = ci if ci_pad is None else ci_pad
    w = w_oihw.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    out = torch.empty((co, kh, kw, ci_pad), dtype=dtype, device=w.device)
    _lib.check(_lib.lib().dsn_pack_weight_fwd(w.data_ptr(), _p(scale), out.data_ptr(), _DT[dtype], co, ci, kh, kw, ci_pad,
                                              stream_ptr()), "pack_weight_fwd")
    return out


def pack_weight_dgrad(w_oihw: torch.Tensor, dtype, co_pad: Optional[int] = None):
    co, ci, kh, kw = w_oihw.shape
    co_pad = co if co_pad is None else co_pad
    w = w_oihw.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    out = torch.empty((ci, kh, kw, co_pad), dtype=dtype, device=w.device)
    _lib.check(_lib.lib().dsn_pack_weight_dgrad(w.data_ptr(), out.data_ptr(), _DT[dtype], co, ci, kh, kw, co_pad,
                                                stream_ptr()), "pack_weight_dgrad")
    return out


def unpack_wgrad(dw_packed, shape_oihw, ci_pad, out: Optional[torch.Tensor] = None):
    co, ci, kh, kw = shape_oihw
    acc = out is not None
    if out is None:
        out = torch.empty(shape_oihw, dtype=torch.float32, device=dw_packed.device)
    _lib.check(_lib.lib().dsn_unpack_wgrad(dw_packed.data_ptr(), out.data_ptr(), co, ci, kh, kw, ci_pad, int(acc),
                                           stream_ptr()), "unpack_wgrad")
    return out


# ------------------------------------------------------------------------------------------------ BN + act
_ws_cache = {}


def _bn_ws(c, device):
    """dsn_bn_stats workspace: zero-filled ONCE (the finalize kernel restores the zeros)."""
    key = (c, device)
    if key not in _ws_cache:
        nbytes = _lib.lib().dsn_bn_workspace_bytes(c)
        _ws_cache[key] = (torch.zeros(nbytes, dtype=torch.uint8, device=device), nbytes)
    return _ws_cache[key]


class _BnArena:
    """Zeroed fp64 accumulator slots for the fused BatchNorm kernels (zero on entry, left dirty).  One buffer per device,
    cleared by ONE memset at the start of a training step (`bn_arena_begin`, part of the captured graph) and handed out
    slot by slot -- the same sequence of addresses every step, as hipGraph replay needs."""
    BYTES = 32 << 20

    def __init__(self, device):
        self.buf = torch.zeros(self.BYTES, dtype=torch.uint8, device=device)
        self.cursor = 0
        self.active = False            # no slots until the first begin()
        self.high = 0                  # high-water mark of handed-out bytes: a captured clear must cover every later replay

    def begin(self):
        # only what the previous step handed out is dirty (the rest is still zero from the allocation / the last clear)
        self.high = max(self.high, (self.cursor + 15) // 16 * 16)
        if self.high:
            zero_(self.buf[:self.high])
        self.cursor = 0
        self.active = True

    def take(self, nbytes):
        n = (nbytes + 255) // 256 * 256
        if not self.active or self.cursor + n > self.BYTES:
            return None
        if torch.cuda.is_current_stream_capturing() and self.cursor + n > self.high:
            # the clear recorded at the start of this capture covers [0, high): a slot beyond it would never be re-zeroed
            raise RuntimeError("BatchNorm accumulator arena grew during hipGraph capture; run the same step eagerly first")
        t = self.buf[self.cursor:self.cursor + n]
        self.cursor += n
        return t


_arenas = {}


def bn_arena_begin(device):
    """Call once at the start of a step (Model does): clears the accumulator arena with one memset."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    a = _arenas.get(device)
    if a is None:
        a = _arenas[device] = _BnArena(device)
    a.begin()


def bn_arena_cursor(device, set_to=None):
    """Read (or rewind) the step arena's hand-out position.  A backward pass captured in several graphs (graph.py, world_size > 1)
    resumes the SECOND half of micro-batch k at the position its first half left, so that both "first" and "next" micro-batch
    graphs use the slots their own arena clear covers."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    a = _arenas.get(device)
    if a is None:
        return 0
    if set_to is not None:
        a.cursor = int(set_to)
    return a.cursor


def bn_acc(c, device):
    """A ZEROED accumulator slot for c channels: from the step arena when one is active, else a fresh zero-filled tensor."""
    nbytes = _lib.lib().dsn_bn_workspace_bytes(c)
    a = _arenas.get(device)
    t = a.take(nbytes) if a is not None else None
    if t is None:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    return t, nbytes


def bn_stats(y, gamma, beta, running_mean, running_var, momentum, eps):
    """Returns (scale, shift, mean, rstd), each fp32 [C]; updates running stats in place when given."""
    c = y.shape[1]
    out = torch.empty((4, c), dtype=torch.float32, device=y.device)
    ws, nbytes = _bn_ws(c, y.device)
    d = desc(y)
    _lib.check(_lib.lib().dsn_bn_stats(C.byref(d), _p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum,
                                       eps, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                       ws.data_ptr(), nbytes, stream_ptr()), "bn_stats")
    return out[0], out[1], out[2], out[3]


def channel_sum(t, out=None, accumulate=False):
    """out[c] (+)= sum over N,H,W of an NHWC activation (fp32 [C]): bias gradients."""
    _require_gpu(t)
    c = t.shape[1]
    if out is None:
        out, accumulate = torch.empty(c, dtype=torch.float32, device=t.device), False
    ws, nbytes = _bn_ws(c, t.device)
    d = desc(t)
    _lib.check(_lib.lib().dsn_channel_sum(C.byref(d), out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, stream_ptr()),
               "channel_sum")
    return out


def bn_act_fwd(y, scale, shift, act, residual, z):
    a, b = desc(y), desc(z)
    r = desc(residual) if residual is not None else None
    _lib.check(_lib.lib().dsn_bn_act_fwd(C.byref(a), _p(scale), _p(shift), act, _ref(r), C.byref(b), stream_ptr()),
               "bn_act_fwd")
    return z


def bn_act_bwd_reduce(dz, y, scale, shift, mean, rstd, act, ws):
    """First half of bn_act_bwd: the two per-channel sums (sum g, sum g * yhat) added into the ZEROED accumulator `ws`."""
    a, b = desc(dz), desc(y, raw=True)
    _lib.check(_lib.lib().dsn_bn_act_bwd_reduce(C.byref(a), C.byref(b), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), act, ws.data_ptr(), ws.numel() * ws.element_size(), stream_ptr()),
               "bn_act_bwd_reduce")
    return ws


def bn_act_bwd(dz, y, scale, shift, mean, rstd, act, dy, dgamma, dbeta, accumulate=False, sync=None, second=None, pre=None):
    """pre = (accumulator, [(k0, k1)]): the sums of channels [k0, k1) are already in `accumulator` -- they were formed in the epilogue
    of the input-gradient convolution that completed dz (conv2d_dgrad(red=...)); only the remaining channels are reduced here."""
    a, b, c = desc(dz), desc(y, raw=True), desc(dy)      # (y is the raw conv output by definition, deferred or not)
    if pre is not None and sync is None:
        L = _lib.lib()
        ws, cov = pre
        nbytes = ws.numel() * ws.element_size()
        ctot, k = y.shape[1], 0
        for k0, k1 in sorted(cov) + [(ctot, ctot)]:
            if k0 > k:      # channels [k, k0) were not covered by a fused epilogue
                da, db_ = desc(dz[:, k:k0]), desc(y[:, k:k0], raw=True)
                _lib.check(L.dsn_bn_act_bwd_reduce_into(C.byref(da), C.byref(db_), scale[k:k0].data_ptr(), shift[k:k0].data_ptr(),
                                                        mean[k:k0].data_ptr(), rstd[k:k0].data_ptr(), act, ws.data_ptr(), ctot, k,
                                                        stream_ptr()), "bn_act_bwd_reduce_into")
            k = max(k, k1)
        _lib.check(L.dsn_bn_act_bwd_apply(C.byref(a), C.byref(b), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), act, C.byref(c), _p(dgamma), _p(dbeta), int(accumulate),
                                          ws.data_ptr(), nbytes, 0.0, 1.0, _bn_split(second)[1] if second is not None else None,
                                          stream_ptr()), "bn_act_bwd_apply")
        return dy
    ws, nbytes = bn_acc(y.shape[1], y.device)
    if sync is not None or second is not None:   # SyncBatchNorm: global sums for dy, per-rank share of dgamma / dbeta
        L = _lib.lib()
        _lib.check(L.dsn_bn_act_bwd_reduce(C.byref(a), C.byref(b), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), act, ws.data_ptr(), nbytes, stream_ptr()), "bn_act_bwd_reduce")
        count, share = 0.0, 1.0
        if sync is not None:
            _sync_sum(ws, nbytes, sync[0])
            count, share = float(y.shape[0] * y.shape[2] * y.shape[3]) * sync[1], 1.0 / sync[1]
        _lib.check(L.dsn_bn_act_bwd_apply(C.byref(a), C.byref(b), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), act, C.byref(c), _p(dgamma), _p(dbeta), int(accumulate),
                                          ws.data_ptr(), nbytes, count, share, _bn_split(second)[1], stream_ptr()),
                   "bn_act_bwd_apply")
        return dy
    _lib.check(_lib.lib().dsn_bn_act_bwd(C.byref(a), C.byref(b), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), act, C.byref(c), _p(dgamma), _p(dbeta), int(accumulate),
                                         ws.data_ptr(), nbytes, stream_ptr()), "bn_act_bwd")
    return dy


def act_bwd(dz, y, act, dy):
    a, b, c = desc(dz), desc(y), desc(dy)
    _lib.check(_lib.lib().dsn_act_bwd(C.byref(a), C.byref(b), act, C.byref(c), stream_ptr()), "act_bwd")
    return dy


# ------------------------------------------------------------------------------------------------ data movement
def focus_s2d(x_nchw: torch.Tensor, y):
    """Focus slicing of an NCHW image batch.  fp32 input: already normalised; uint8 input (what the reference's loader
    yields): `imgs.float() / 255.0` (train.py:329) is folded into the same launch."""
    _require_gpu(x_nchw)
    x = x_nchw
    d = desc(y)
    if x.dtype == torch.uint8:
        x = x if x.is_contiguous() else x.contiguous()
        n, c, h, w = x.shape
        _lib.check(_lib.lib().dsn_focus_s2d_u8(x.data_ptr(), n, c, h, w, C.byref(d), stream_ptr()), "focus_s2d_u8")
        return y
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    n, c, h, w = x.shape
    _lib.check(_lib.lib().dsn_focus_s2d(x.data_ptr(), n, c, h, w, C.byref(d), stream_ptr()), "focus_s2d")
    return y


def maxpool_s1(x, y, k, idx: Optional[torch.Tensor] = None):
    a, b = desc(x), desc(y)
    _lib.check(_lib.lib().dsn_maxpool_s1(C.byref(a), C.byref(b), _p(idx), k, stream_ptr()), "maxpool_s1")
    return y


def maxpool_s1_bwd(dy, idx, dx, k, accumulate=False):
    a, b = desc(dy), desc(dx)
    _lib.check(_lib.lib().dsn_maxpool_s1_bwd(C.byref(a), idx.data_ptr(), C.byref(b), k, int(accumulate), stream_ptr()),
               "maxpool_s1_bwd")
    return dx


def maxpool_s1_multi(x, ys, ks, idxs=None):
    """Stride-1 max pools of one input at several window sizes in one launch (SPP).  idxs: list of int32 tensors or None."""
    n = len(ys)
    a = desc(x)
    arr = (dsn_tensor * n)(*[desc(t) for t in ys])
    kk = (C.c_int32 * n)(*[int(k) for k in ks])
    ip = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in (idxs or [None] * n)])
    _lib.check(_lib.lib().dsn_maxpool_s1_multi(C.byref(a), C.cast(arr, C.c_void_p), C.cast(ip, C.c_void_p),
                                               C.cast(kk, C.c_void_p), n, stream_ptr()), "maxpool_s1_multi")
    return ys


def maxpool_s1_bwd_multi(dys, idxs, ks, dx, accumulate=False):
    """dx (+)= sum_i maxpool_s1_bwd(dys[i], idxs[i], ks[i]) in one pass (SPP: three pools of one input)."""
    n = len(dys)
    arr = (dsn_tensor * n)(*[desc(t) for t in dys])
    ip = (C.c_void_p * n)(*[t.data_ptr() for t in idxs])
    kk = (C.c_int32 * n)(*[int(k) for k in ks])
    b = desc(dx)
    _lib.check(_lib.lib().dsn_maxpool_s1_bwd_multi(C.cast(arr, C.c_void_p), C.cast(ip, C.c_void_p), C.cast(kk, C.c_void_p), n,
                                                   C.byref(b), int(accumulate), stream_ptr()), "maxpool_s1_bwd_multi")
    return dx


def upsample_nearest2x(x, y):
    a, b = desc(x), desc(y)
    _lib.check(_lib.lib().dsn_upsample_nearest2x(C.byref(a), C.byref(b), stream_ptr()), "upsample_nearest2x")
    return y


def upsample_nearest2x_bwd(dy, dx, accumulate=False):
    a, b = desc(dy), desc(dx)
    _lib.check(_lib.lib().dsn_upsample_nearest2x_bwd(C.byref(a), C.byref(b), int(accumulate), stream_ptr()),
               "upsample_nearest2x_bwd")
    return dx


def _desc_nchw(t: torch.Tensor) -> dsn_tensor:
    """Descriptor of a CONTIGUOUS NCHW fp32 tensor (seg logits / their gradient); ldc is unused by NCHW kernels."""
    _require_gpu(t)
    assert t.is_contiguous() and t.dtype == torch.float32
    n, c, h, w = t.shape
    return dsn_tensor(t.data_ptr(), DSN_F32, n, h, w, c, c)


def bilinear_ac(x, y, out_nchw=False):
    a = desc(x)
    b = _desc_nchw(y) if out_nchw else desc(y)
    _lib.check(_lib.lib().dsn_bilinear_ac(C.byref(a), C.byref(b), int(out_nchw), stream_ptr()), "bilinear_ac")
    return y


def _wr_ws(nseg, rows, c, device):
    nbytes = _lib.lib().dsn_window_reduce_workspace_bytes(nseg, rows, c)
    return scratch(max(nbytes, 16), device), nbytes


def bilinear_ac_bwd(dy, dx, dy_nchw=False, accumulate=False):
    a = _desc_nchw(dy) if dy_nchw else desc(dy)
    b = desc(dx)
    ws, nbytes = None, 0
    if not dy_nchw and dx.shape[2] * dx.shape[3] <= 64:
        ws, nbytes = _wr_ws(dx.shape[0] * dx.shape[2] * dx.shape[3], dy.shape[2], dx.shape[1], dx.device)
    _lib.check(_lib.lib().dsn_bilinear_ac_bwd(C.byref(a), int(dy_nchw), C.byref(b), int(accumulate), _p(ws), nbytes,
                                              stream_ptr()), "bilinear_ac_bwd")
    return dx


def adaptive_avgpool(x, y):
    a, b = desc(x), desc(y)
    ws, nbytes = _wr_ws(y.shape[0] * y.shape[2] * y.shape[3], x.shape[2], x.shape[1], x.device)
    _lib.check(_lib.lib().dsn_adaptive_avgpool(C.byref(a), C.byref(b), ws.data_ptr(), nbytes, stream_ptr()),
               "adaptive_avgpool")
    return y


def _wr_multi_ws(smalls, rows_list, cs, device):
    L = _lib.lib()
    nbytes = sum(int(L.dsn_window_reduce_workspace_bytes(t.shape[0] * t.shape[2] * t.shape[3], r, c))
                 for t, r, c in zip(smalls, rows_list, cs))
    return scratch(nbytes, device), nbytes


def _vec16(t) -> bool:
    """16-byte channel vectors possible: channel count and pixel stride multiples of the vector, base aligned."""
    vw = 4 if t.dtype == torch.float32 else 8
    ldc = _nhwc_ldc(t)
    return ldc is not None and t.shape[1] % vw == 0 and ldc % vw == 0 and t.data_ptr() % 16 == 0


def adaptive_avgpool_multi(x, ys):
    """ys[j] = adaptive average pool of x to ys[j]'s size, all (<= 4) in one reduction + one finalize launch."""
    n = len(ys)
    if n > 4 or not _vec16(x):
        for y in ys:
            adaptive_avgpool(x, y)
        return ys
    arr = (dsn_tensor * n)(*[desc(t) for t in ys])
    a = desc(x)
    ws, nbytes = _wr_multi_ws(ys, [x.shape[2]] * n, [x.shape[1]] * n, x.device)
    _lib.check(_lib.lib().dsn_adaptive_avgpool_multi(C.byref(a), arr, n, ws.data_ptr(), nbytes, stream_ptr()),
               "adaptive_avgpool_multi")
    return ys


def bilinear_ac_multi(xs, ys):
    """ys[j] = bilinear(align_corners=True) of xs[j]; all ys share one spatial size (slices of a concat buffer)."""
    n = len(xs)
    if n > 4 or not all(_vec16(t) for t in list(xs) + list(ys)):
        for x, y in zip(xs, ys):
            bilinear_ac(x, y)
        return ys
    ax = (dsn_tensor * n)(*[desc(t) for t in xs])
    ay = (dsn_tensor * n)(*[desc(t) for t in ys])
    _lib.check(_lib.lib().dsn_bilinear_ac_multi(ax, ay, n, stream_ptr()), "bilinear_ac_multi")
    return ys


def bilinear_ac_bwd_multi(dys, dxs, accumulate=False):
    """dxs[j] (+)= backward of bilinear_ac for small sources (<= 64 pixels each), one reduction + one finalize launch."""
    n = len(dys)
    if n > 4 or not all(_vec16(t) for t in dys) or any(t.shape[2] * t.shape[3] > 64 for t in dxs):
        for dy, dx in zip(dys, dxs):
            bilinear_ac_bwd(dy, dx, accumulate=accumulate)
        return dxs
    ay = (dsn_tensor * n)(*[desc(t) for t in dys])
    ax = (dsn_tensor * n)(*[desc(t) for t in dxs])
    ws, nbytes = _wr_multi_ws(dxs, [t.shape[2] for t in dys], [t.shape[1] for t in dxs], dxs[0].device)
    _lib.check(_lib.lib().dsn_bilinear_ac_bwd_multi(ay, ax, n, int(accumulate), ws.data_ptr(), nbytes, stream_ptr()),
               "bilinear_ac_bwd_multi")
    return dxs


def adaptive_avgpool_bwd(dy, dx, accumulate=False):
    a, b = desc(dy), desc(dx)
    _lib.check(_lib.lib().dsn_adaptive_avgpool_bwd(C.byref(a), C.byref(b), int(accumulate), stream_ptr()),
               "adaptive_avgpool_bwd")
    return dx


def adaptive_avgpool_bwd_multi(dys, dx, accumulate=False):
    """dx (+)= sum_i adaptive_avgpool_bwd(dys[i]) in one pass (PyramidPooling: four grids pool the same input)."""
    n = len(dys)
    arr = (dsn_tensor * n)(*[desc(t) for t in dys])
    b = desc(dx)
    _lib.check(_lib.lib().dsn_adaptive_avgpool_bwd_multi(C.cast(arr, C.c_void_p), n, C.byref(b), int(accumulate),
                                                         stream_ptr()), "adaptive_avgpool_bwd_multi")
    return dx


def copy(x, y, accumulate=False):
    a, b = desc(x), desc(y)
    _lib.check(_lib.lib().dsn_copy(C.byref(a), C.byref(b), int(accumulate), stream_ptr()), "copy")
    return y


def ffm_scale(feat, att, out):
    a, b, c = desc(feat), desc(att), desc(out)
    _lib.check(_lib.lib().dsn_ffm_scale(C.byref(a), C.byref(b), C.byref(c), stream_ptr()), "ffm_scale")
    return out


def ffm_scale_bwd(dout, feat, att, dfeat, datt, accumulate=False):
    a, b, c, d, e = desc(dout), desc(feat), desc(att), desc(dfeat), desc(datt)
    ws, nbytes = _wr_ws(feat.shape[0], feat.shape[2], feat.shape[1], feat.device)
    _lib.check(_lib.lib().dsn_ffm_scale_bwd(C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), int(accumulate),
                                            ws.data_ptr(), nbytes, stream_ptr()), "ffm_scale_bwd")
    return dfeat, datt


# ------------------------------------------------------------------------------------------------ detect / nms
def detect_decode(t, raw, pred, row_offset, na, no, stride, anchors_px):
    d = desc(t)
    total = 0 if pred is None else pred.shape[1]
    _lib.check(_lib.lib().dsn_detect_decode(C.byref(d), raw.data_ptr(), _p(pred), total, row_offset, na, no,
                                            float(stride), _p(anchors_px), stream_ptr()), "detect_decode")


def detect_decode_multi(ts, raws, pred, row_offsets, na, no, strides, anchors_px):
    """All Detect levels in one launch: ts[l] head outputs -> raws[l] (+ the decoded rows of pred when given)."""
    nl = len(ts)
    arr = (dsn_tensor * nl)(*[desc(t) for t in ts])
    rp = (C.c_void_p * nl)(*[r.data_ptr() for r in raws])
    ro = (C.c_int64 * nl)(*[int(v) for v in row_offsets])
    st = (C.c_float * nl)(*[float(v) for v in strides])
    total = 0 if pred is None else pred.shape[1]
    _lib.check(_lib.lib().dsn_detect_decode_multi(arr, rp, nl, _p(pred), total, ro, na, no, st, _p(anchors_px), stream_ptr()),
               "detect_decode_multi")


def detect_head_fwd_supported(dtype, na, no, channels) -> bool:
    """Can the three head convolutions + permute of a training forward run as ONE launch (detect_head_fwd)?"""
    import math
    k = 0
    for c in channels:
        k = math.gcd(k, int(c))
    return bool(_lib.lib().dsn_detect_head_fwd_supported(_DT[dtype], na, no, k))


def detect_head_fwd(xs, ws, biases, raws, na, no):
    """raws[l] = permute(conv1x1(xs[l], ws[l]) + biases[l]) for every Detect level, one launch (training forward)."""
    nl = len(xs)
    arr = (dsn_tensor * nl)(*[desc(x) for x in xs])
    wp = (C.c_void_p * nl)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * nl)(*[(b.data_ptr() if b is not None else None) for b in biases])
    rp = (C.c_void_p * nl)(*[r.data_ptr() for r in raws])
    _lib.check(_lib.lib().dsn_detect_head_fwd_multi(arr, wp, bp, rp, nl, na, no, stream_ptr()), "detect_head_fwd_multi")


_det_ws = {}


def detect_raw_bwd_multi(draws, dts, na, no, bias_grads=None):
    """dts[l] (row padding zero-filled) from the raw-output gradients of every level in one launch; bias_grads[l] += the heads'
    bias gradients (one more tiny launch for all levels)."""
    nl = len(dts)
    gs = [d if (d.is_contiguous() and d.dtype == torch.float32) else d.float().contiguous() for d in draws]
    arr = (dsn_tensor * nl)(*[desc(t) for t in dts])
    dp = (C.c_void_p * nl)(*[g.data_ptr() for g in gs])
    zp = (C.c_int32 * nl)(*[int(a.ldc) for a in arr])
    bp, ws, nbytes = None, None, 0
    if bias_grads is not None:
        bp = (C.c_void_p * nl)(*[b.data_ptr() for b in bias_grads])
        nbytes = nl * 512 * na * no * 4
        key = (dts[0].device, nbytes)
        ws = _det_ws.get(key)
        if ws is None:
            ws = _det_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dts[0].device)
    _lib.check(_lib.lib().dsn_detect_raw_bwd_multi(dp, arr, nl, na, no, zp, bp, _p(ws), nbytes, stream_ptr()),
               "detect_raw_bwd_multi")
    return dts


def detect_raw_bwd(draw, dt, na, no, zero_padding=False):
    """zero_padding: dt is a row-padded activation (new_act ldc_align) whose padding lanes must read as zeros."""
    d = desc(dt)
    g = draw if (draw.is_contiguous() and draw.dtype == torch.float32) else draw.float().contiguous()
    _lib.check(_lib.lib().dsn_detect_raw_bwd(g.data_ptr(), C.byref(d), na, no, int(d.ldc) if zero_padding else 0,
                                             stream_ptr()), "detect_raw_bwd")
    return dt


def nms(pred: torch.Tensor, conf_thres, iou_thres, multi_label=False, agnostic=False, classes=None, max_det=300):
    """Returns (out [bs, max_det, 6] fp32, count [bs] int32), both on the device."""
    _require_gpu(pred)
    p = pred if (pred.dtype == torch.float32 and pred.is_contiguous()) else pred.float().contiguous()
    bs, n, no = p.shape
    nc = no - 5
    mask = 0
    if classes is not None:
        for c in classes:
            mask |= 1 << int(c)
        if mask == 0:
            mask = 1 << 63 if nc < 64 else 0   # empty filter keeps nothing
    L = _lib.lib()
    nbytes = L.dsn_nms_workspace_bytes(bs, n, nc, int(multi_label))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    out = torch.zeros((bs, max_det, 6), dtype=torch.float32, device=p.device)
    cnt = torch.zeros((bs,), dtype=torch.int32, device=p.device)
    _lib.check(L.dsn_nms(p.data_ptr(), bs, n, nc, float(conf_thres), float(iou_thres), int(multi_label), int(agnostic),
                         mask, max_det, out.data_ptr(), cnt.data_ptr(), ws.data_ptr(), nbytes, stream_ptr()), "nms")
    return out, cnt


def zero_(t: torch.Tensor) -> torch.Tensor:
    """t.zero_() as a library launch (keeps ATen fill kernels / memset nodes out of the captured step).  t: contiguous, a
    multiple of 4 bytes."""
    _require_gpu(t)
    nbytes = t.numel() * t.element_size()
    if not t.is_contiguous() or nbytes % 4 or t.data_ptr() % 4:
        return t.zero_()
    _lib.check(_lib.lib().dsn_fill32(t.data_ptr(), 0, nbytes // 4, stream_ptr()), "fill32")
    return t


def copy_multi(pairs):
    """[(dst, src | None)] flat copies in one launch: dst[:src.numel()] = src, the rest of dst zero (dst, src: contiguous tensors of
    one dtype on the device; src None = clear dst).  At most 4 pairs."""
    segs = (_lib.dsn_copy_seg * len(pairs))()
    for i, (dst, src) in enumerate(pairs):
        if not dst.is_contiguous() or (src is not None and (not src.is_contiguous() or src.dtype != dst.dtype or src.device != dst.device)):
            raise ValueError("copy_multi: contiguous tensors of one dtype on one device")
        nb = src.numel() * src.element_size() if src is not None else 0
        segs[i] = _lib.dsn_copy_seg(dst.data_ptr(), src.data_ptr() if src is not None and nb else None, nb,
                                    dst.numel() * dst.element_size())
    _lib.check(_lib.lib().dsn_copy_multi(segs, len(pairs), stream_ptr()), "copy_multi")


def add_i64_(t: torch.Tensor, value: int) -> torch.Tensor:
    _require_gpu(t)
    assert t.dtype == torch.int64 and t.is_contiguous()
    _lib.check(_lib.lib().dsn_add_i64(t.data_ptr(), t.numel(), int(value), stream_ptr()), "add_i64")
    return t


def cast(src: torch.Tensor, dtype) -> torch.Tensor:
    s = src.detach()
    if s.dtype != torch.float32 or not s.is_contiguous():
        s = s.float().contiguous()
    out = torch.empty(s.shape, dtype=dtype, device=s.device)
    _lib.check(_lib.lib().dsn_cast(s.data_ptr(), out.data_ptr(), _DT[dtype], s.numel(), stream_ptr()), "cast")
    return out


# ------------------------------------------------------------------------------------------------ live profiler
def profile_enable(on: bool):
    _lib.check(_lib.lib().dsn_profile_enable(int(on)), "profile_enable")


last_event_pair_overhead_us = 0.0


def profile_collect(by_layer: bool = False):
    """{label: dict(launches, ms, flops, bytes)} for every kernel launched since profile_enable(True); label = rocprofv3 symbol
    family / dtype / tile / direction for the convolutions.  by_layer=True: (that dict, {(label, layer): dict(...)})."""
    L = _lib.lib()
    need = int(L.dsn_profile_dump(None, 0))
    if need < 0:
        _lib.check(-need, "profile_dump")
    buf = C.create_string_buffer(max(need, 1))
    rc = int(L.dsn_profile_dump(C.cast(buf, C.c_void_p), need))
    if rc < 0:
        _lib.check(-rc, "profile_dump")
    out, layers = {}, {}
    global last_event_pair_overhead_us
    for line in buf.value.decode().splitlines():
        label, layer, n, ms, fl, by = line.split("\t")
        if label == "__event_pair_overhead":       # what the library subtracted from every record (one queue marker)
            last_event_pair_overhead_us = float(ms) * 1e3
            continue
        rec = dict(launches=int(float(n)), ms=float(ms), flops=float(fl), bytes=float(by))
        layers[(label, layer)] = rec
        a = out.setdefault(label, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        for k in rec:
            a[k] += rec[k]
    return (out, layers) if by_layer else out


# ------------------------------------------------------------------------------------------------ PyramidPooling branches
def pp_stages_supported(max_pixels: int, c: int, co: int, dtype) -> bool:
    """Can dsn_pp_stages_fwd / _bwd take branches of up to max_pixels pooled pixels, c -> co channels (bf16, fits LDS)?"""
    if dtype != torch.bfloat16:
        return False
    return bool(_lib.lib().dsn_pp_stages_supported(int(max_pixels), int(c), int(co), DSN_BF16))


def _pp_args(stages, c, co, act, accumulate=0, momentum=0.0, eps=0.0):
    a = _lib.dsn_pp_args()
    a.nstage, a.C, a.Co, a.dtype, a.act, a.accumulate = len(stages), c, co, DSN_BF16, act, int(accumulate)
    a.momentum, a.eps = float(momentum), float(eps)
    for j, st in enumerate(stages):
        for k, v in st.items():
            setattr(a.s[j], k, v)
    return a


def pp_stages_fwd(xs, ws, bns, zs, ys, stats, act, momentum, eps):
    """xs[j]: pooled map [n, c, k, k] (contiguous NHWC), ws[j]: forward-packed [co, c] weights, bns[j]: the branch's BatchNorm2d or
    None (1x1 map), zs / ys[j]: raw and activated outputs [n, co, k, k], stats[j]: fp32 [4, co] or None.  One launch."""
    n, c = xs[0].shape[0], xs[0].shape[1]
    co = zs[0].shape[1]
    stages = []
    for x, w, bn, z, y, st in zip(xs, ws, bns, zs, ys, stats):
        _require_gpu(x)
        if _nhwc_ldc(x) != c:
            raise ValueError("pp_stages_fwd: the pooled maps are staged as dense [pixels][C] rows")
        stages.append(dict(x=x.data_ptr(), w=w.data_ptr(), z=z.data_ptr(), y=y.data_ptr(), stats=_p(st),
                           gamma=_p(bn.weight) if bn is not None else None, beta=_p(bn.bias) if bn is not None else None,
                           running_mean=_p(bn.running_mean) if bn is not None else None,
                           running_var=_p(bn.running_var) if bn is not None else None,
                           zld=_nhwc_ldc(z), yld=_nhwc_ldc(y), P=x.shape[0] * x.shape[2] * x.shape[3], has_bn=int(bn is not None)))
    a = _pp_args(stages, c, co, act, 0, momentum, eps)
    _lib.check(_lib.lib().dsn_pp_stages_fwd(C.byref(a), stream_ptr()), "pp_stages_fwd")


def pp_stages_bwd(xs, ws, zs, dys, dxs, stats, dgammas, dbetas, dws, act, accumulate):
    """Backward of pp_stages_fwd: dxs[j] [n, c, k, k] written; dgammas / dbetas / dws[j] fp32 (+= when accumulate)."""
    c, co = xs[0].shape[1], zs[0].shape[1]
    stages = []
    for x, w, z, dy, dx, st, dg, db, dw in zip(xs, ws, zs, dys, dxs, stats, dgammas, dbetas, dws):
        if _nhwc_ldc(x) != c or _nhwc_ldc(dx) != c:
            raise ValueError("pp_stages_bwd: x and dx are dense [pixels][C] rows")
        stages.append(dict(x=x.data_ptr(), w=w.data_ptr(), z=z.data_ptr(), dy=dy.data_ptr(), dx=dx.data_ptr(), stats=_p(st),
                           dgamma=_p(dg), dbeta=_p(db), dw=_p(dw), zld=_nhwc_ldc(z), dyld=_nhwc_ldc(dy),
                           P=x.shape[0] * x.shape[2] * x.shape[3], has_bn=int(st is not None)))
    a = _pp_args(stages, c, co, act, accumulate)
    _lib.check(_lib.lib().dsn_pp_stages_bwd(C.byref(a), stream_ptr()), "pp_stages_bwd")


# ------------------------------------------------------------------------------------------------ losses
def det_loss(p, targets, anchors_host, balance, h_box, h_obj, h_cls, cls_pw, obj_pw, anchor_t, cp, cn, nc, gain=1.0,
             fl_gamma=0.0, balance_dev=None, autobalance=False, ssi=0):
    """p: list of fp32 contiguous [bs,na,ny,nx,5+nc] raw Detect outputs.  Returns (out [4] = {gain*(lbox+lobj+lcls)*bs,
    gain*lbox, gain*lobj, gain*lcls} on the device, [d out[0] / d p_i]).
    fl_gamma > 0: focal loss around both BCE criteria (loss.py:106-110).  balance_dev: fp32 [nl] DEVICE tensor used instead of
    the host list `balance`; autobalance updates it in place (loss.py:158-164, ssi = index of the stride-16 level)."""
    L = _lib.lib()
    nl = len(p)
    for t in p:
        _require_gpu(t)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("det_loss expects contiguous fp32 [bs,na,ny,nx,no] tensors (Detect's raw outputs)")
    bs, na, _, _, no = p[0].shape
    nt = int(targets.shape[0])
    tg = targets if (targets.dtype == torch.float32 and targets.is_contiguous()) else targets.float().contiguous()
    dp = [torch.empty_like(t) for t in p]
    ny = (C.c_int32 * nl)(*[t.shape[2] for t in p])
    nx = (C.c_int32 * nl)(*[t.shape[3] for t in p])
    pp = (C.c_void_p * nl)(*[t.data_ptr() for t in p])
    dpp = (C.c_void_p * nl)(*[t.data_ptr() for t in dp])
    anc = (C.c_float * (nl * na * 2))(*[float(v) for v in anchors_host])
    bal = (C.c_float * nl)(*[float(v) for v in balance])
    max_cells = max(t.numel() // no for t in p)
    nbytes = L.dsn_det_loss_workspace_bytes(nl, na, nt, nc, max_cells)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p[0].device)
    out = torch.empty(4, dtype=torch.float32, device=p[0].device)
    if fl_gamma > 0 or balance_dev is not None:
        if balance_dev is not None and (balance_dev.dtype != torch.float32 or balance_dev.numel() < nl or not balance_dev.is_cuda):
            raise ValueError("det_loss: balance_dev must be a CUDA fp32 tensor of nl elements")
        _lib.check(L.dsn_det_loss_opt(pp, dpp, ny, nx, nl, bs, na, nc, tg.data_ptr() if nt else None, nt, anc, bal,
                                      h_box * gain, h_obj * gain, h_cls * gain, cls_pw, obj_pw, anchor_t, cp, cn, float(fl_gamma),
                                      _p(balance_dev), int(bool(autobalance)), int(ssi), out.data_ptr(), ws.data_ptr(), nbytes,
                                      stream_ptr()), "det_loss_opt")
        return out, dp
    _lib.check(L.dsn_det_loss(pp, dpp, ny, nx, nl, bs, na, nc, tg.data_ptr() if nt else None, nt, anc, bal,
                              h_box * gain, h_obj * gain, h_cls * gain, cls_pw, obj_pw, anchor_t, cp, cn, out.data_ptr(),
                              ws.data_ptr(), nbytes, stream_ptr()), "det_loss")
    return out, dp


def seg_ce(logits, target, ignore_index=-1, want_grad=True):
    """nn.CrossEntropyLoss(ignore_index) on contiguous NCHW fp32 logits.  Returns (out [2] = {loss, 1/valid}, dlogits|None)."""
    _require_gpu(logits)
    lg = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
    tg = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.long().contiguous()
    n, c, h, w = lg.shape
    L = _lib.lib()
    nbytes = L.dsn_seg_ce_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=lg.device)
    out = torch.empty(2, dtype=torch.float32, device=lg.device)
    dl = torch.empty_like(lg) if want_grad else None
    _lib.check(L.dsn_seg_ce(lg.data_ptr(), tg.data_ptr(), n, c, h, w, int(ignore_index), out.data_ptr(), _p(dl),
                            ws.data_ptr(), nbytes, stream_ptr()), "seg_ce")
    return out, dl


_seg_up_ws = {}


def seg_ce_up(logits, target, out_hw, ignore_index=-1, gain=1.0):
    """Fused x-scale bilinear(align_corners=True) + CrossEntropyLoss on the LOW-resolution seg logits (an NHWC activation
    [N, C, h, w]): returns (out [2] = {mean CE, 1/valid}, dlogits: zero-padded NHWC activation of the logits' shape holding
    gain * d loss / d logits).  Raises KernelUnsupported for shapes the fused kernel does not take."""
    L = _lib.lib()
    tg = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.long().contiguous()
    n, c, h, w = logits.shape
    H, W = int(out_hw[0]), int(out_hw[1])
    a = desc(logits)
    vec = 4 if logits.dtype == torch.float32 else 8
    dl = new_act(n, c, h, w, logits.dtype, logits.device, ldc_align=vec)
    b = desc(dl)
    nbytes = L.dsn_seg_ce_up_workspace_bytes(n, h, w, c, H)
    key = (logits.device, nbytes)
    ws = _seg_up_ws.get(key)
    if ws is None:
        ws = _seg_up_ws[key] = torch.zeros(nbytes, dtype=torch.uint8, device=logits.device)     # zero ONCE: the call restores it
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    rc = L.dsn_seg_ce_up(C.byref(a), tg.data_ptr(), H, W, int(ignore_index), float(gain), out.data_ptr(), C.byref(b),
                         ws.data_ptr(), nbytes, stream_ptr())
    if rc == DSN_EUNSUPPORTED:
        raise KernelUnsupported(_lib.lib().dsn_last_error().decode(errors="replace"))
    _lib.check(rc, "seg_ce_up")
    dl._dsn_zero_padded = True
    return out, dl


# ------------------------------------------------------------------------------------------------ evaluation arithmetic
def box_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """IoU matrix [N, M] of fp32 xyxy boxes on the device (reference operation order)."""
    _require_gpu(box1)
    a = box1 if (box1.dtype == torch.float32 and box1.is_contiguous()) else box1.float().contiguous()
    b = box2 if (box2.dtype == torch.float32 and box2.is_contiguous()) else box2.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().dsn_box_iou(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), stream_ptr()),
               "box_iou")
    return out


def seg_eval_counts(logits: torch.Tensor, target: torch.Tensor, nclass: int):
    """(correct, labelled, inter[nclass-1], pred[nclass-1], lab[nclass-1]) -- ints / int64 numpy arrays (one host sync)."""
    _require_gpu(logits)
    lg = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
    tg = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.long().contiguous()
    n, c, h, w = lg.shape
    nb = nclass - 1
    out = torch.empty(2 + 3 * nb, dtype=torch.int64, device=lg.device)
    _lib.check(_lib.lib().dsn_seg_eval_counts(lg.data_ptr(), tg.to(lg.device).data_ptr(), n, c, h, w, nclass, out.data_ptr(),
                                              stream_ptr()), "seg_eval_counts")
    v = out.cpu().numpy()
    return int(v[0]), int(v[1]), v[2:2 + nb].copy(), v[2 + nb:2 + 2 * nb].copy(), v[2 + 2 * nb:2 + 3 * nb].copy()


def resize_bilinear_nchw(x: torch.Tensor, size, align_corners: bool = False) -> torch.Tensor:
    """F.interpolate(x, size, mode='bilinear', align_corners=...) for contiguous NCHW fp32 (evaluation resize of the seg logits)."""
    _require_gpu(x)
    xx = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
    n, c, h, w = xx.shape
    ho, wo = int(size[0]), int(size[1])
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dsn_resize_bilinear_nchw(xx.data_ptr(), y.data_ptr(), n * c, h, w, ho, wo, int(align_corners),
                                                   stream_ptr()), "resize_bilinear_nchw")
    return y


def seg_argmax_nearest(logits: torch.Tensor, size=None) -> torch.Tensor:
    """argmax over classes (first maximum) as float, nearest-resized to `size`: [N, size[0], size[1]] fp32."""
    _require_gpu(logits)
    lg = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
    n, c, h, w = lg.shape
    ho, wo = (h, w) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((n, ho, wo), dtype=torch.float32, device=lg.device)
    _lib.check(_lib.lib().dsn_seg_argmax_nearest(lg.data_ptr(), out.data_ptr(), n, c, h, w, ho, wo, stream_ptr()),
               "seg_argmax_nearest")
    return out


def letterbox_u8(src_hwc: torch.Tensor, out_hw, new_hw, top: int, left: int, color=(114, 114, 114), chw_reversed=False):
    """Resize a uint8 HWC image to new_hw, place it at (top, left) of an out_hw canvas filled with `color`; optionally emit CHW
    with reversed channel order (BGR -> RGB).  One launch."""
    _require_gpu(src_hwc)
    if src_hwc.dtype != torch.uint8 or src_hwc.dim() != 3 or src_hwc.shape[2] != 3:
        raise TypeError("letterbox_u8: expects a uint8 [H, W, 3] image")
    src = src_hwc if src_hwc.is_contiguous() else src_hwc.contiguous()
    h, w = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((3, h, w) if chw_reversed else (h, w, 3), dtype=torch.uint8, device=src.device)
    _lib.check(_lib.lib().dsn_letterbox_u8(src.data_ptr(), src.shape[0], src.shape[1], out.data_ptr(), h, w, int(new_hw[0]),
                                           int(new_hw[1]), int(top), int(left), int(color[0]), int(color[1]), int(color[2]),
                                           int(chw_reversed), stream_ptr()), "letterbox_u8")
    return out
