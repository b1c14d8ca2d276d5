"""Host-side planning logic that decides where BatchNorm work is fused (no GPU): which top-level layers complete the gradient of
their input (Model._final_consumer -> conv_impl.conv_block_bwd(fuse_up=True)), and the memory-range registry that maps a consumer's input view to the BatchNorm blocks that produced it
(runtime.Tape.bn_register / bn_producers, zero-copy concats and channel slices included)."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(name):
    from desenet_amd.core.models.yolo import Model
    return Model(os.path.join(ROOT, "desenet_amd", "cfg", name), ch=3, nc=6)


def test_final_consumer_plan_of_desenet_s():
    """yolov5s_seg graph (reference core/models/yolov5s_seg.yaml + README layer table): a layer completes its input's gradient iff it is
    the lowest-index consumer of a convolution-block layer.  4 and 6 also feed the head Concats (16, 12), 17 and 20 also feed
    Detect (25): their stride-2 successors 5, 7, 18, 21 are the LAST writers and accumulate; 10 and 14 feed an Upsample first
    (lowest index 11 / 15: not a conv block), 12 / 16 / 19 / 22 are Concats (their sources have consumers of their own)."""
    m = _model("desenet_s.yaml")
    assert sorted(m._final_consumer) == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 18, 21]
    types = {l.i: type(l).__name__ for l in m.model}
    assert all(types[i] in ("Conv", "C3", "SPP") for i in m._final_consumer)
    # config 5's graph has the same topology
    assert sorted(_model("desenet_m.yaml")._final_consumer) == sorted(m._final_consumer)


def test_bn_producer_registry_resolves_views_by_memory():
    from desenet_amd import hip_ops as ops
    from desenet_amd.runtime import Tape
    tape = Tape()
    n, h, w = 2, 5, 7
    cat = torch.zeros(n, h, w, 48).permute(0, 3, 1, 2)             # NHWC storage behind a logical-NCHW view, as ops.new_act makes
    assert ops._nhwc_ldc(cat) == 48
    rec_a, rec_b = {"name": "a"}, {"name": "b"}
    tape.bn_register(cat[:, 0:16], rec_a)                          # producer A wrote channels 0..15 of the concat buffer
    tape.bn_register(cat[:, 24:48], rec_b)                         # producer B channels 24..47 (8..23 came from something else)
    hits = tape.bn_producers(cat)
    assert [(c0, c1, r["name"], k0) for c0, c1, r, k0 in hits] == [(0, 16, "a", 0), (24, 48, "b", 0)]
    hits = tape.bn_producers(cat[:, 8:32])                         # a consumer reading a slice that straddles both
    assert [(c0, c1, r["name"], k0) for c0, c1, r, k0 in hits] == [(0, 8, "a", 8), (16, 24, "b", 0)]
    assert tape.bn_producers(cat[:, 16:24]) is None
    other = torch.zeros(n, h, w, 48).permute(0, 3, 1, 2)
    assert tape.bn_producers(other) is None                        # same shape, different memory


def test_bnred_plan_refuses_what_the_kernels_cannot_take():
    from desenet_amd import conv_impl
    from desenet_amd.runtime import Tape
    tape = Tape()
    n, h, w = 2, 4, 4
    buf = torch.zeros(n, h, w, 32, dtype=torch.bfloat16).permute(0, 3, 1, 2)
    dx = torch.zeros(n, h, w, 32, dtype=torch.bfloat16).permute(0, 3, 1, 2)
    assert conv_impl._bnred_plan(tape, buf, dx, None) == (None, ())          # nothing registered
    stats = torch.zeros(4, 32)
    rec = dict(y=torch.zeros(n, h, w, 32, dtype=torch.bfloat16).permute(0, 3, 1, 2), scale=stats[0], shift=stats[1], mean=stats[2],
               rstd=stats[3], act=1, sync=("group", 2))
    tape.bn_register(buf, rec)
    assert conv_impl._bnred_plan(tape, buf, dx, None) == (None, ())          # SyncBatchNorm: the sums are all-reduced between the passes
    rec["sync"] = None
    rec["bnred_cov"] = [(0, 32)]
    assert conv_impl._bnred_plan(tape, buf, dx, None) == (None, ())          # already summed by another launch of this pass
    rec["bnred_cov"] = []
    rec["y"] = rec["y"].float()
    assert conv_impl._bnred_plan(tape, buf, dx, None) == (None, ())          # dtype mismatch between y and dx


def test_fused_pyramid_pooling_eligibility_is_a_host_decision():
    """dsn_pp_stages_supported (pure host code): bf16, 16-byte channel vectors, at most 512 pixels per branch, and the branch's tiles
    must fit the 160 KB of LDS of one CU -- DeSeNet-s (8 x 6 x 6 pixels, 128 -> 32) does, 64 pixels of 256 -> 64 do, 144 or 288 pixels
    of 256 -> 64 do not; PyramidPooling._fusable refuses everything that is not a bf16 training step."""
    import ctypes
    from desenet_amd import _lib
    L = _lib.lib()
    bf16, f32 = _lib.DSN_BF16, _lib.DSN_F32
    assert L.dsn_pp_stages_supported(288, 128, 32, bf16) == 1
    assert L.dsn_pp_stages_supported(64, 256, 64, bf16) == 1
    assert L.dsn_pp_stages_supported(144, 256, 64, bf16) == 0          # does not fit LDS
    assert L.dsn_pp_stages_supported(288, 256, 64, bf16) == 0
    assert L.dsn_pp_stages_supported(288, 128, 32, f32) == 0           # bf16 only
    assert L.dsn_pp_stages_supported(288, 124, 32, bf16) == 0          # 16-byte channel vectors
    assert L.dsn_pp_stages_supported(600, 32, 8, bf16) == 0            # more than 512 pixels per branch
    from desenet_amd.core.models.common import PyramidPooling
    pp = PyramidPooling(128).train()
    x = torch.zeros(2, 128, 8, 8, dtype=torch.bfloat16)
    assert pp._fusable(x, [1, 2, 3, 6], tape=None) is False             # no tape: not a training step of this package
    assert pp.eval()._fusable(x, [1, 2, 3, 6], tape=object()) is False
    assert pp.train()._fusable(x.float(), [1, 2, 3, 6], tape=object()) is False
