#!/usr/bin/env python3
"""Per-op, per-shape device time of ONE DeSeNet-s training step (batch 8, 640x640, bf16, eager launches).

Every launching function of desenet_amd.hip_ops is wrapped with a pair of stream events; the table groups calls by
(op, tensor shapes) and sorts by total time, so the layers that dominate a step are visible by shape -- the library's own
profiler (dsn_profile_*) aggregates per kernel id only.  Usage: python tools/layer_profile.py [top_n] [batch]"""
import sys
from collections import defaultdict

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import desenet_amd  # noqa: E402
from desenet_amd import hip_ops as ops  # noqa: E402

SKIP = {"new_act", "as_act", "desc", "scratch", "conv_params", "conv_out_hw", "stream_ptr", "profile_enable",
        "profile_collect", "pack_weight_fwd", "pack_weight_dgrad"}


def shape_key(args):
    out = []
    for a in args:
        if isinstance(a, torch.Tensor):
            out.append("x".join(map(str, a.shape)))
        elif isinstance(a, ops.dsn_conv_params):
            out.append(f"k{a.kh}s{a.stride}d{a.dil}")
        elif isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
            out.append("[" + ",".join("x".join(map(str, t.shape)) for t in a) + "]")
    return " ".join(out[:4])


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    desenet_amd.set_compute_dtype(torch.bfloat16)
    model = bench.build_model(dev).train()
    from desenet_amd.core.utils.hyp import scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.parallel import FlatGradients
    from desenet_amd.synth import synth_images, synth_targets
    model.hyp = scale_hyp(6, 640)
    flat = FlatGradients(model.parameters())
    cl, sl = ComputeLoss(model), SegmentationLosses()
    x = synth_images(batch, 640, 3).to(dev)
    det_t, seg_t = synth_targets(batch, 640, 3)
    det_t, seg_t = det_t.to(dev), seg_t.to(dev)

    def step():
        flat.zero()
        det_pred, seg_pred = model(x)
        (cl(det_pred, det_t)[0] * bench.DETGAIN + sl(seg_pred, seg_t) * bench.SEGGAIN).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()

    records = []
    import types
    for name in dir(ops):
        fn = getattr(ops, name)
        if name.startswith("_") or name in SKIP or not isinstance(fn, types.FunctionType) or fn.__module__ != ops.__name__:
            continue

        def make(fn=fn, name=name):
            def wrapped(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a, **k)
                e1.record()
                records.append((name, shape_key(a), e0, e1))
                return r
            return wrapped
        setattr(ops, name, make())

    reps = 3
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    per_op = defaultdict(lambda: [0, 0.0])
    for name, key, e0, e1 in records:
        t = e0.elapsed_time(e1) * 1e3
        agg[(name, key)][0] += 1
        agg[(name, key)][1] += t
        per_op[name][0] += 1
        per_op[name][1] += t
    total = sum(v[1] for v in per_op.values()) / reps
    print(f"sum of op times: {total / 1e3:.2f} ms/step (eager, event pairs; includes launch gaps inside multi-kernel ops)")
    print("-- by op")
    for name, (n, t) in sorted(per_op.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:24s} {n / reps:6.0f} calls {t / reps / 1e3:8.3f} ms  avg {t / n:7.1f} us")
    print("-- by op and shape")
    for (name, key), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name:20s} {n / reps:4.0f}x {t / n:7.1f} us = {t / reps / 1e3:6.3f} ms  {key}")


if __name__ == "__main__":
    main()
