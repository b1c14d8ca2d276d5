#!/usr/bin/env python3
"""bench.py -- DeSeNet-s hot path on MI355X: images/sec (640x640).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|infer] [--dtype bf16|fp32] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

mode train (default; BASELINE.json config 3 / 4): one step = zero grads -> forward (HIP) -> det + seg loss -> backward (HIP) ->
  [RCCL all-reduce of the flat gradient buffer] -> SGD-nesterov step, batch 8 per GPU, bf16 storage / fp32 accumulate,
  synthetic 640x640 images + targets (seed 3 + rank), hash-filled weights.  Weak scaling: per-GPU work is fixed.
mode infer (config 2): fused model, batch 16, fp32, forward + Detect decode + NMS (conf .25 / IoU .45 / max_det 1000).

Prints ONE JSON line (rank 0).  `roofline` is the kernel with the largest share of device time in the timed region,
measured live with HIP events recorded by the library on the launch stream (dsn_profile_*); `cpu_baseline` is the CPU
oracle (oracle/, a PyTorch-CPU restatement of the reference graph) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"hbm_GBs": 8000.0, "mfma_bf16_TFs": 2500.0, "mfma_f32_TFs": 157.3}   # MI355X_MICROARCH.md chip-level table
from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN  # noqa: E402  (scripts/train.py:285)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the cores this process may actually use (cgroup/affinity), capped at the 16-core
    share a one-GPU box grants."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, os.cpu_count() or n, 16))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default=None)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 8 train / 16 infer)")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--model", choices=["s", "m"], default="s", help="s = DeSeNet-s (configs 1-4, the headline), m = config 5's graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dump-layers", default="", help="write the per-(kernel label, layer) table of the profiled eager steps to this file")
    ap.add_argument("--dump-json", default="", help="write the full result (per-kernel and per-layer tables, the `also` children's tables) "
                    "to this file; stdout carries only the compact line.  Default: gpurun_out/bench_detail_<config>.json")
    ap.add_argument("--no-ema", action="store_true", help="train: leave out the rank-0 ModelEMA update (train.py:374)")
    ap.add_argument("--eager", action="store_true", help="train: launch every kernel from Python instead of hipGraph replay")
    ap.add_argument("--accumulate", type=int, default=1, help="train: micro-batches per optimizer step (train.py:146; 1 = step every batch)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the extra `also` sections (config 2 inference, config 5 DeSeNet-m 1280) after the headline")
    ap.add_argument("--also-only", action="store_true", help=argparse.SUPPRESS)      # child run of an `also` section
    return ap.parse_args()


def build_model(device, which="s"):
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.synth import synthetic_checkpoint
    m = Model(f"desenet_{which}.yaml", ch=3, nc=6)
    sd = m.state_dict()
    synthetic_checkpoint(sd)
    m.load_state_dict(sd)
    return m.to(device)


def cpu_baseline_train(img, batch=2, budget_s=12.0):
    """The oracle's training step (fwd + losses + bwd, fp32, NCHW, stock ATen CPU kernels) on the host cores."""
    import yaml
    from oracle import desenet_ref as R
    from oracle import loss_ref
    from desenet_amd.synth import synth_images, synth_targets, synthetic_checkpoint
    threads = host_threads()
    torch.set_num_threads(threads)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "desenet_amd", "cfg", "desenet_s.yaml")))
    sd0 = R.make_state_dict(cfg)
    synthetic_checkpoint(sd0)
    is_p = lambda k: "running" not in k and "num_batches" not in k and "anchor" not in k
    x = synth_images(batch, img, 3)
    dt, st = synth_targets(batch, img, 3)

    def step():
        sd = {k: v.clone().requires_grad_(is_p(k)) for k, v in sd0.items()}
        raws, seg, _ = R.forward(cfg, sd, x, training=True)
        total, *_ = loss_ref.step_loss(raws, seg, dt, st, sd["model.25.anchors"], 6, img)
        total.backward()

    step()
    t0 = time.perf_counter()
    timed = 0
    while timed < 2 or time.perf_counter() - t0 < budget_s:
        step()
        timed += 1
    el = time.perf_counter() - t0
    return {"value": batch * timed / el, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle train step (fwd+loss+bwd, fp32), batch {batch} x {timed} timed steps after 1 warm-up, "
                      f"{img}x{img}, torch {torch.__version__} CPU, {threads} threads"}


def cpu_baseline_infer(img, batch=2, budget_s=12.0):
    import yaml
    from oracle import desenet_ref as R
    from oracle import nms_ref
    from desenet_amd.synth import synth_images, synthetic_checkpoint
    threads = host_threads()
    torch.set_num_threads(threads)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "desenet_amd", "cfg", "desenet_s.yaml")))
    sd = R.make_state_dict(cfg)
    synthetic_checkpoint(sd)
    sd = R.fold_bn(sd)
    x = synth_images(batch, img, 2)

    def step():
        with torch.no_grad():
            (pred, _), seg, _ = R.forward(cfg, sd, x, fused=True)
        nms_ref.non_max_suppression(pred.numpy(), 0.25, 0.45, max_det=1000)

    step()
    t0 = time.perf_counter()
    timed = 0
    while timed < 2 or time.perf_counter() - t0 < budget_s:
        step()
        timed += 1
    el = time.perf_counter() - t0
    return {"value": batch * timed / el, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle fused eval forward + NMS, batch {batch} x {timed} timed after 1 warm-up, {img}x{img}, "
                      f"torch {torch.__version__} CPU, {threads} threads"}


def _traffic_table():
    """profiles/traffic.json: HBM-side bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950
    corrections of MI355X_MICROARCH.md, HBM), collected by tools/pmc_traffic.sh on config 3's shapes and keyed by kernel label."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tfile))
    except Exception:
        return {}


def roofline_from_profile(prof, steps, dtype):
    """prof: {label: launches/ms/flops/bytes} (hip_ops.profile_collect); label = rocprofv3 symbol family/dtype/tile/direction."""
    if not prof:
        return None, {}
    table = {}
    for name, r in prof.items():
        table[name] = {"launches_per_step": r["launches"] / steps, "ms_per_step": r["ms"] / steps,
                       "TFLOPs": (r["flops"] / (r["ms"] * 1e-3) / 1e12) if r["flops"] else None,
                       "GBs": r["bytes"] / (r["ms"] * 1e-3) / 1e9}
    name = max(prof, key=lambda k: prof[k]["ms"])
    r = prof[name]
    avg_ms = r["ms"] / r["launches"]
    peak_tf = PEAK["mfma_bf16_TFs"] if "bf16" in name else PEAK["mfma_f32_TFs"]
    ridge = peak_tf * 1e12 / (PEAK["hbm_GBs"] * 1e9)          # FLOP per byte where the two roofs meet (312 bf16 / 20 fp32)
    intensity = r["flops"] / r["bytes"] if r["bytes"] else 0.0
    # a kernel whose algorithmic intensity is below the ridge is priced against HBM, above it against MFMA
    if r["flops"] > 0 and intensity >= ridge:
        ach = r["flops"] / r["launches"] / (avg_ms * 1e-3) / 1e12
        roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf}
    else:
        ach = r["bytes"] / r["launches"] / (avg_ms * 1e-3) / 1e9
        roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK["hbm_GBs"], "unit": "GB/s",
                "frac": ach / PEAK["hbm_GBs"]}
    roof.update({"avg_launch_us": avg_ms * 1e3, "launches_per_step": r["launches"] / steps,
                 "algorithmic_GBs": r["bytes"] / r["launches"] / (avg_ms * 1e-3) / 1e9,
                 "algorithmic_TFLOPs": (r["flops"] / r["launches"] / (avg_ms * 1e-3) / 1e12) if r["flops"] else None,
                 "flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                 # PMC bytes per launch (profiles/traffic.json, keyed by symbol family / dtype / tile: fwd and dgrad share a symbol)
                 "traffic": _traffic_table().get(name, _traffic_table().get(name.rsplit("/", 1)[0])),
                 "timing": "HIP events around the same launches on eager steps right after the timed hipGraph replays, minus one "
                           f"queue marker (half of what an EMPTY event pair measures on that stream: {_pair_overhead_us():.2f} us "
                           "subtracted per launch); profiles/*_kernel_stats.csv hold the rocprofv3 view of the replayed step"})
    return roof, table


def _pair_overhead_us():
    from desenet_amd import hip_ops
    return hip_ops.last_event_pair_overhead_us


def roofline_by_layer(layers, steps, train):
    """The two quantities BASELINE.json's north_star names, per layer, from the same HIP-event records:
      train: the C3 / Bottleneck 3x3 convolutions (k3, stride 1, Ci == Co: common.py:107) forward and dgrad against the dense bf16
             MFMA peak;
      infer: every fused Conv+BN+SiLU launch (BN folded, activation in the epilogue: common.py:55-56) against the HBM roof."""
    if not layers:
        return None
    rows, tot = [], {}
    for (label, layer), r in sorted(layers.items()):
        if not layer or r["ms"] <= 0:
            continue
        sym, dt, tile, direction = (label.split("/") + ["", "", "", ""])[:4]
        peak_tf = PEAK["mfma_bf16_TFs"] if dt == "bf16" else PEAK["mfma_f32_TFs"]
        try:
            geom, chans = layer.split(" ")[0], layer.split(" ")[1]
            ci, co = (int(v) for v in chans.split("->"))
        except (IndexError, ValueError):
            continue                     # (a labelled launch that is not a convolution layer)
        sec = r["ms"] * 1e-3
        row = {"kernel": label, "layer": layer, "launches_per_step": r["launches"] / steps,
               "avg_us": r["ms"] / r["launches"] * 1e3, "TFLOPs": r["flops"] / sec / 1e12, "GBs": r["bytes"] / sec / 1e9,
               "mfma_frac": r["flops"] / sec / 1e12 / peak_tf, "hbm_frac": r["bytes"] / sec / 1e9 / PEAK["hbm_GBs"]}
        if train:
            if not (geom.startswith("k3s1") and ci == co):
                continue
            key = "c3_3x3_" + direction
        else:
            key = "conv_bn_silu_fused_k" + geom[1]
        rows.append(row)
        t = tot.setdefault(key, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "peak_tf": peak_tf})
        t["ms"] += r["ms"]; t["flops"] += r["flops"]; t["bytes"] += r["bytes"]
    summary = {k: {"TFLOPs": t["flops"] / (t["ms"] * 1e-3) / 1e12, "mfma_frac": t["flops"] / (t["ms"] * 1e-3) / 1e12 / t["peak_tf"],
                   "GBs": t["bytes"] / (t["ms"] * 1e-3) / 1e9, "hbm_frac": t["bytes"] / (t["ms"] * 1e-3) / 1e9 / PEAK["hbm_GBs"],
                   "ms_per_step": t["ms"] / steps} for k, t in tot.items() if t["ms"] > 0}
    return {"what": ("C3/Bottleneck 3x3 convolutions vs the dense bf16 MFMA peak (2.5 PFLOP/s)" if train else
                     "fused Conv+BN+SiLU (eval) launches vs the HBM roof (8 TB/s); algorithmic bytes = input + output + weights once"),
            "summary": summary, "layers": rows}


LINE_LIMIT = 4000      # the driver parses the LAST ~8 KB of stdout; round 3's 46 KB line came back as "parsed": null


def _round(v, nd=4):
    if isinstance(v, float):
        return float(f"{v:.{nd}g}") if abs(v) < 1 else round(v, 3)
    return v


def compact_roofline(r):
    """The keys the task statement names for `roofline` (+ kernel label, launch time): no prose, no duplicate rates."""
    if not r:
        return None
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step", "flop_per_byte")
    return {k: _round(r.get(k)) for k in keep if k in r}


def compact_layer_classes(by_layer):
    """roofline_by_layer.summary reduced to {class: {frac-of-its-roof, rate, ms}} -- the two north_star targets."""
    if not by_layer or not by_layer.get("summary"):
        return None
    out = {}
    for k, t in by_layer["summary"].items():
        if k.startswith("c3_3x3"):
            out[k] = {"bound": "mfma", "frac": _round(t["mfma_frac"]), "TFLOPs": _round(t["TFLOPs"]), "ms_per_step": _round(t["ms_per_step"])}
        else:
            out[k] = {"bound": "hbm", "frac": _round(t["hbm_frac"]), "GBs": _round(t["GBs"]), "mfma_frac": _round(t["mfma_frac"]),
                      "ms_per_step": _round(t["ms_per_step"])}
    return out


def compact_line(out):
    """ONE stdout JSON line below LINE_LIMIT bytes: the contract keys + `roofline` + `cpu_baseline` + per-class fractions + a
    compact `also`.  Everything tabular (`kernels`, `roofline_by_layer.layers`, the children's tables) lives in the detail file."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
    line = {k: _round(out[k]) if not isinstance(out.get(k), dict) else out[k] for k in keys if k in out}
    line["roofline"] = compact_roofline(out.get("roofline"))
    cpu = out.get("cpu_baseline")
    line["cpu_baseline"] = ({k: _round(v) for k, v in cpu.items()} if cpu else None)
    line["layer_classes"] = compact_layer_classes(out.get("roofline_by_layer"))
    if out.get("also"):
        also = {}
        for name, sec in out["also"].items():
            if "error" in sec:
                also[name] = {"error": str(sec["error"])[-160:]}
                continue
            r = sec.get("roofline") or {}
            also[name] = {"value": _round(sec.get("value")), "unit": sec.get("unit"), "ms_per_step": _round(sec.get("ms_per_step")),
                          "dtype": sec.get("dtype"), "steps": sec.get("steps"),
                          "roofline": {k: _round(r.get(k)) for k in ("kernel", "bound", "frac") if k in r},
                          "layer_classes": {k: v.get("frac") for k, v in (sec.get("layer_classes") or {}).items()}}
        line["also"] = also
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_LIMIT:             # never let a long label cost the round its number again: shed optional parts
        for victim in ("also", "layer_classes"):
            if victim in line and len(txt) > LINE_LIMIT:
                line[victim] = None
                txt = json.dumps(line, separators=(",", ":"))
        if len(txt) > LINE_LIMIT:
            line["config"]["workload"] = line["config"]["workload"][:200]
            if line.get("cpu_baseline"):
                line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
            txt = json.dumps(line, separators=(",", ":"))
    return txt


def default_detail_path(a):
    """Where the tables go when --dump-json is not given: gpurun_out/ (merged back by gpurun), else nowhere."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return ""
    tag = f"{a.mode}_{a.model}_{a.img}_b{a.batch or 0}_{a.dtype or 'def'}_g{a.gpus}"
    return os.path.join(d, f"bench_detail_{tag}.json")


def also_sections(detail_dir=""):
    """The other single-GPU configurations of BASELINE.json, measured by child runs of this script right after the headline
    (fresh processes: no shared caches with the timed region above): config 2 (fused fp32 inference + NMS, batch 16) and
    config 5's per-GPU shape (DeSeNet-m, 1280x1280, batch 4, bf16 training step)."""
    import subprocess
    runs = {"config2_infer_fp32_b16": ["--mode", "infer", "--steps", "30", "--warmup", "8"],
            "config2_infer_bf16_b16": ["--mode", "infer", "--dtype", "bf16", "--steps", "30", "--warmup", "8"],
            "config5_m1280_bf16_b4": ["--model", "m", "--img", "1280", "--batch", "4", "--steps", "12", "--warmup", "4"]}
    out = {}
    for name, extra in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--also-only", *extra]
        if detail_dir:
            cmd += ["--dump-json", os.path.join(detail_dir, f"bench_detail_{name}.json")]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": (r.stderr or r.stdout)[-400:]}
                continue
            j = json.loads(line[-1])
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                         "steps": j["steps"], "warmup": j["warmup"], "dtype": j["dtype"], "workload": j["config"]["workload"],
                         "roofline": j["roofline"], "layer_classes": j.get("layer_classes"),
                         "wall_s": time.perf_counter() - t0}
        except Exception as e:      # never lose the headline line to a failing side section
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        log(f"also[{name}]: {out[name].get('value', out[name].get('error'))}")
    return out


def self_launch(a):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, as fresh child processes of
    `python -m torch.distributed.run` (one rank per GPU), BEFORE this process has made any GPU call, relay rank 0's JSON line
    and exit with the launcher's status (scripts/train.py:555-561 is started the same way)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    # the task statement's environment note: the host driver supports dmabuf IPC only; without this RCCL's cross-process buffer
    # registration fails with `hipIpcGetMemHandle: invalid argument`.  Already exported on the boxes; kept for a bare shell.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("no launcher around --gpus", a.gpus, "-> starting", " ".join(cmd[1:8]), "...")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(r.returncode if r.returncode or lines else 1)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or plain `python bench.py --gpus N`)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path (no CPU fallback)")
    # (rehearsal on a one-GPU box: DSN_BENCH_BACKEND=gloo lets N ranks share the card; the driver's runs use RCCL, one GPU each)
    backend = os.environ.get("DSN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.synth import synth_images, synth_targets

    train = a.mode == "train"
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype or ("bf16" if train else "fp32")]
    batch = a.batch or (8 if train else 16)
    desenet_amd.set_compute_dtype(dtype)
    model = build_model(dev, a.model)

    if train:
        from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
        from desenet_amd.parallel import FlatGradients, broadcast_parameters, sgd_param_groups
        from desenet_amd.core.utils.hyp import scale_hyp
        model.train()
        model.hyp = scale_hyp(6, a.img)
        broadcast_parameters(model)
        flat = FlatGradients(model.parameters())
        from desenet_amd.optim import FusedSGD
        opt = FusedSGD(sgd_param_groups(model), lr=0.01, momentum=0.937, nesterov=True)     # torch.optim.SGD math, one launch
        compute_loss, compute_seg_loss = ComputeLoss(model), SegmentationLosses()
        # the loader's product: a uint8 NCHW batch on the device; its `.float() / 255` (train.py:329) is folded into Focus
        x = (synth_images(batch, a.img, 3 + rank) * 255.0).round().to(torch.uint8).to(dev)
        det_t, seg_t = synth_targets(batch, a.img, 3 + rank)
        ema = None
        if rank == 0 and not a.no_ema:           # train.py:179: ModelEMA on rank 0 only, updated after every optimizer step
            from desenet_amd.core.utils.torch_utils import ModelEMA
            ema = ModelEMA(model)
        det_t, seg_t = det_t.to(dev), seg_t.to(dev)

        def loss_fn(det_pred, seg_pred):
            det_loss, _ = compute_loss(det_pred, det_t)
            return det_loss * DETGAIN + compute_seg_loss(seg_pred, seg_t) * SEGGAIN

        def loss_and_grads(det_pred, seg_pred, det_labels, seg_labels):
            # same losses, gradients straight from the kernels (graph-capturable); labels come from the step's static buffers
            out, d_det = compute_loss.forward_backward(det_pred, det_labels, gain=DETGAIN)
            sout, d_seg = compute_seg_loss.forward_backward(seg_pred, seg_labels, gain=SEGGAIN)
            return (out, sout), d_det, d_seg

        micro = [0]

        def eager_step():
            if micro[0] == 0:
                flat.zero()
            det_pred, seg_pred = model(x)
            loss_fn(det_pred, seg_pred).backward()
            micro[0] = (micro[0] + 1) % a.accumulate
            if micro[0] == 0:
                flat.all_reduce()
                opt.step()
                if ema is not None:
                    ema.update(model)

        step = eager_step
        mode_note = "eager launches"
        if not a.eager:
            try:
                from desenet_amd.graph import GraphedTrainStep
                graphed = GraphedTrainStep(model, loss_and_grads, flat, opt, x, ema=ema, det_targets=det_t, seg_targets=seg_t,
                                           max_targets=max(256, int(det_t.shape[0])), accumulate=a.accumulate)
                # every replay is handed the batch (images + labels are copied into the step's static buffers, as a loader
                # would; here the same resident synthetic batch each time)
                step = lambda: graphed(x, det_t, seg_t)
                mode_note = ("one hipGraph replay per step (u8 input /255 + pack + fwd + losses + bwd + SGD + EMA)" if world == 1 and a.accumulate == 1 else
                             f"hipGraph replays: fwd + losses + bwd halves around the asynchronous RCCL all-reduce of the flat-buffer tail "
                             f"(split at layer {graphed.split}), then SGD (+ EMA on rank 0); accumulate {a.accumulate}")
            except Exception as e:   # keep the bench alive, but say so loudly
                log(f"hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches")
    else:
        from desenet_amd.core.utils.general import non_max_suppression
        model.eval().fuse()
        # the loader's product: uint8 NCHW on the device (detect.py:127-129 `img.float() / 255` is folded into Focus)
        x = (synth_images(batch, a.img, 2 + rank) * 255.0).round().to(torch.uint8).to(dev)
        fwd = lambda: model(x)
        if not a.eager:
            try:
                from desenet_amd.graph import GraphedInference
                ginf = GraphedInference(model, x)
                fwd = lambda: ginf()
            except Exception as e:
                log(f"hipGraph capture of the forward pass failed ({type(e).__name__}: {e}); eager launches")

        def eager_step():
            with torch.no_grad():
                (pred, _), seg = model(x)
                return non_max_suppression(pred, 0.25, 0.45, max_det=1000), seg

        def step():
            with torch.no_grad():
                (pred, _), seg = fwd()
                return non_max_suppression(pred, 0.25, 0.45, max_det=1000), seg

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model ready, mode={a.mode} dtype={dtype} batch={batch}; warm-up x{a.warmup}")
    for i in range(a.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    sync()
    log(f"timing {a.steps} steps")
    graphed_run = not a.eager and step is not eager_step
    if not a.no_profile and not graphed_run:
        ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    prof, prof_layers = {}, {}
    prof_steps = a.steps
    if not a.no_profile:
        if graphed_run:
            ops.profile_enable(True)
            # graph replay cannot carry the library's event pairs: time the SAME launches (same kernels, shapes, stream)
            # on eager steps right after the timed region
            prof_steps = min(a.steps, 5)
            ops.profile_collect()
            for _ in range(prof_steps):
                eager_step()
            torch.cuda.synchronize()
        prof, prof_layers = ops.profile_collect(by_layer=True)
        ops.profile_enable(False)
        if a.dump_layers and rank == 0:
            with open(a.dump_layers, "w") as f:
                f.write(f"{'us/step':>9s} {'n/step':>6s} {'avg us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}  kernel label | layer\n")
                for (label, layer), r in sorted(prof_layers.items(), key=lambda kv: -kv[1]["ms"]):
                    sec = max(r["ms"] * 1e-3, 1e-12)
                    f.write(f"{r['ms'] * 1e3 / prof_steps:9.1f} {r['launches'] / prof_steps:6.1f} {r['ms'] * 1e3 / max(r['launches'], 1):8.1f} "
                            f"{r['flops'] / sec / 1e12:8.1f} {r['bytes'] / sec / 1e9:7.0f}  {label} | {layer}\n")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log(f"timed region: {elapsed:.3f} s")
    comm_info = {}
    if world > 1:
        # evidence that the collective really spans `world` ranks (outside the timed region): every rank contributes 1, and the
        # flat gradient buffer's all-reduce -- the one exchange of the step -- is timed on its own
        one = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(one)
        comm_info = {"backend": dist.get_backend(), "rccl_ranks": int(round(float(one.item())))}
        if train:
            buf = torch.zeros_like(flat.flat)
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            comm_info["allreduce_us"] = round((time.perf_counter() - t1) / 10 * 1e6, 1)
            comm_info["allreduce_MB"] = round(buf.numel() * buf.element_size() / 1e6, 1)
        log(f"rank {rank}: {comm_info}")
    if rank == 0:
        roof, table = roofline_from_profile(prof, prof_steps, dtype)
        if roof is not None and not (train and a.model == "s" and batch == 8 and a.img == 640):
            roof["traffic"] = None      # profiles/traffic.json holds the PMC passes of config 3's shapes only
        by_layer = roofline_by_layer(prof_layers, prof_steps, train)
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            log(f"CPU baseline on {host_threads()} threads ...")
            cpu = cpu_baseline_train(a.img) if train else cpu_baseline_infer(a.img)
        if train:
            metric = f"images/sec ({a.img}x{a.img}) train fwd+bwd"
            workload = (f"config {('3' if world == 1 else '4') if a.model == 's' else '5'}: DeSeNet-{a.model} training step (fwd + det/seg loss + bwd + "
                        f"{'RCCL flat all-reduce + ' if world > 1 else ''}SGD), batch {batch}/GPU, {a.img}x{a.img}, "
                        f"{mode_note}")
        else:
            metric = f"images/sec ({a.img}x{a.img}) inference fwd+NMS"
            workload = (f"config 2: DeSeNet-s fused inference (uint8 input /255 + fwd + Detect decode + NMS), batch {batch}, "
                        f"{a.img}x{a.img}, {'eager launches' if a.eager else 'forward pass replayed from a hipGraph, NMS eager'}")
        out = {
            "metric": metric, "value": world * batch * a.steps / elapsed, "unit": "images/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if dtype == torch.bfloat16 else "f32",
            "data": "synthetic (seeded uniform images, seeded boxes/masks, hash-filled weights)",
            "config": {"workload": workload, "batch_per_gpu": batch, "img": a.img, "parallelism": f"dp{world}", **comm_info},
            "roofline": roof, "roofline_by_layer": by_layer, "cpu_baseline": cpu, "kernels": table,
        }
        detail_path = a.dump_json or default_detail_path(a)
        if world == 1 and train and a.model == "s" and not a.no_also and not a.also_only:
            out["also"] = also_sections(os.path.dirname(detail_path) if detail_path else "")
        if detail_path:                      # the per-kernel / per-layer tables: a side file, never the stdout line
            try:
                os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
                with open(detail_path, "w") as f:
                    json.dump(out, f, indent=1)
                log("detail tables ->", detail_path)
            except OSError as e:
                log(f"could not write {detail_path}: {e}")
        print(compact_line(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
