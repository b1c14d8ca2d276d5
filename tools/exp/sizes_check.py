#!/usr/bin/env python3
"""Whole-network check at image / batch sizes other than the benchmark's: three eager training steps (bf16) with the default kernel
selection against the same steps with the round-1 kernels only (implicit GEMM everywhere, per-branch PyramidPooling): the losses must
agree.  usage: python tools/exp/sizes_check.py  (runs itself twice per size in child processes: the selection is read at load time)"""
import json, os, subprocess, sys

SIZES = [(640, 8), (608, 6), (672, 4), (512, 10), (416, 12), (320, 24), (736, 3)]
CONSERVATIVE = {"DSN_WS": "0", "DSN_WS3": "0", "DSN_WS_S2": "0", "DSN_HALO": "0", "DSN_DMA1X1": "0", "DSN_PP_FUSED": "0",
                "DSN_MAXPOOL_CASCADE": "0"}


def child(img, batch):
    import torch
    sys.path.insert(0, ".")
    import bench, desenet_amd
    from desenet_amd.core.utils.hyp import DETGAIN, SEGGAIN, scale_hyp
    from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
    from desenet_amd.optim import FusedSGD
    from desenet_amd.parallel import FlatGradients, sgd_param_groups
    from desenet_amd.synth import synth_images, synth_targets
    dev = torch.device("cuda", 0)
    desenet_amd.set_compute_dtype(torch.bfloat16)
    m = bench.build_model(dev).train()
    m.hyp = scale_hyp(6, img)
    flat = FlatGradients(m.parameters())
    opt = FusedSGD(sgd_param_groups(m), lr=0.01, momentum=0.937, nesterov=True)
    cl, sl = ComputeLoss(m), SegmentationLosses()
    x = synth_images(batch, img, 3).to(dev)
    det_t, seg_t = synth_targets(batch, img, 3)
    det_t, seg_t = det_t.to(dev), seg_t.to(dev)
    losses = []
    g1 = None
    for i in range(3):
        flat.zero()
        det, seg = m(x)
        loss = cl(det, det_t)[0] * DETGAIN + sl(seg, seg_t) * SEGGAIN
        loss.backward()
        if i == 0:       # the first step's gradients: same weights in both runs
            g1 = {k: float(p.grad.float().norm()) for k, p in m.named_parameters() if p.grad is not None}
        opt.step()
        losses.append(float(loss))
    print(json.dumps({"losses": losses, "grad_norm": sum(v * v for v in g1.values()) ** 0.5, "g1": g1}))


if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(int(sys.argv[1]), int(sys.argv[2]))
        sys.exit(0)
    bad = 0
    for img, batch in SIZES:
        res = []
        for env_extra in ({}, CONSERVATIVE):
            env = dict(os.environ, **env_extra)
            r = subprocess.run([sys.executable, __file__, str(img), str(batch)], capture_output=True, text=True, env=env, timeout=300)
            if r.returncode != 0:
                print(f"img {img} batch {batch}: FAILED\n{r.stderr[-1500:]}")
                bad += 1
                res = None
                break
            res.append(json.loads(r.stdout.strip().splitlines()[-1]))
        if res:
            a, b = res
            rel = max(abs(p - q) / max(abs(q), 1e-6) for p, q in zip(a["losses"], b["losses"]))
            gr = abs(a["grad_norm"] - b["grad_norm"]) / max(b["grad_norm"], 1e-6)
            worst = sorted(((abs(a["g1"][k] - v) / max(v, 1e-3 * b["grad_norm"]), k) for k, v in b["g1"].items()), reverse=True)[:3]
            ok = rel < 3e-2 and gr < 2e-2 and worst[0][0] < 0.2
            if True:
                print("   worst parameters (first-step gradient norms):", [(k, "%.3f" % r) for r, k in worst])
            bad += 0 if ok else 1
            print(f"img {img:4d} batch {batch:3d}: losses {['%.4f' % v for v in a['losses']]} vs {['%.4f' % v for v in b['losses']]}  "
                  f"max rel {rel:.2e}  grad-norm rel {gr:.2e}  {'ok' if ok else 'MISMATCH'}", flush=True)
    sys.exit(1 if bad else 0)
