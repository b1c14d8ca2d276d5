"""Evaluation arithmetic (val.py process_batch, metrics.py ap_per_class / compute_ap / segmentation counters) against
tests/golden/metrics.npz, which tools/gen_golden.py produced by running the reference's own functions.  Host logic runs on
CPU here; the two HIP kernels behind it (IoU matrix, segmentation counters) are checked on the GPU -- integer results and the
matching are bit-exact, the AP integration to 1e-12."""
import numpy as np
import pytest
import torch

from tests.util import golden


def _cases(prefix):
    g = golden("metrics")
    return g, sorted({k.split("/")[0] for k in g.files if k.startswith(prefix)})


def _check_process_batch(device):
    if device == "cpu":
        from oracle.metrics_ref import process_batch          # the CPU restatement (the product path is HIP-only)
    else:
        from desenet_amd.core.utils.metrics import process_batch
    g, names = _cases("pb")
    iouv = torch.linspace(0.5, 0.95, 10).to(device)
    for n in names:
        det, lab = torch.from_numpy(g[f"{n}/det"]).to(device), torch.from_numpy(g[f"{n}/lab"]).to(device)
        got = process_batch(det, lab, iouv).cpu().numpy()
        assert np.array_equal(got, g[f"{n}/correct"]), n
    empty = process_batch(torch.zeros(0, 6, device=device), torch.from_numpy(g["pb0/lab"]).to(device), iouv)
    assert empty.shape == (0, 10)


def test_process_batch_oracle_vs_reference_golden():
    _check_process_batch("cpu")


def test_ap_per_class_and_compute_ap():
    from desenet_amd.core.utils.metrics import ap_per_class, compute_ap
    g = golden("metrics")
    for tag, pk, tk in (("ap", "ap/pcls", "ap/tcls"), ("ap2", "ap2/pcls", "ap2/tcls")):
        p, r, ap, f1, cls = ap_per_class(g["ap/tp"], g["ap/conf"], g[pk], g[tk])
        assert np.array_equal(cls, g[f"{tag}/cls"])
        for name, v in (("p", p), ("r", r), ("ap", ap), ("f1", f1)):
            np.testing.assert_allclose(v, g[f"{tag}/{name}"], rtol=0, atol=1e-12, err_msg=f"{tag}/{name}")
    for j in range(3):
        a, _, _ = compute_ap(g[f"cap{j}/rec"], g[f"cap{j}/prec"])
        assert abs(a - float(g[f"cap{j}/ap"])) < 1e-12
    with pytest.raises(NotImplementedError):
        ap_per_class(g["ap/tp"], g["ap/conf"], g["ap/pcls"], g["ap/tcls"], plot=True)


def test_seg_counters_oracle_vs_reference_golden():
    from oracle.metrics_ref import seg_counts
    g, names = _cases("seg")
    for n in names:
        correct, labeled, inter, union = seg_counts(torch.from_numpy(g[f"{n}/logits"]), torch.from_numpy(g[f"{n}/target"]),
                                                    int(g[f"{n}/ncls"]))
        assert (correct, labeled) == (int(g[f"{n}/correct"]), int(g[f"{n}/labeled"])), n
        assert np.array_equal(inter, g[f"{n}/inter"]) and np.array_equal(union, g[f"{n}/union"]), n


def _check_seg(device):
    from desenet_amd.core.utils.metrics import SegEvaluator, batch_intersection_union, batch_pix_accuracy
    g, names = _cases("seg")
    for n in names:
        logits, target = torch.from_numpy(g[f"{n}/logits"]).to(device), torch.from_numpy(g[f"{n}/target"]).to(device)
        ncls = int(g[f"{n}/ncls"])
        correct, labeled = batch_pix_accuracy(logits, target)
        inter, union = batch_intersection_union(logits, target, ncls)
        assert (correct, labeled) == (int(g[f"{n}/correct"]), int(g[f"{n}/labeled"])), n
        assert np.array_equal(inter, g[f"{n}/inter"]) and np.array_equal(union, g[f"{n}/union"]), n
        ev = SegEvaluator(ncls)
        ev.update(logits, target)
        ev.update(logits, target)
        pix, miou = ev.result()
        assert abs(pix - correct / (np.spacing(1) + labeled)) < 1e-12
        assert abs(miou - (g[f"{n}/inter"] / (np.spacing(1) + g[f"{n}/union"])).mean()) < 1e-12


@pytest.mark.gpu
def test_process_batch_gpu_iou_kernel():
    _check_process_batch("cuda")
    from desenet_amd import hip_ops as ops
    g = golden("metrics")
    a, b = torch.from_numpy(g["pb1/lab"][:, 1:]), torch.from_numpy(g["pb1/det"][:, :4])
    from oracle.metrics_ref import box_iou
    assert torch.equal(ops.box_iou(a.cuda(), b.cuda()).cpu(), box_iou(a, b))      # same operation order: bit-exact


@pytest.mark.gpu
def test_seg_counters_gpu_kernel():
    _check_seg("cuda")
    from desenet_amd.core.utils.metrics import batch_intersection_union, batch_pix_accuracy
    from oracle.metrics_ref import seg_counts
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(8, 2, 640, 640, generator=g)
    target = torch.randint(0, 2, (8, 640, 640), generator=g)
    c0, l0, i0, u0 = seg_counts(logits, target, 2)
    i1, u1 = batch_intersection_union(logits.cuda(), target.cuda(), 2)
    assert np.array_equal(i0, i1) and np.array_equal(u0, u1)
    assert batch_pix_accuracy(logits.cuda(), target.cuda()) == (c0, l0)


@pytest.mark.gpu
def test_det_evaluator_reproduces_the_val_loop():
    """val.run's per-image statistics + final reduction (val.py:231-262, 279-289) on two synthetic NMS-output batches."""
    from desenet_amd.core.utils.metrics import DetEvaluator
    g = golden("metrics")
    shapes = [((480, 640), ((1.0, 1.0), (0.0, 80.0))), ((720, 1280), ((0.5, 0.5), (0.0, 140.0))), ((640, 640), None)]
    ev = DetEvaluator(nc=6)
    for b in range(2):
        out = [torch.from_numpy(g[f"de{b}/out{si}"]).cuda() for si in range(3)]
        ev.update(out, torch.from_numpy(g[f"de{b}/targets"]).cuda(), (640, 640), shapes)
    mp, mr, map50, map_, nt, ap_class, *_ = ev.result()
    assert ev.seen == int(g["de/seen"])
    assert np.allclose([mp, mr, map50, map_], g["de/summary"], rtol=0, atol=1e-12)
    assert np.array_equal(nt, g["de/nt"]) and np.array_equal(ap_class, g["de/ap_class"])
    empty = DetEvaluator(nc=6)
    assert empty.result()[:4] == (0.0, 0.0, 0.0, 0.0)


@pytest.mark.gpu
def test_segoutput_to_target_and_eval_resize():
    from desenet_amd import hip_ops as ops
    from desenet_amd.core.utils.metrics import segoutput_to_target
    g = golden("metrics")
    lg = torch.from_numpy(g["s2t/logits"]).cuda()
    for j, size in enumerate([None, (40, 56), (33, 17), (10, 14)]):
        assert np.array_equal(segoutput_to_target(lg, size).cpu().numpy(), g[f"s2t/out{j}"]), size     # index work: bit-exact
    for j, size in enumerate([(40, 56), (33, 17), (10, 14), (20, 28)]):
        got = ops.resize_bilinear_nchw(lg, size, align_corners=False).cpu().numpy()
        assert np.allclose(got, g[f"s2t/bil{j}"], rtol=1e-5, atol=1e-5), (size, np.abs(got - g[f"s2t/bil{j}"]).max())   # fp32 bilinear
    ac = ops.resize_bilinear_nchw(lg, (40, 56), align_corners=True).cpu()
    want = torch.nn.functional.interpolate(lg.cpu(), (40, 56), mode="bilinear", align_corners=True)
    assert torch.allclose(ac, want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_seg_validation_loop_over_a_loader():
    """seg_validation (val.py:40-76) with a two-batch in-memory loader: uint8 images in, labels at another resolution."""
    import desenet_amd
    from desenet_amd import hip_ops as ops
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.core.utils.metrics import SegEvaluator, seg_validation
    from tests.util import load_cfg
    desenet_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    m = Model(load_cfg(), ch=3, nc=6).cuda()
    gen = torch.Generator().manual_seed(2)
    loader = [(torch.randint(0, 256, (2, 3, 64, 96), generator=gen, dtype=torch.uint8), None,
               torch.randint(0, 2, (2, 48, 80), generator=gen), None, None) for _ in range(2)]
    miou = seg_validation(m, 2, loader, half_precision=False)
    assert desenet_amd.compute_dtype() == torch.float32 and not m.training
    ev = SegEvaluator(2)
    with torch.no_grad():
        for img, _, tgt, _, _ in loader:
            pred = m(img.cuda())[1]
            ev.update(ops.resize_bilinear_nchw(pred, tgt.shape[1:], align_corners=False), tgt.cuda())
    assert miou == ev.result()[1] and 0.0 <= miou <= 1.0
    assert isinstance(seg_validation(m, 2, loader, half_precision=True), float)        # bf16 path runs, dtype restored
    assert desenet_amd.compute_dtype() == torch.float32
