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
