"""Checkpoint interchange (CPU, build container): a checkpoint pickled by the REAL reference loads into the mirrored Model
with identical weights; the exported plain state_dict loads back into the reference."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_checkpoint_roundtrip(tmp_path):
    from desenet_amd.checkpoint import export_state_dict, load_reference_checkpoint
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.synth import synthetic_checkpoint
    pt = str(tmp_path / "ref_last.pt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ref_checkpoint.py"), pt], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ckpt = load_reference_checkpoint(pt)
    assert ckpt["epoch"] == 7 and ckpt["updates"] == 123 and abs(ckpt["best_fitness"] - 0.25) < 1e-12
    m = ckpt["model"]
    assert isinstance(m, Model) and isinstance(ckpt["ema"], Model)
    assert m.names == [f"c{i}" for i in range(6)]
    want = Model("desenet_s.yaml", ch=3, nc=6).state_dict()
    synthetic_checkpoint(want)
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if want[k].is_floating_point():
            assert got[k].dtype == torch.float32
            assert torch.equal(got[k], want[k].half().float()), k       # the reference stores fp16
        elif k.endswith("num_batches_tracked"):
            assert int(got[k]) in (0, 1), k        # the reference's 256x256 stride probe (yolo.py:313-315) ticks the used BNs once
        else:
            assert torch.equal(got[k], want[k]), k
    # and back: a plain state_dict the reference-side loader accepts (same keys / shapes / dtypes)
    out = export_state_dict(m, str(tmp_path / "weights_sd.pt"))
    back = torch.load(out, weights_only=True)
    assert list(back.keys()) == list(want.keys()) and all(back[k].shape == want[k].shape for k in want)
