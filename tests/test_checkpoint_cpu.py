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


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_loads_our_checkpoint(tmp_path):
    """The other direction (train.py:125-131, experimental.py:91-92): a checkpoint written by save_reference_checkpoint is
    unpickled by the REAL reference as its own classes and runs its forward with identical results; optimizer / epoch /
    updates travel in train.py's dict layout.  It also loads back into this package."""
    from desenet_amd.checkpoint import load_reference_checkpoint, save_reference_checkpoint
    from desenet_amd.core.models.yolo import Model
    from desenet_amd.parallel import sgd_param_groups
    from desenet_amd.synth import synthetic_checkpoint
    m = Model("desenet_s.yaml", ch=3, nc=6)
    sd = m.state_dict()
    synthetic_checkpoint(sd)
    m.load_state_dict(sd)
    opt = torch.optim.SGD(sgd_param_groups(m), lr=0.0, momentum=0.937, nesterov=True)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()                                           # creates momentum buffers (lr 0: the weights stay the hash fill)
    import sys
    before = {k: sys.modules.get(k) for k in ("core", "core.models", "core.models.yolo", "core.models.common")}
    pt = save_reference_checkpoint(str(tmp_path / "ours_last.pt"), m, ema=m, optimizer=opt, epoch=3, best_fitness=0.5, updates=17)
    assert {k: sys.modules.get(k) for k in before} == before, "save_reference_checkpoint must not leave stub modules behind"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ref_loads_checkpoint.py"), pt],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    back = load_reference_checkpoint(pt)
    assert back["epoch"] == 3 and isinstance(back["model"], Model)
    got = back["model"].state_dict()
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            assert torch.equal(got[k], v.half().float()), k


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_shim_makes_reference_imports_and_loaders_hand_out_the_mirror(tmp_path):
    """INTEGRATION.md 2 ("scripts run unmodified"): with desenet_amd.shim.install() the reference's own import lines
    (train.py:34,48,53), its checkpoint block (train.py:125-131) and attempt_load (experimental.py:85-92) produce mirrored,
    fused, plan-carrying models from a checkpoint pickled by the real reference."""
    pt = str(tmp_path / "ref_last.pt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ref_checkpoint.py"), pt], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_shim.py"), pt], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
