"""Shared helpers for the parity tests."""
import os
from collections import OrderedDict
from functools import lru_cache

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(ROOT, "desenet_amd", "cfg", "desenet_s.yaml")


@lru_cache(maxsize=None)
def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def load_cfg():
    with open(CFG) as f:
        return yaml.safe_load(f)


def case_weights(g, case, group="w"):
    """OrderedDict of tensors stored under `<case>/<group>/<key>`."""
    pre = f"{case}/{group}/"
    return OrderedDict((k[len(pre):], torch.from_numpy(g[k])) for k in g.files if k.startswith(pre))


def rel_err(a, b):
    """max |a-b| / (max|b| + tiny): the 'within 1e-3 rel' measure of BASELINE.json, per tensor."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def elem_err(a, b, rtol=1e-3):
    """Element-wise companion of rel_err for fp32 results: the worst |a-b| / (rtol*|b| + rtol*rms(b)) over the tensor.  <= 1 means
    every element is within rtol of its OWN magnitude plus rtol of the tensor's rms (the absolute floor keeps elements that are
    zero by cancellation from demanding infinite relative accuracy).  Unlike max|a-b| / max|b| it does not let the large elements
    of a tensor hide a wrong small one (seg logits near 0, raw w/h channels)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    rms = b.pow(2).mean().sqrt()
    return ((a - b).abs() / (rtol * b.abs() + rtol * rms + 1e-30)).max().item()


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def stats(t):
    t = torch.as_tensor(t).detach().double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.min().item(), t.max().item(), t.numel()], np.float64)


def subsample(t, n=4096):
    f = torch.as_tensor(t).detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]
