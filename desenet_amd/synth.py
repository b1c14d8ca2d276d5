"""Deterministic synthetic weights and batches (no dataset, no checkpoint: there is no network).

Every state_dict tensor is overwritten by a closed-form counter hash so that golden fixtures stay tiny:
the generator (tools/gen_golden.py, which imports the reference), the oracle, the tests, bench.py and
smoke() all call :func:`hash_fill_state_dict` and obtain bit-identical fp32 weights.

The fill is splitmix64(crc32(key) << 32 | flat_index) -> top 24 bits -> u in [0, 1) (exact in fp32),
mapped to a per-tensor range chosen from the key name:

* conv / linear weights   U(-a, a), a = gain * sqrt(3 / fan_in)   (variance gain^2 / fan_in)
* BatchNorm weight        U(0.5, 1.5)        bias U(-0.1, 0.1)
* BatchNorm running_mean  U(-0.1, 0.1)       running_var U(0.5, 1.5)
* conv bias (Detect / seg classifier)        U(-0.5, 0.5)
* anchors, anchor_grid, num_batches_tracked  left untouched
"""
from __future__ import annotations

import math
import os
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
CONV_GAIN = 1.0
BN_CALIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg", "desenet_s_bn_calib.npz")


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def hash_uniform(ordinal: int, numel: int, lo: float, hi: float) -> np.ndarray:
    """fp32 vector of `numel` values in [lo, hi), a pure function of (ordinal, index)."""
    idx = np.arange(numel, dtype=np.uint64) + (np.uint64(ordinal) << np.uint64(32))
    u = (splitmix64(idx) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (np.float32(lo) + np.float32(hi - lo) * u).astype(np.float32)


def _range_for(key: str, t: torch.Tensor, sd: dict):
    if key.endswith(("num_batches_tracked", "anchors", "anchor_grid")):
        return None
    if key.endswith("running_mean"):
        return (-0.1, 0.1)
    if key.endswith("running_var"):
        return (0.5, 1.5)
    if t.dim() == 4:  # conv weight [co, ci/g, kh, kw]
        fan_in = t.shape[1] * t.shape[2] * t.shape[3]
        a = CONV_GAIN * math.sqrt(3.0 / fan_in)
        return (-a, a)
    if key.endswith("weight") and t.dim() == 1:  # BatchNorm gamma
        return (0.5, 1.5)
    if key.endswith("bias") and t.dim() == 1:
        is_bn = (key[: -len("bias")] + "running_mean") in sd
        return (-0.1, 0.1) if is_bn else (-0.5, 0.5)
    raise KeyError(f"no fill rule for {key} {tuple(t.shape)}")


def hash_fill_state_dict(sd: dict) -> dict:
    """Overwrite `sd` (an ordered state_dict of fp32 tensors) in place; returns it."""
    for k, t in sd.items():
        r = _range_for(k, t, sd)
        if r is None:
            continue
        ordinal = zlib.crc32(k.encode()) & 0xFFFFFFFF  # keyed by NAME: independent of state_dict order
        v = hash_uniform(ordinal, t.numel(), *r)
        with torch.no_grad():
            t.copy_(torch.from_numpy(v).view(t.shape))
    return sd


def load_bn_calibration(sd: dict, path: str = BN_CALIB) -> dict:
    """Overwrite BatchNorm running statistics with the calibrated ones shipped in cfg/desenet_s_bn_calib.npz.

    Hash-filled running stats make an 80-layer eval-mode network either collapse onto its biases or blow up; the
    calibration file holds the batch statistics of ONE train-mode pass of the hash-filled DeSeNet-s over
    synth_images(2, 320, seed=99) (tools/gen_golden.py: gen_calibration, run on the reference itself), so eval-mode
    activations stay O(1).  BNs that never see batch statistics (PyramidPooling.conv1, quirk Q1) keep the hash fill.
    """
    with np.load(path) as z:
        for k in z.files:
            if k in sd and tuple(sd[k].shape) == tuple(z[k].shape):     # (other widths, e.g. DeSeNet-m, keep the hash fill)
                with torch.no_grad():
                    sd[k].copy_(torch.from_numpy(z[k]))
    return sd


def synthetic_checkpoint(sd: dict) -> dict:
    """hash-filled weights + calibrated BN statistics: the 'random-init checkpoint' every test and bench uses."""
    return load_bn_calibration(hash_fill_state_dict(sd))


def synth_images(batch: int, size, seed: int) -> torch.Tensor:
    """Uniform [0,1) NCHW fp32 images from a seeded CPU generator (C1..C4 inputs)."""
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 3, h, w, generator=g)


def synth_targets(batch: int, size, seed: int, boxes_per_image: int = 8, de_nc: int = 6, p_class1: float = 0.3):
    """Seeded detection targets (n,6)=[img, cls, cx, cy, w, h] (normalised) and seg masks (B,H,W) int64."""
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(seed + 7919)
    n = batch * boxes_per_image
    img = torch.arange(batch).repeat_interleave(boxes_per_image).float()
    cls = torch.randint(0, de_nc, (n,), generator=g).float()
    cxy = 0.1 + 0.8 * torch.rand(n, 2, generator=g)
    wh = 0.05 + 0.35 * torch.rand(n, 2, generator=g)
    det = torch.cat([img[:, None], cls[:, None], cxy, wh], 1)
    seg = (torch.rand(batch, h, w, generator=g) < p_class1).long()
    return det, seg
