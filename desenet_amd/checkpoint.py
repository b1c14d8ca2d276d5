"""Checkpoint interchange with the reference.

The reference saves PICKLED MODULE OBJECTS: `{'epoch', 'best_fitness', 'model': deepcopy(model).half(), 'ema', 'updates',
'optimizer', 'wandb_id'}` (scripts/train.py:427-435), and loads them back with `torch.load(weights)['model']`
(train.py:125-131, experimental.py:92).  Unpickling that file needs the classes `core.models.yolo.Model`,
`core.models.common.Conv`, ... to be importable.  `load_reference_checkpoint` resolves those names to the mirrored classes
of this package (same attribute names, same `state_dict` keys), then REBUILDS a clean `desenet_amd` Model from the
checkpoint's yaml and copies the (fp32-cast) weights in -- so the returned model carries every internal plan of this
package no matter which version of the reference wrote the file.  Classes the hot path never instantiates resolve to inert
stand-ins (their weights are still readable through `state_dict`).

`export_state_dict` writes what the reference can read back without this package: a plain `{key: tensor}` file for
`model.load_state_dict(torch.load(path))` / `intersect_dicts` (train.py:129-131).
"""
from __future__ import annotations

import copy
import pickle
import types
from typing import Any, Dict

import torch
import torch.nn as nn

_REF_MODULE_PREFIXES = ("core.models.", "models.", "core.utils.", "utils.")


class _Inert(nn.Module):
    """Stand-in for a reference class outside the hot path (never executed; only unpickled so its tensors stay readable)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("this module came from a reference checkpoint and is outside the DeSeNet hot path")


def _resolve(module: str, name: str):
    from .core.models import common, yolo
    from .core.utils import autoanchor, general, loss, torch_utils
    tail = module.split(".")[-1]
    table = {"common": common, "yolo": yolo, "general": general, "loss": loss, "torch_utils": torch_utils,
             "autoanchor": autoanchor}
    mod = table.get(tail)
    if mod is not None and hasattr(mod, name):
        return getattr(mod, name)
    return type(name, (_Inert,), {"__module__": __name__})


class _RefUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith(_REF_MODULE_PREFIXES):
            return _resolve(module, name)
        return super().find_class(module, name)


_shim = types.SimpleNamespace(Unpickler=_RefUnpickler, load=lambda f, **k: _RefUnpickler(f, **k).load(), __name__="pickle")


def load_reference_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    """Read a checkpoint written by the reference's train.py.  Returns the same dict with 'model' (and 'ema' when present)
    replaced by freshly built `desenet_amd.core.models.yolo.Model`s in fp32 holding the checkpoint's weights."""
    from .core.models.yolo import Model
    ckpt = torch.load(path, map_location=map_location, pickle_module=_shim, weights_only=False)
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise ValueError(f"{path}: not a reference checkpoint (no 'model' entry)")

    def rebuild(obj):
        if obj is None:
            return None
        cfg = copy.deepcopy(obj.yaml)
        sd = {k: (v.float() if v.is_floating_point() else v) for k, v in obj.state_dict().items()}
        nc = cfg.get("de_nc", cfg.get("nc"))
        m = Model(cfg, ch=cfg.get("ch", 3), nc=nc)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise ValueError(f"{path}: state_dict mismatch (missing {list(missing)[:5]}, unexpected {list(unexpected)[:5]})")
        for attr in ("names", "de_names", "se_names", "hyp", "de_nc", "se_nc", "class_weights", "stride"):
            if hasattr(obj, attr) and attr != "stride":
                setattr(m, attr, copy.deepcopy(getattr(obj, attr)))
        return m

    out = dict(ckpt)
    out["model"] = rebuild(ckpt["model"])
    if ckpt.get("ema") is not None:
        out["ema"] = rebuild(ckpt["ema"])
    return out


def export_state_dict(model: nn.Module, path: str, half: bool = False):
    """Plain tensor file the reference reads with `model.load_state_dict(torch.load(path))` (no class pickling)."""
    sd = {k: (v.detach().cpu().half() if (half and v.is_floating_point()) else v.detach().cpu()) for k, v in model.state_dict().items()}
    torch.save(sd, path)
    return path
