"""Checkpoint interchange with the reference.

The reference saves PICKLED MODULE OBJECTS: `{'epoch', 'best_fitness', 'model': deepcopy(model).half(), 'ema', 'updates',
'optimizer', 'wandb_id'}` (scripts/train.py:427-435), and loads them back with `torch.load(weights)['model']`
(train.py:125-131, experimental.py:92).  Unpickling that file needs the classes `core.models.yolo.Model`,
`core.models.common.Conv`, ... to be importable.  `load_reference_checkpoint` resolves those names to the mirrored classes
of this package (same attribute names, same `state_dict` keys), then REBUILDS a clean `desenet_amd` Model from the
checkpoint's yaml and copies the (fp32-cast) weights in -- so the returned model carries every internal plan of this
package no matter which version of the reference wrote the file.  Classes the hot path never instantiates resolve to inert
stand-ins (their weights are still readable through `state_dict`).

`save_reference_checkpoint` writes the other direction: a train.py-format dict whose 'model' / 'ema' entries unpickle, inside
the reference, as ITS `core.models.yolo.Model` / `core.models.common.*` objects (every reference loader needs that:
train.py:125-129 `ckpt['model'].yaml` / `.float().state_dict()`, experimental.py:91-92 `attempt_load`, general.py:755
`strip_optimizer`), plus epoch / best_fitness / updates / optimizer, so a run can be resumed by either side.

`export_state_dict` writes a plain `{key: tensor}` file; the reference has no loader for it -- it is for a manual
`model.load_state_dict(torch.load(path))` on a model the caller built (keys and shapes are the reference's).
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


# ---- writing the reference's format ---------------------------------------------------------------------------------------
# our class -> (reference module, reference class name).  Classes that only exist here map to the stock torch class they extend.
def _ref_name(cls):
    from .core.models import common, yolo
    if cls.__module__ == common.__name__:
        if cls is common.Upsample:
            return ("torch.nn.modules.upsampling", "Upsample")
        if cls is common._ConvBnAct:
            return ("torch.nn.modules.container", "Sequential")
        return ("core.models.common", cls.__name__)
    if cls.__module__ == yolo.__name__:
        if cls is yolo._Bilinear:
            return ("torch.nn.modules.upsampling", "Upsample")
        return ("core.models.yolo", cls.__name__)
    return None


def _as_reference_objects(module: nn.Module, stubs: dict) -> nn.Module:
    """Shallow re-typed copy of a mirrored module tree: every object of a desenet_amd class becomes an instance of a stub class
    that pickles under the reference's module path; tensors are shared, package-private plans (`_dsn_*`, `_cat_slot`) dropped."""
    ref = _ref_name(type(module))
    if ref is None:
        cls = type(module)
    elif ref[0].startswith("torch."):
        import importlib
        cls = getattr(importlib.import_module(ref[0]), ref[1])
    else:
        cls = stubs.get(ref)
        if cls is None:
            cls = stubs[ref] = type(ref[1], (nn.Module,), {"__module__": ref[0], "__qualname__": ref[1]})
    new = object.__new__(cls)
    d = {k: v for k, v in module.__dict__.items() if not (k.startswith("_dsn") or k in ("_cat_slot", "seg_index", "yaml_file_"))}
    d["_modules"] = type(module._modules)((k, _as_reference_objects(v, stubs) if v is not None else None)
                                           for k, v in module._modules.items())
    new.__dict__ = d
    return new


def save_reference_checkpoint(path: str, model: nn.Module, ema=None, optimizer=None, epoch: int = -1,
                              best_fitness=None, updates=None, half: bool = True, wandb_id=None):
    """Write `{'epoch', 'best_fitness', 'model', 'ema', 'updates', 'optimizer', 'wandb_id'}` exactly as scripts/train.py:427-435
    does -- 'model' = deepcopy(model).half(), 'ema' = deepcopy(ema.ema).half() -- pickled so that the REFERENCE unpickles its own
    classes (no desenet_amd import needed there).  `ema`: a ModelEMA (its .ema / .updates are used) or a module; `optimizer`: an
    optimizer or its state_dict."""
    import sys
    stubs: dict = {}

    def conv(m):
        if m is None:
            return None
        m = copy.deepcopy(m).cpu()
        m = m.half() if half else m.float()
        return _as_reference_objects(m, stubs)

    ema_mod = getattr(ema, "ema", ema)
    ckpt = {"epoch": epoch, "best_fitness": best_fitness, "model": conv(model), "ema": conv(ema_mod),
            "updates": updates if updates is not None else getattr(ema, "updates", None),
            "optimizer": optimizer.state_dict() if hasattr(optimizer, "state_dict") else optimizer, "wandb_id": wandb_id}
    # pickle stores classes by name and checks that the name resolves to the very object being saved: publish the stubs under
    # the reference's module paths for the duration of the save, then put back whatever was there
    saved = {}
    names = sorted({mod for mod, _ in stubs}) + ["core", "core.models"]
    try:
        for name in names:
            saved[name] = sys.modules.get(name)
        for name in ("core", "core.models", *sorted({mod for mod, _ in stubs})):
            sys.modules[name] = types.ModuleType(name)
        for (mod, cname), cls in stubs.items():
            setattr(sys.modules[mod], cname, cls)
        torch.save(ckpt, path)
    finally:
        for name, old in saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
    return path


def export_state_dict(model: nn.Module, path: str, half: bool = False):
    """Plain `{key: tensor}` file (no class pickling) for a manual `model.load_state_dict(torch.load(path))`; the reference's
    own loaders expect pickled modules -- use save_reference_checkpoint for those."""
    sd = {k: (v.detach().cpu().half() if (half and v.is_floating_point()) else v.detach().cpu()) for k, v in model.state_dict().items()}
    torch.save(sd, path)
    return path
