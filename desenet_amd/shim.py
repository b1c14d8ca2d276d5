"""Run the reference's scripts unmodified on this package: `install()` imports the REFERENCE's own modules
(`core.models.yolo`, `core.models.common`, `core.utils.general`, `core.utils.loss`, `core.utils.torch_utils`) and rebinds, in
place, exactly the names the hot path owns to their desenet_amd mirrors -- everything else in those modules (dataset code,
plots, CLI helpers) stays the reference's.  After it,

    from core.models.yolo import Model                       (scripts/train.py:34)   -> desenet_amd's Model
    yaml strings resolved by parse_model                     (core/models/yolo.py:451) -> desenet_amd's table
    attempt_load(weights) / torch.load(...)['model']         (experimental.py:85-92, train.py:125-128): pickled
        `core.models.yolo.Model` / `core.models.common.*` objects unpickle AS the mirrored classes (a mirrored Model adopts an
        unpickled tree: desenet_amd.core.models.yolo.Model.__setstate__)
    non_max_suppression, ComputeLoss, SegmentationLosses, ModelEMA                    -> the HIP-backed mirrors

Call it before the scripts import their model code, e.g. `python -c "import desenet_amd.shim as s; s.install(); import
runpy; runpy.run_module('scripts.train', run_name='__main__')"` from the reference checkout.

Limitation: scripts/train.py:427-438 pickles `deepcopy(model).half()` itself; under the shim those objects are the mirrored
classes, so the last.pt / best.pt it writes need desenet_amd to be read back (a plain reference checkout cannot).  For checkpoints
that travel to the plain reference use desenet_amd.checkpoint.save_reference_checkpoint ("checkpoints")."""
from __future__ import annotations

import importlib

_COMMON = ("autopad", "Conv", "Focus", "Bottleneck", "C3", "SPP", "Concat", "FFM", "RFB2", "PyramidPooling")
_YOLO = ("Model", "Detect", "SegMaskPSP", "parse_model")


def install(ema: bool = True, losses: bool = True, nms: bool = True):
    """Rebind the hot-path names inside the imported reference modules; returns {(reference module, name): replaced object}."""
    from .core.models import common, yolo
    from .core.utils import general, loss, torch_utils
    ref_common = importlib.import_module("core.models.common")
    ref_yolo = importlib.import_module("core.models.yolo")
    replaced = {}

    def bind(mod, name, obj):
        replaced[(mod.__name__, name)] = getattr(mod, name, None)
        setattr(mod, name, obj)

    for n in _COMMON:
        bind(ref_common, n, getattr(common, n))
        if hasattr(ref_yolo, n):                     # yolo.py:19 `from core.models.common import *`
            bind(ref_yolo, n, getattr(common, n))
    for n in _YOLO:
        bind(ref_yolo, n, getattr(yolo, n))
    if nms:
        bind(importlib.import_module("core.utils.general"), "non_max_suppression", general.non_max_suppression)
    if losses:
        ref_loss = importlib.import_module("core.utils.loss")
        bind(ref_loss, "ComputeLoss", loss.ComputeLoss)
        bind(ref_loss, "SegmentationLosses", loss.SegmentationLosses)
    if ema:
        bind(importlib.import_module("core.utils.torch_utils"), "ModelEMA", torch_utils.ModelEMA)
    return replaced
