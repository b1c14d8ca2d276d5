"""CPU: the C-ABI library builds, loads, and exports every symbol include/desenet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from tests.util import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "desenet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from desenet_amd import _lib, build
    build.build(verbose=False)
    names = _declared()
    assert len(names) >= 30
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in desenet_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"
    lib = _lib.lib()
    assert lib.dsn_version() == 100
    assert lib.dsn_bn_workspace_bytes(64) > 0 and lib.dsn_nms_workspace_bytes(2, 25200, 6, 1) > 0


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the product path must fail loudly rather than compute on the host."""
    import torch
    from desenet_amd import hip_ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.as_act(torch.zeros(1, 4, 2, 2))
