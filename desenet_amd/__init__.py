"""desenet_amd: DeSeNet's CNN forward/backward hot path on MI355X (gfx950) -- hand-written HIP kernels behind a C ABI
(include/desenet_hip.h), re-exposed as drop-in mirrors of the reference's core.models.common / core.models.yolo.

There is no CPU compute path: importing is cheap and works anywhere, running needs libdesenet_hip.so and an MI355X."""
from .runtime import compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"
