"""perceiverio_pytorch_amd: the PerceiverIO forward hot path as hand-written MI355X (gfx950) HIP kernels
behind a C-ABI (include/pio_hip.h), exposed through the reference's own nn.Module surface."""
from .runtime import (get_backend, get_precision_policy, invalidate_packed_weights, set_backend,  # noqa: F401
                      set_precision_policy)
from ._lib import PioError, build, lib  # noqa: F401

__version__ = "0.1.0"
