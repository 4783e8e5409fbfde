"""cactus_amd: MI355X (gfx950) backend for Cactus's quantized hot path.

The product is the C-ABI shared library cactus_amd/lib/libcactus_hip.so
(include/cactus_hip.h).  This package only loads it and passes torch device
tensors' addresses through; there is no CPU or PyTorch fallback -- importing
`cactus_amd.hip` without the built library raises.
"""
from . import hip  # noqa: F401
