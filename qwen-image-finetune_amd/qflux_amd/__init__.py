"""qflux_amd -- MI355X-native LoRA-training hot path for the qflux (Qwen-Image-Edit / FLUX-Kontext) DiT.

Host-side mirror of the reference's operator surface for this path (module names, state-dict
keys, forward signatures) over the C ABI of libqfx.so (include/qfx.h).  Importing the package
dlopens the HIP library and fails loudly if it is absent -- there is no eager/CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when libqfx.so is missing)

__version__ = "0.1.0"
