from .transformer_qwenimage import QwenImageTransformer2DModel  # noqa: F401
from .transformer_flux import FluxTransformer2DModel  # noqa: F401
