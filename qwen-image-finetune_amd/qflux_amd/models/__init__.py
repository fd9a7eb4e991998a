from .transformer_qwenimage import QwenImageTransformer2DModel  # noqa: F401
