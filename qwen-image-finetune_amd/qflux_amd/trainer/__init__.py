from .qwen_step import QwenLoraTrainStep, flowmatch_tables  # noqa: F401
from .flux_step import FluxKontextTrainStep  # noqa: F401
