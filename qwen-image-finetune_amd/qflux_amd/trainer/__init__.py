from .qwen_step import (QwenLoraTrainStep, flowmatch_tables, get_scheduler, map_mask_to_latent,  # noqa: F401
                        optimizer_kwargs_from_config)
from .flux_step import FluxKontextTrainStep  # noqa: F401
