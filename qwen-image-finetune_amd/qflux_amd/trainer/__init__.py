from .qwen_step import QwenLoraTrainStep, flowmatch_tables  # noqa: F401
