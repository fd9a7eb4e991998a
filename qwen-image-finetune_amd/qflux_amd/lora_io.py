"""LoRA checkpoint I/O in the formats the reference reads and writes.

Reference: BaseTrainer.save_lora (src/qflux/trainer/base_trainer.py:858-875) =
get_peft_model_state_dict -> convert_state_dict_to_diffusers -> Pipeline.save_lora_weights, i.e. a
`pytorch_lora_weights.safetensors` with keys `transformer.<module>.lora.down.weight` / `.lora.up.weight`
(docs/guide/lora.md:171-180); the loader accepts that DIFFUSERS style and the PEFT style
`<module>.lora_A[.<adapter>].weight` (src/qflux/utils/lora_utils.py:12-22, base_trainer.py:977-999).
"""
from __future__ import annotations

import os
import re

import torch
from safetensors.torch import load_file, save_file

from .modules import LoraConfig, QfxLoraLinear

WEIGHT_NAME = "pytorch_lora_weights.safetensors"


def classify_lora_keys(keys) -> str:
    """Same labels as the reference's classify_lora_weight (src/qflux/utils/lora_utils.py:12-22) on a file's key list; pinned by
    tests/golden/ref_lora_classify.json (labels assigned by the reference's own function)."""
    keys = list(keys)
    peft = any(re.search(r"\.lora_[AB](\.|$)", k) for k in keys)
    diff = any(".lora.down.weight" in k or ".lora.up.weight" in k for k in keys)
    if peft and not diff:
        return "PEFT"
    if diff:
        return "DIFFUSERS(attn-processor)" if any(".processor" in k for k in keys) else "DIFFUSERS"
    return "UNKNOWN"


# diffusers.utils.convert_state_dict_to_diffusers (PEFT -> DIFFUSERS table, third party, restated -- parity unpinned): only the
# attention projections below are renamed to `.lora.down/.lora.up`; every other adapted module keeps `.lora_A/.lora_B.weight`.
# A file the reference saves with broad targets (all-linear) therefore carries BOTH spellings; the loaders accept either per key.
_DIFFUSERS_RENAMED = ("to_q", "to_k", "to_v", "to_out.0")


def get_lora_state_dict(model, style: str = "diffusers", prefix: str = "transformer.") -> dict:
    out = {}
    for name, m in model.named_modules():
        if isinstance(m, QfxLoraLinear):
            a, b = m.A.detach().cpu().contiguous(), m.B.detach().cpu().contiguous()
            if style == "diffusers" and name.endswith(_DIFFUSERS_RENAMED):
                out[f"{prefix}{name}.lora.down.weight"] = a
                out[f"{prefix}{name}.lora.up.weight"] = b
            elif style == "diffusers":
                out[f"{prefix}{name}.lora_A.weight"] = a
                out[f"{prefix}{name}.lora_B.weight"] = b
            else:  # peft (adapter name stripped, as get_peft_model_state_dict does)
                out[f"{name}.lora_A.weight"] = a
                out[f"{name}.lora_B.weight"] = b
    return out


def get_peft_model_state_dict(model, state_dict: dict | None = None, adapter_name: str = "default") -> dict:
    """peft.utils.get_peft_model_state_dict for a LoRA config with bias="none" (third party, restated -- parity unpinned), i.e. what
    BaseTrainer.save_lora feeds to convert_state_dict_to_diffusers (base_trainer.py:870-872):
        config = model.peft_config[adapter_name]                                  (KeyError for an unknown adapter, like peft)
        keep   = {k: v for k, v in model.state_dict().items() if "lora_" in k}   (bias == "none")
        keep   = {k: v for k in keep if "lora_" in k and adapter_name in k}
        return {k.replace(f".{adapter_name}", ""): v}
    The drop-in DiTs satisfy the same rule under the real function: `peft_config[adapter_name]` exists after add_adapter and the
    state-dict keys are `<module>.lora_A.<adapter>.weight` / `<module>.lora_B.<adapter>.weight`."""
    cfg = model.peft_config[adapter_name]
    if getattr(cfg, "bias", "none") != "none" or getattr(cfg, "use_dora", False):
        raise NotImplementedError("only LoRA configs with bias='none' and no DoRA are produced by the reference (base_trainer.py:932-937)")
    sd = model.state_dict() if state_dict is None else state_dict
    keep = {k: v for k, v in sd.items() if "lora_" in k and adapter_name in k}
    return {k.replace(f".{adapter_name}", ""): v for k, v in keep.items()}


def save_lora_weights(model, save_folder: str, style: str = "diffusers") -> str:
    os.makedirs(save_folder, exist_ok=True)
    path = os.path.join(save_folder, WEIGHT_NAME)
    alphas = {n: str(m.lora_alpha[m.active_adapter]) for n, m in model.named_modules() if isinstance(m, QfxLoraLinear)}
    save_file(get_lora_state_dict(model, style), path, metadata={"format": "pt", "lora_alpha": repr(alphas)})
    return path


def _normalise(sd: dict) -> dict:
    """-> {module_name: {"A": tensor, "B": tensor}} from either key style."""
    mods: dict = {}
    for k, v in sd.items():
        key = k[len("transformer."):] if k.startswith("transformer.") else k
        m = re.match(r"(.*)\.lora\.(down|up)\.weight$", key)
        if m:
            mods.setdefault(m.group(1), {})["A" if m.group(2) == "down" else "B"] = v
            continue
        m = re.match(r"(.*)\.lora_(A|B)(?:\.[^.]+)?\.weight$", key)
        if m:
            mods.setdefault(m.group(1), {})[m.group(2)] = v
    return mods


def load_lora_adapter(model, path: str, adapter_name: str = "default", lora_alpha: float | None = None):
    """Create (if needed) and fill adapters from a DIFFUSERS- or PEFT-style safetensors file or folder."""
    if os.path.isdir(path):
        path = os.path.join(path, WEIGHT_NAME)
    mods = _normalise(load_file(path))
    if not mods:
        raise ValueError(f"no LoRA weights recognised in {path}")
    existing = {n: m for n, m in model.named_modules() if isinstance(m, QfxLoraLinear)}
    missing = [n for n in mods if n not in existing]
    if missing:
        r = next(iter(mods.values()))["A"].shape[0]
        model.add_adapter(LoraConfig(r=r, lora_alpha=lora_alpha if lora_alpha is not None else r, target_modules=list(missing)), adapter_name)
        existing = {n: m for n, m in model.named_modules() if isinstance(m, QfxLoraLinear)}
    with torch.no_grad():
        for n, ab in mods.items():
            m = existing[n]
            if tuple(ab["A"].shape) != tuple(m.A.shape) or tuple(ab["B"].shape) != tuple(m.B.shape):
                raise ValueError(f"LoRA shape mismatch for {n}: file {tuple(ab['A'].shape)}/{tuple(ab['B'].shape)}")
            m.A.copy_(ab["A"].to(m.A.device, torch.float32))
            m.B.copy_(ab["B"].to(m.B.device, torch.float32))
    return sorted(mods)
