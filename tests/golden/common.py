"""Shared helpers for the golden-vector generator and the tests that read the vectors.

Weights are NOT stored in the fixtures (a tiny 2-block DiT is already 20 MB in fp32); they are
regenerated bit-exactly from a seed with an integer hash that does not depend on torch's RNG
implementation, so the same tensors come out in the build container and on the GPU box.
"""
from __future__ import annotations

import torch

# tiny config = the reference's own test config (tests/src/models/test_qwen_per_sample_rope.py:118-133)
TINY = dict(patch_size=2, in_channels=64, out_channels=16, num_layers=2, attention_head_dim=64,
            num_attention_heads=4, joint_attention_dim=512, axes_dims_rope=(8, 28, 28))


# tiny FLUX config = the reference's own test config (tests/src/models/test_flux_per_sample_rope.py:266-278), 2 single blocks
FLUX_TINY = dict(patch_size=1, in_channels=64, out_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=64,
                 num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16, guidance_embeds=False,
                 axes_dims_rope=(8, 28, 28))


def det_uniform(shape, seed: int) -> torch.Tensor:
    """Deterministic U(-1,1) fp32 tensor: 24-bit integer hash of the flat index (exact in fp32)."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.int64)
    h = (i * 2654435761 + (seed + 1) * 40503) & 0xFFFFFFFF
    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
    h = (h ^ (h >> 13)) * 3266489917 & 0xFFFFFFFF
    h = (h ^ (h >> 16)) >> 8  # 24 bits
    return (h.to(torch.float32) / float(1 << 23) - 1.0).reshape(shape)


def fill_weights(model: torch.nn.Module, seed: int = 1) -> None:
    """Fill every parameter deterministically (order = named_parameters order, which is the same
    for the reference model, the oracle and the HIP modules because the names are the same)."""
    with torch.no_grad():
        for k, (n, p) in enumerate(sorted(model.named_parameters(), key=lambda kv: kv[0])):
            u = det_uniform(tuple(p.shape), seed * 7919 + k)
            if "lora_B" in n:
                v = u * 2e-2
            elif "lora_A" in n:
                v = u * (1.7 / 4.0)
            elif p.ndim == 1 and ("norm" in n):
                v = 1.0 + 0.2 * u
            elif p.ndim == 2:
                v = u * (0.9 / p.shape[1] ** 0.5)
            else:
                v = u * 0.08
            p.copy_(v.to(p.dtype))


def weight_checksum(model: torch.nn.Module) -> torch.Tensor:
    s = torch.zeros((), dtype=torch.float64)
    for n, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
        s = s + p.detach().double().abs().sum()
    return s.reshape(1)
