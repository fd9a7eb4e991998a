"""CPU: LoRA injection naming (peft look-alike) and checkpoint round trip in the reference's two key styles."""
import os

import pytest
import torch
from safetensors.torch import load_file

from common import FLUX_TINY, TINY


def _models():
    from qflux_amd.models import FluxTransformer2DModel, QwenImageTransformer2DModel
    return QwenImageTransformer2DModel(**TINY), FluxTransformer2DModel(**FLUX_TINY)


def test_adapter_names_and_trainable_filter():
    from qflux_amd.modules import LoraConfig
    q, f = _models()
    names = q.add_adapter(LoraConfig(r=4, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"]), "lora_edit")
    assert len(names) == 4 * TINY["num_layers"]
    sd = q.state_dict()
    assert "transformer_blocks.0.attn.to_q.base_layer.weight" in sd
    assert sd["transformer_blocks.0.attn.to_q.lora_A.lora_edit.weight"].shape == (4, 256)
    assert sd["transformer_blocks.1.attn.to_out.0.lora_B.lora_edit.weight"].shape == (256, 4)
    for n, p in q.named_parameters():
        assert p.requires_grad == ("lora" in n)                      # qwen_image_edit_trainer.py:312-318
    assert sd["transformer_blocks.0.attn.to_q.lora_B.lora_edit.weight"].abs().max() == 0  # peft: B starts at zero
    # regex target (peft full-match) on the FLUX model: single-block projections too
    names = f.add_adapter(LoraConfig(r=8, lora_alpha=8, target_modules=".*attn[.]to_[qkv]"), "a")
    assert len(names) == 3 * (FLUX_TINY["num_layers"] + FLUX_TINY["num_single_layers"])
    # every Linear of the DiT can carry an adapter (target_modules "all-linear", configs/example_with_sampling.yaml:9): the
    # conditioning-head linears switch the head to the autograd evaluation (cond_torch.py)
    q2, f2 = _models()
    assert not q2.cond_lora
    names = q2.add_adapter(LoraConfig(r=4, target_modules="all-linear"), "b")
    n_lin = 2 + 2 + 1 + 1 + TINY["num_layers"] * (2 + 8 + 4)     # timestep embedder, embedders, norm_out, proj_out, per block mod/attn/ff
    assert len(names) == n_lin and q2.cond_lora
    names = f2.add_adapter(LoraConfig(r=4, target_modules=["norm1.linear", "norm.linear", "proj_mlp"]), "b")
    assert len(names) == FLUX_TINY["num_layers"] + 2 * FLUX_TINY["num_single_layers"] and f2.cond_lora


@pytest.mark.parametrize("targets", [("to_k", "to_q", "to_v", "to_out.0"), ("to_k", "to_q", "to_v", "to_out.0", "net.0.proj", "net.2")])
def test_lora_checkpoint_roundtrip_both_styles(tmp_path, targets):
    from qflux_amd.lora_io import classify_lora_keys
    from qflux_amd.modules import LoraConfig
    q, _ = _models()
    q.add_adapter(LoraConfig(r=4, lora_alpha=8, target_modules=list(targets)), "lora_edit", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in q.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.01)
    ref = {n: p.detach().clone() for n, p in q.named_parameters() if "lora" in n}
    for style in ("diffusers", "peft"):
        path = q.save_lora_weights(str(tmp_path / style), style=style)
        keys = list(load_file(path).keys())
        assert os.path.basename(path) == "pytorch_lora_weights.safetensors"
        assert classify_lora_keys(keys) == ("DIFFUSERS" if style == "diffusers" else "PEFT")
        if style == "diffusers":
            assert "transformer.transformer_blocks.1.attn.to_k.lora.down.weight" in keys   # docs/guide/lora.md:171-180
        q2, _ = _models()
        loaded = q2.load_lora_adapter(str(tmp_path / style), adapter_name="lora_edit", lora_alpha=8)
        assert len(loaded) == 2 * len(targets) + (4 if len(targets) > 4 else 0)    # 2 blocks; net.* matches both streams
        if len(targets) > 4 and style == "diffusers":
            # convert_state_dict_to_diffusers renames the attention projections only; other modules keep the PEFT spelling
            assert "transformer.transformer_blocks.0.img_mlp.net.0.proj.lora_A.weight" in keys
            assert "transformer.transformer_blocks.1.txt_mlp.net.2.lora_B.weight" in keys
        got = {n: p for n, p in q2.named_parameters() if "lora" in n}
        assert set(got) == set(ref)
        for n in ref:
            assert torch.equal(got[n], ref[n]), n
