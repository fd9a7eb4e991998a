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
    # conditioning-head linears switch the head to the adapted launch sequence (cond_hip.py)
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


@pytest.mark.parametrize("sharded", [False, True])
def test_from_pretrained_reads_a_diffusers_layout_checkpoint(tmp_path, sharded):
    """ModelMixin.from_pretrained as the reference's loaders call it (load_model.py:34-47, flux_kontext_loader.py:168-175):
    <root>/transformer/config.json + diffusion_pytorch_model.safetensors (or the sharded index form)."""
    import json
    from safetensors.torch import save_file
    from common import fill_weights
    from qflux_amd.models import FluxTransformer2DModel, QwenImageTransformer2DModel
    for cls, cfg in ((QwenImageTransformer2DModel, TINY), (FluxTransformer2DModel, dict(FLUX_TINY, guidance_embeds=True))):
        src = cls(**cfg)
        fill_weights(src, seed=4)
        root = tmp_path / cls.__name__ / ("s" if sharded else "m")
        os.makedirs(root / "transformer")
        conf = dict(cfg, _class_name=cls.__name__, _diffusers_version="0.36.0", axes_dims_rope=list(cfg["axes_dims_rope"]))
        with open(root / "transformer" / "config.json", "w") as f:
            json.dump(conf, f)
        sd = {k: v.detach().clone().contiguous() for k, v in src.state_dict().items()}
        if sharded:
            keys = sorted(sd)
            half = len(keys) // 2
            parts = {"diffusion_pytorch_model-00001-of-00002.safetensors": keys[:half], "diffusion_pytorch_model-00002-of-00002.safetensors": keys[half:]}
            for fn, ks in parts.items():
                save_file({k: sd[k] for k in ks}, str(root / "transformer" / fn))
            with open(root / "transformer" / "diffusion_pytorch_model.safetensors.index.json", "w") as f:
                json.dump({"metadata": {}, "weight_map": {k: fn for fn, ks in parts.items() for k in ks}}, f)
        else:
            save_file(sd, str(root / "transformer" / "diffusion_pytorch_model.safetensors"))
        m = cls.from_pretrained(str(root), subfolder="transformer", torch_dtype=torch.bfloat16, use_safetensors=True,
                                attn_implementation="flash_attention_2", device_map="cpu")
        assert type(m) is cls and m.config.num_layers == cfg["num_layers"]
        for (n, a), (_, b) in zip(sorted(src.state_dict().items()), sorted(m.state_dict().items())):
            assert torch.equal(a, b), n
    with pytest.raises(FileNotFoundError):
        QwenImageTransformer2DModel.from_pretrained("Qwen/Qwen-Image-Edit", subfolder="transformer")


def test_peft_config_state_dict_rule_and_set_adapter():
    """What the reference's UNCHANGED trainer touches around the adapter: `peft_config[adapter_name]` for
    peft.get_peft_model_state_dict (base_trainer.py:870-872), set_adapter (base_trainer.py:940-941)."""
    from qflux_amd.lora_io import get_lora_state_dict, get_peft_model_state_dict
    from qflux_amd.modules import LoraConfig
    q, f = _models()
    cfg = LoraConfig(r=4, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"])
    q.add_adapter(cfg, adapter_name="lora_edit")
    assert q.peft_config == {"lora_edit": cfg} and q._hf_peft_config_loaded
    assert (cfg.peft_type, cfg.bias, cfg.use_dora, cfg.is_prompt_learning) == ("LORA", "none", False, False)
    # peft's filtering rule on the drop-in's state dict == the PEFT-style file this repo writes (adapter name stripped)
    sd = get_peft_model_state_dict(q, adapter_name="lora_edit")
    want = get_lora_state_dict(q, style="peft")
    assert set(sd) == set(want) and len(sd) == 2 * 4 * TINY["num_layers"]
    assert "transformer_blocks.0.attn.to_q.lora_A.weight" in sd and all(".lora_edit" not in k for k in sd)
    for k in sd:
        assert torch.equal(sd[k].cpu(), want[k])
    with pytest.raises(KeyError):
        get_peft_model_state_dict(q, adapter_name="nope")
    # an arbitrary config object is kept as is (a real peft.LoraConfig passes through: only r / lora_alpha / init_lora_weights /
    # target_modules are read here, the rest is peft's business)
    class Foreign:
        r, lora_alpha, init_lora_weights, target_modules = 4, 4, "gaussian", ["to_q"]
        peft_type, bias, use_dora = "LORA", "none", False
    fc = Foreign()
    f.add_adapter(fc, adapter_name="x")
    assert f.peft_config["x"] is fc
    # set_adapter: known adapter = no-op for the plans (same version); unknown adapter raises like peft; a switch invalidates
    v = q._version
    q.set_adapter("lora_edit")
    assert q._version == v
    q.set_adapter(["lora_edit"])
    assert q._version == v
    with pytest.raises(ValueError):
        q.set_adapter("other")
    m = q.transformer_blocks[0].attn.to_q
    m.lora_A["second"] = type(m.lora_A["lora_edit"])(torch.zeros_like(m.A), True)
    m.lora_B["second"] = type(m.lora_B["lora_edit"])(torch.zeros_like(m.B), True)
    m.r["second"], m.lora_alpha["second"], m.scaling["second"] = 4, 8, 2.0
    q.set_adapter("second")
    assert m.active_adapter == "second" and q._version == v + 1          # cached plans (baked adapter pointers) are gone


def test_optimizer_weight_decay_is_never_reinterpreted():
    """an explicit weight_decay is honoured or refused, never silently replaced; None = the optimizer class's default."""
    from qflux_amd.trainer import QwenLoraTrainStep
    q, _ = _models()
    assert QwenLoraTrainStep(q).weight_decay == 0.01                                   # torch.optim.AdamW default
    assert QwenLoraTrainStep(q, weight_decay=0.05).weight_decay == 0.05
    assert QwenLoraTrainStep(q, optimizer="adam8bit").weight_decay == 0.0             # bnb Adam8bit default
    assert QwenLoraTrainStep(q, optimizer="adam", weight_decay=0.0).weight_decay == 0.0
    assert QwenLoraTrainStep(q, optimizer="prodigy").weight_decay == 0.0              # prodigyopt default
    assert QwenLoraTrainStep(q, optimizer="prodigy", weight_decay=0.01).weight_decay == 0.01
    for opt in ("adam", "adam8bit"):
        with pytest.raises(NotImplementedError):
            QwenLoraTrainStep(q, optimizer=opt, weight_decay=0.01)                     # bnb would apply L2 decay: refuse, do not drop it
