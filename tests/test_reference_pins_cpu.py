"""SURVEY 8(f1)/(f3) pinned to files and labels produced by the REFERENCE's own code (tests/golden/make_golden.py --f1f3 runs
EmbeddingCacheManager.save_cache_embedding / load_cache, src/qflux/data/cache_manager.py:48-125, and classify_lora_weight,
src/qflux/utils/lora_utils.py:12-22, in the build container; only the resulting data files travel)."""
import filecmp
import glob
import json
import os

import torch
from safetensors.torch import load_file

from common import FLUX_TINY, TINY


def test_dataset_reads_a_cache_written_by_the_reference(golden_dir):
    from qflux_amd.data import CachedEmbeddingDataset, collate_cached
    root = os.path.join(golden_dir, "ref_cache")
    exp = load_file(os.path.join(golden_dir, "ref_cache_expected.safetensors"))
    ds = CachedEmbeddingDataset(root)
    assert len(ds) == 3
    for i in range(3):
        it = ds[i]
        assert it["main_hash"] == f"{i:04x}main" and it["cached"] is True
        keys = sorted(k.split(".", 2)[2] for k in exp if k.startswith(f"load.{i}."))
        assert keys == sorted(k for k, v in it.items() if isinstance(v, torch.Tensor))      # same key set as load_cache: no empty_* keys
        for k in keys:
            assert it[k].dtype == torch.float16 and torch.equal(it[k], exp[f"load.{i}.{k}"]), (i, k)
        assert it["img_shapes"] == [(3, 48, 64)] * (2 if i < 2 else 3)
    # caption dropout == load_cache(replace_empty_embeddings=True, prompt_empty_drop_keys=...) (dataset.py:548-554)
    dd = CachedEmbeddingDataset(root, caption_dropout_rate=1.0, prompt_empty_drop_keys=("empty_prompt_embeds", "empty_prompt_embeds_mask"))
    for i in range(3):
        it = dd[i]
        for k in ("prompt_embeds", "prompt_embeds_mask", "image_latents", "control_latents"):
            assert torch.equal(it[k], exp[f"load_drop.{i}.{k}"]), (i, k)
    b = collate_cached([ds[0], ds[1], ds[2]])
    assert b["image_latents"].shape == (3, 20, 64) and b["control_latents"].shape == (3, 24, 64) and b["prompt_embeds"].shape == (3, 7, 32)


def test_writer_reproduces_the_reference_cache_files(golden_dir, tmp_path):
    """write_cache_sample on the tensors the reference's writer was given: identical metadata files, identical tensor payloads."""
    from qflux_amd.data import write_cache_sample
    root = os.path.join(golden_dir, "ref_cache")
    exp = load_file(os.path.join(golden_dir, "ref_cache_expected.safetensors"))
    hash_of = dict(image_latents="img", control_latents="ctl", prompt_embeds="txt", prompt_embeds_mask="txt")
    for i in range(3):
        tensors = {k.split(".", 2)[2]: v for k, v in exp.items() if k.startswith(f"in.{i}.")}
        order = ["image_latents", "control_latents", "prompt_embeds", "prompt_embeds_mask", "empty_prompt_embeds", "empty_prompt_embeds_mask"]
        tensors = {k: tensors[k] for k in order}
        hashes = {k: (f"{i:04x}{hash_of[k]}" if k in hash_of else "emptyprompt") for k in order}
        write_cache_sample(tmp_path, f"{i:04x}main", tensors, img_shapes=[(3, 48, 64)] * (2 if i < 2 else 3), hashes=hashes)
    ref_files = sorted(os.path.relpath(f, root) for f in glob.glob(os.path.join(root, "**", "*"), recursive=True) if os.path.isfile(f))
    got_files = sorted(os.path.relpath(f, tmp_path) for f in glob.glob(os.path.join(str(tmp_path), "**", "*"), recursive=True) if os.path.isfile(f))
    assert ref_files == got_files
    for rel in ref_files:
        a, b = os.path.join(root, rel), os.path.join(str(tmp_path), rel)
        if rel.endswith(".json"):
            assert filecmp.cmp(a, b, shallow=False), rel          # byte-identical metadata
        else:
            ta, tb = torch.load(a, weights_only=False), torch.load(b, weights_only=False)
            assert ta.dtype == tb.dtype == torch.float16 and torch.equal(ta, tb), rel


def test_lora_file_labels_match_the_reference_classifier(golden_dir, tmp_path):
    from qflux_amd.lora_io import classify_lora_keys
    from qflux_amd.models import FluxTransformer2DModel, QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    with open(os.path.join(golden_dir, "ref_lora_classify.json")) as f:
        pins = json.load(f)
    for v in pins["reference_test_vectors"]:                      # the reference's own known answers (test_lora_utils.py:16-52)
        assert classify_lora_keys(v["keys"]) == v["label"], v
    assert sorted(v["label"] for v in pins["reference_test_vectors"]) == ["DIFFUSERS", "DIFFUSERS(attn-processor)", "PEFT", "UNKNOWN"]
    made = {}
    for e in pins["repo_written_files"]:
        # the files this repo writes today carry exactly the key set the reference's classifier was shown, and get the same label
        key = (e["model"], tuple(e["targets"]))
        if key not in made:
            model = (FluxTransformer2DModel(**FLUX_TINY) if e["model"].startswith("flux") else QwenImageTransformer2DModel(**TINY))
            model.add_adapter(LoraConfig(r=4, lora_alpha=8, target_modules=list(e["targets"])), "lora_edit", generator=torch.Generator().manual_seed(0))
            made[key] = model
        model = made[key]
        if e["style"] == "state_dict":
            keys = sorted(k for k in model.state_dict() if "lora" in k)
        else:
            keys = sorted(load_file(model.save_lora_weights(str(tmp_path / (e["model"] + e["style"])), style=e["style"])).keys())
        assert keys == e["keys"], (e["model"], e["style"])
        assert classify_lora_keys(keys) == e["label"] == ("DIFFUSERS" if e["style"] == "diffusers" else "PEFT")


def test_state_dict_style_file_loads_like_the_reference_peft_branch(tmp_path):
    """The reference's PEFT branch (base_trainer.py:985-990): add the adapter, then load_state_dict(file, strict=False) and refuse
    unexpected keys.  A file of full state-dict names must therefore load into this model with no unexpected keys -- and through
    load_lora_adapter as well."""
    import safetensors.torch as st
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    src = QwenImageTransformer2DModel(**TINY)
    src.add_adapter(LoraConfig(r=4, lora_alpha=8), "lora_edit", generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        for n, p in src.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.02)
    f = str(tmp_path / "model.safetensors")
    st.save_file({k: v.detach().clone() for k, v in src.state_dict().items() if "lora" in k}, f)
    dst = QwenImageTransformer2DModel(**TINY)
    dst.add_adapter(LoraConfig(r=4, lora_alpha=8), "lora_edit")
    missing, unexpected = dst.load_state_dict(st.load_file(f), strict=False)
    assert not unexpected and all("lora" not in m for m in missing)
    d2 = QwenImageTransformer2DModel(**TINY)
    d2.load_lora_adapter(f, adapter_name="lora_edit", lora_alpha=8)
    for (n, a), (_, b), (_, c) in zip(sorted(src.state_dict().items()), sorted(dst.state_dict().items()), sorted(d2.state_dict().items())):
        if "lora" in n:
            assert torch.equal(a, b) and torch.equal(a, c), n
