#!/usr/bin/env python
"""Generate the committed golden vectors under tests/golden/ by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing here travels to the GPU
box except the resulting *.safetensors data files.

How: the reference's vendored DiT file src/qflux/models/transformer_qwenimage.py is imported
unmodified by file path.  Its third-party imports (diffusers, absent offline) are satisfied by
a throw-away *shim* package written to a temp dir by this script: a ~150-line restatement of
the diffusers primitives the file uses (SURVEY.md section 8(c) semantics table).  The shim is
our code, not reference source; the reference file itself is never copied.

Vectors written (all tiny; the tiny config equals the reference's own test config,
tests/src/models/test_qwen_per_sample_rope.py:118-133):
  qwen_tiny_fwd.safetensors     fp32: inputs, reference forward output (weights: common.fill_weights seed 1)
  qwen_tiny_grad.safetensors    fp32: d(loss)/d(input) and d(loss)/d(to_q.weight) of the reference
  qwen_rope.safetensors         RoPE tables of the reference for 3 shape lists (incl. 2509 3-image)
  qwen_tiny_lora_step.safetensors  ORACLE (not reference; peft is absent => parity unpinned for the
                                LoRA half) fp32 LoRA training step: loss, pred, dA/dB for all targets
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

SHIM = {
    "diffusers/__init__.py": "",
    "diffusers/configuration_utils.py": '''
import functools, inspect
class ConfigMixin:
    pass
def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *a, **kw):
        sig = inspect.signature(init)
        ba = sig.bind(self, *a, **kw); ba.apply_defaults()
        cfg = {k: v for k, v in ba.arguments.items() if k != "self"}
        self.config = type("Cfg", (), cfg)()
        init(self, *a, **kw)
    return wrapper
''',
    "diffusers/loaders/__init__.py": "class FromOriginalModelMixin: pass\nclass PeftAdapterMixin: pass\nclass FluxTransformer2DLoadersMixin: pass\n",
    "diffusers/models/_modeling_parallel.py": "class ContextParallelInput:\n    def __init__(self, **k): pass\nclass ContextParallelOutput:\n    def __init__(self, **k): pass\n",
    "diffusers/models/__init__.py": "",
    "diffusers/models/attention.py": '''
import torch.nn as nn, torch.nn.functional as F
class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__(); self.proj = nn.Linear(dim_in, dim_out, bias=bias); self.approximate = approximate
    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)
class AttentionMixin: pass
class AttentionModuleMixin:
    fused_projections = False
    def set_processor(self, processor): self.processor = processor
class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner = int(dim * mult); dim_out = dim_out or dim
        self.net = nn.ModuleList([GELU(dim, inner, approximate="tanh", bias=bias), nn.Dropout(dropout), nn.Linear(inner, dim_out, bias=bias)])
    def forward(self, x, *a, **k):
        for m in self.net: x = m(x)
        return x
''',
    "diffusers/models/attention_dispatch.py": '''
import torch.nn.functional as F
def dispatch_attention_fn(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, backend=None, parallel_config=None, **kw):
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
    return o.permute(0, 2, 1, 3)
''',
    "diffusers/models/attention_processor.py": '''
import inspect, torch.nn as nn
from .normalization import RMSNorm
class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, added_kv_proj_dim=None, dim_head=64, heads=8,
                 out_dim=None, context_pre_only=None, bias=False, processor=None, qk_norm=None, eps=1e-5):
        super().__init__()
        inner = out_dim if out_dim is not None else dim_head * heads
        self.heads = inner // dim_head; self.processor = processor
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
        self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
        self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, out_dim or query_dim, bias=True), nn.Dropout(0.0)])
        self.to_add_out = nn.Linear(inner, query_dim, bias=True)
        assert qk_norm == "rms_norm"
        self.norm_q = RMSNorm(dim_head, eps=eps); self.norm_k = RMSNorm(dim_head, eps=eps)
        self.norm_added_q = RMSNorm(dim_head, eps=eps); self.norm_added_k = RMSNorm(dim_head, eps=eps)
    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        params = set(inspect.signature(self.processor.__call__).parameters.keys())
        kw = {k: v for k, v in kw.items() if k in params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)
''',
    "diffusers/models/cache_utils.py": "class CacheMixin: pass\n",
    "diffusers/models/embeddings.py": '''
import math, torch, torch.nn as nn
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    return emb
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__(); self.num_channels = num_channels; self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift; self.scale = scale
    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift, scale=self.scale)
class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__(); self.linear_1 = nn.Linear(in_channels, time_embed_dim); self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)
    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))
class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__(); out_features = out_features or hidden_size
        self.linear_1 = nn.Linear(in_features, hidden_size); self.act_1 = nn.SiLU(); self.linear_2 = nn.Linear(hidden_size, out_features)
    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))
class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")
    def forward(self, timestep, pooled_projection):
        timesteps_proj = self.time_proj(timestep)
        timesteps_emb = self.timestep_embedder(timesteps_proj.to(dtype=pooled_projection.dtype))
        return timesteps_emb + self.text_embedder(pooled_projection)
class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.guidance_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")
    def forward(self, timestep, guidance, pooled_projection):
        timesteps_emb = self.timestep_embedder(self.time_proj(timestep).to(dtype=pooled_projection.dtype))
        guidance_emb = self.guidance_embedder(self.time_proj(guidance).to(dtype=pooled_projection.dtype))
        return (timesteps_emb + guidance_emb) + self.text_embedder(pooled_projection)
def get_1d_rotary_pos_embed(dim, pos, theta=10000.0, use_real=False, linear_factor=1.0, ntk_factor=1.0, repeat_interleave_real=True, freqs_dtype=torch.float32):
    theta = theta * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=freqs_dtype, device=pos.device) / dim)) / linear_factor
    freqs = torch.outer(pos, freqs)
    assert use_real and repeat_interleave_real
    return freqs.cos().repeat_interleave(2, dim=1, output_size=freqs.shape[1] * 2).float(), freqs.sin().repeat_interleave(2, dim=1, output_size=freqs.shape[1] * 2).float()
def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1, sequence_dim=2):
    cos, sin = freqs_cis
    assert sequence_dim == 1
    cos = cos[None, :, None, :]; sin = sin[None, :, None, :]
    cos, sin = cos.to(x.device), sin.to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)
''',
    "diffusers/models/modeling_outputs.py": "class Transformer2DModelOutput:\n    def __init__(self, sample): self.sample = sample\n",
    "diffusers/models/modeling_utils.py": "import torch.nn as nn\nclass ModelMixin(nn.Module): pass\n",
    "diffusers/models/normalization.py": '''
import torch, torch.nn as nn
class RMSNorm(nn.Module):
    def __init__(self, dim, eps, elementwise_affine=True, bias=False):
        super().__init__(); self.eps = eps; self.weight = nn.Parameter(torch.ones(dim))
    def forward(self, hidden_states):
        input_dtype = hidden_states.dtype
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.eps)
        if self.weight.dtype in [torch.float16, torch.bfloat16]:
            hidden_states = hidden_states.to(self.weight.dtype)
        return hidden_states * self.weight
class AdaLayerNormZero(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, norm_type="layer_norm", bias=True):
        super().__init__(); self.emb = None; self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)
    def forward(self, x, timestep=None, class_labels=None, hidden_dtype=None, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp
class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, embedding_dim, norm_type="layer_norm", bias=True):
        super().__init__(); self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)
    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa
class AdaLayerNormContinuous(nn.Module):
    def __init__(self, embedding_dim, conditioning_embedding_dim, elementwise_affine=True, eps=1e-5, bias=True, norm_type="layer_norm"):
        super().__init__(); self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_embedding_dim, embedding_dim * 2, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps, elementwise_affine, bias)
    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]
''',
    "diffusers/utils/__init__.py": '''
USE_PEFT_BACKEND = False
class _L:
    def get_logger(self, name):
        import logging as _l; return _l.getLogger(name)
logging = _L()
def scale_lora_layers(*a, **k): pass
def unscale_lora_layers(*a, **k): pass
''',
    "diffusers/utils/torch_utils.py": "def maybe_allow_in_graph(cls): return cls\n",
}


def import_reference_qwen():
    tmp = tempfile.mkdtemp(prefix="qfx_shim_")
    for rel, src in SHIM.items():
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(src)
    sys.path.insert(0, tmp)
    for name in ("qflux", "qflux.models"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "src", *name.split("."))]
        sys.modules[name] = m
    return importlib.import_module("qflux.models.transformer_qwenimage")


def import_reference_flux():
    return importlib.import_module("qflux.models.transformer_flux")


sys.path.insert(0, HERE)
from common import FLUX_TINY, TINY, fill_weights, weight_checksum  # noqa: E402


def tiny_inputs(seed=0, B=2, shapes=((1, 4, 6), (1, 4, 6)), T=5):
    g = torch.Generator().manual_seed(seed)
    S_i = sum(f * h * w for f, h, w in shapes)
    return dict(
        hidden_states=torch.randn(B, S_i, 64, generator=g),
        encoder_hidden_states=torch.randn(B, T, 512, generator=g) * 4,
        timestep=torch.tensor([0.7109, 0.1611][:B]),  # constants of tests/e2e/test_flux_loss.py:119
        mask=torch.ones(B, T, dtype=torch.int64),
    ), [list(map(tuple, shapes))] * B, [T] * B



def import_reference_by_path(*mods):
    """Import reference modules by file path without executing qflux/__init__.py (it needs dotenv + a HF login).  The two
    third-party hash libraries qflux/utils/tools.py imports at module level (imagehash, blake3: absent offline, used only inside
    hashing helpers that the cache save/load path never calls) are satisfied by empty stand-in modules."""
    for name in ("imagehash", "blake3"):
        try:
            importlib.import_module(name)
        except Exception:  # noqa: BLE001
            mod = types.ModuleType(name)
            mod.blake3 = None
            sys.modules[name] = mod
    for pk, sub in (("qflux", ""), ("qflux.utils", "utils"), ("qflux.data", "data")):
        if pk not in sys.modules:
            m = types.ModuleType(pk)
            m.__path__ = [os.path.join(REF, "src", "qflux", sub)]
            sys.modules[pk] = m
    return [importlib.import_module(m) for m in mods]


def make_f1_f3_fixtures():
    """SURVEY 8(f1)/(f3) pins, produced by the reference's OWN code:
      tests/golden/ref_cache/            a 3-sample embedding cache written by EmbeddingCacheManager.save_cache_embedding
                                         (src/qflux/data/cache_manager.py:48-93)
      tests/golden/ref_cache_expected.safetensors   what EmbeddingCacheManager.load_cache returns for each sample (plain and with
                                         replace_empty_embeddings), + the tensors that were handed to the writer
      tests/golden/ref_lora_classify.json   labels given by classify_lora_weight (src/qflux/utils/lora_utils.py:12-22) to (a) the
                                         key sets of the reference's own tests (tests/src/utils/test_lora_utils.py:16-52) and (b) the
                                         files qflux_amd.lora_io writes in both key styles
    """
    import json
    import shutil
    cm, lu = import_reference_by_path("qflux.data.cache_manager", "qflux.utils.lora_utils")
    root = os.path.join(HERE, "ref_cache")
    shutil.rmtree(root, ignore_errors=True)
    mgr = cm.EmbeddingCacheManager(root)
    g = torch.Generator().manual_seed(20260926)
    expected = {}
    samples = []
    for i in range(3):
        S_t = (12, 20, 12)[i]
        S_c = (12, 20, 24)[i]                 # sample 2: two control images
        T = 5 + i
        data = dict(image_latents=torch.randn(S_t, 64, generator=g), control_latents=torch.randn(S_c, 64, generator=g),
                    prompt_embeds=torch.randn(T, 32, generator=g) * 4, prompt_embeds_mask=torch.ones(T, dtype=torch.int64),
                    empty_prompt_embeds=torch.randn(2, 32, generator=g), empty_prompt_embeds_mask=torch.ones(2, dtype=torch.int64))
        hash_maps = dict(image_latents="image_hash", control_latents="control_hash", prompt_embeds="prompt_hash",
                         prompt_embeds_mask="prompt_hash", empty_prompt_embeds="empty_prompt_hash",
                         empty_prompt_embeds_mask="empty_prompt_hash")
        fh = dict(main_hash=f"{i:04x}main", image_hash=f"{i:04x}img", control_hash=[f"{i:04x}ctl"], prompt_hash=f"{i:04x}txt",
                  empty_prompt_hash="emptyprompt")
        shapes = [[3, 48, 64], [3, 48, 64]] if i < 2 else torch.tensor([[3, 48, 64], [3, 48, 64], [3, 48, 64]])
        mgr.save_cache_embedding(data, hash_maps, fh, img_shapes=shapes)
        samples.append((data, fh))
        for k, v in data.items():
            expected[f"in.{i}.{k}"] = v.clone()
    for i, (data, fh) in enumerate(samples):
        got = mgr.load_cache({"file_hashes": {"main_hash": fh["main_hash"]}})
        for k, v in got.items():
            if isinstance(v, torch.Tensor):
                expected[f"load.{i}.{k}"] = v.clone()
        got = mgr.load_cache({"file_hashes": {"main_hash": fh["main_hash"]}}, replace_empty_embeddings=True,
                             prompt_empty_drop_keys=["empty_prompt_embeds", "empty_prompt_embeds_mask"])
        for k, v in got.items():
            if isinstance(v, torch.Tensor):
                expected[f"load_drop.{i}.{k}"] = v.clone()
    assert cm.EmbeddingCacheManager.exist(root)
    save_file({k: v.contiguous() for k, v in expected.items()}, os.path.join(HERE, "ref_cache_expected.safetensors"),
              metadata={"writer": "EmbeddingCacheManager.save_cache_embedding (reference)", "reader": "EmbeddingCacheManager.load_cache"})
    # ---- f3: the reference's classifier on its own test vectors and on the files this repo writes
    import safetensors.torch as st
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    from qflux_amd.modules import LoraConfig
    from qflux_amd.models import FluxTransformer2DModel, QwenImageTransformer2DModel
    out = {"reference_test_vectors": [], "repo_written_files": []}
    tmp = tempfile.mkdtemp()
    vecs = [["model.layer.lora_A", "model.layer.lora_B"], ["model.layer.lora.down.weight", "model.layer.lora.up.weight"],
            ["model.layer.processor.lora.down.weight", "model.layer.processor.lora.up.weight"], ["model.layer.weight"]]
    for j, keys in enumerate(vecs):
        f = os.path.join(tmp, f"v{j}.safetensors")
        st.save_file({k: torch.zeros(2, 2) for k in keys}, f)
        out["reference_test_vectors"].append({"keys": keys, "label": lu.classify_lora_weight(f)})
    for tag, model, targets in (
            ("qwen_default_targets", QwenImageTransformer2DModel(**TINY), ["to_k", "to_q", "to_v", "to_out.0"]),
            ("qwen_attention_and_ff", QwenImageTransformer2DModel(**TINY), ["to_k", "to_q", "to_v", "to_out.0", "net.0.proj", "net.2"]),
            ("flux_attention_both_streams", FluxTransformer2DModel(**FLUX_TINY), ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "to_add_out"])):
        model.add_adapter(LoraConfig(r=4, lora_alpha=8, target_modules=targets), "lora_edit", generator=torch.Generator().manual_seed(0))
        for style in ("diffusers", "peft"):
            path = model.save_lora_weights(os.path.join(tmp, tag + "_" + style), style=style)
            out["repo_written_files"].append({"model": tag, "targets": targets, "style": style, "label": lu.classify_lora_weight(path),
                                              "keys": sorted(st.load_file(path).keys())})
        # the accelerate `model.safetensors` flavour: full state-dict names incl. the adapter name (what the reference's PEFT branch
        # feeds to load_state_dict(strict=False), base_trainer.py:985-990)
        f = os.path.join(tmp, tag + "_statedict.safetensors")
        st.save_file({k: v.detach().clone() for k, v in model.state_dict().items() if "lora" in k}, f)
        out["repo_written_files"].append({"model": tag, "targets": targets, "style": "state_dict", "label": lu.classify_lora_weight(f),
                                          "keys": sorted(st.load_file(f).keys())})
    with open(os.path.join(HERE, "ref_lora_classify.json"), "w") as fjs:
        json.dump(out, fjs, indent=1)
    print("f1/f3: wrote ref_cache/ (reference writer), ref_cache_expected.safetensors (reference reader), ref_lora_classify.json:",
          [(e["model"], e["style"], e["label"]) for e in out["repo_written_files"]])



def _ref_functions(path, names, glb):
    """Compile the named function definitions of a reference file WHERE IT LIES (ast of /root/reference/...; nothing is copied
    into this repository) into namespaces that supply the module-level names they use.  Decorators (@staticmethod / @classmethod)
    are dropped: the callers below pass `self` / `cls` explicitly."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in out:
            node.decorator_list = []
            ns = dict(glb)
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            out[node.name] = ns[node.name]
    missing = set(names) - set(out)
    assert not missing, (path, missing)
    return out


class _Sched:
    """FlowMatchEulerDiscreteScheduler as constructed from the model repo's scheduler_config.json (third party, restated:
    timesteps = linspace(1, N, N)[::-1], sigmas = timesteps / N with a trailing 0; dynamic shifting => unshifted at init)."""

    def __init__(self, n=1000):
        self.config = type("Cfg", (), {"num_train_timesteps": n})()
        self.timesteps = torch.linspace(1, n, n).flip(0)
        self.sigmas = torch.cat([self.timesteps / n, torch.zeros(1)])


class _Rec(torch.nn.Module):
    """Records what the step caller hands to the DiT and what comes back."""

    def __init__(self, dit):
        super().__init__()
        self.dit, self.config, self.calls = dit, dit.config, []

    def forward(self, **kw):
        out = self.dit(**kw)
        self.calls.append((kw, out[0].detach().clone()))
        return out


def make_step_caller_fixtures():
    """SURVEY 8 rows a1 / a13 pinned by EXECUTION: the bodies of the reference's own step callers
         QwenImageEditTrainer._compute_loss (+ _get_sigmas)                qwen_image_edit_trainer.py:777-861
         FluxKontextLoraTrainer._compute_loss_shared_mode                  flux_kontext_trainer.py:494-577
         FluxKontextLoraTrainer._compute_loss_multi_resolution_mode        flux_kontext_trainer.py:579-796
         BaseTrainer.forward_loss / convert_img_shapes_to_latent           base_trainer.py:478-506,184-240
       run here on a stub `self` (cpu accelerator, fp32, the reference's own tiny DiT / loss classes / pad_latents_for_multi_res),
       unbound, compiled from the files where they lie.  The trainer MODULES cannot be imported (diffusers pipelines, transformers,
       peft at module level), their function bodies can.  Third-party names the bodies use are restated: the two diffusers
       training_utils helpers for weighting_scheme="none" (u ~ U(0,1) on the CPU; weighting = ones) and the scheduler tables.
       Written: tests/golden/ref_step_callers.safetensors (inputs, draws, what reached the DiT, prediction, loss)."""
    import copy
    import types as _t
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    ref = import_reference_qwen()
    reff = import_reference_flux()
    refc = importlib.import_module("qflux.models.transformer_flux_custom")
    (tools,) = import_reference_by_path("qflux.utils.tools")
    for m_ in ("qflux.losses",):
        if m_ not in sys.modules:
            pk = types.ModuleType(m_); pk.__path__ = [os.path.join(REF, "src", "qflux", "losses")]; sys.modules[m_] = pk
    mse_mod = importlib.import_module("qflux.losses.mse_loss")
    am_mod = importlib.import_module("qflux.losses.attention_mask_loss")

    def density(weighting_scheme, batch_size, logit_mean=None, logit_std=None, mode_scale=None, device="cpu", generator=None):
        assert weighting_scheme == "none"
        return torch.rand(size=(batch_size,), device=device, generator=generator)     # diffusers.training_utils (third party)

    def weighting_sd3(weighting_scheme, sigmas=None):
        assert weighting_scheme == "none"
        return torch.ones_like(sigmas)                                                  # diffusers.training_utils (third party)

    glb = dict(torch=torch, copy=copy, compute_density_for_timestep_sampling=density, compute_loss_weighting_for_sd3=weighting_sd3,
               pad_latents_for_multi_res=tools.pad_latents_for_multi_res)
    tdir = os.path.join(REF, "src", "qflux", "trainer")
    qf = _ref_functions(os.path.join(tdir, "qwen_image_edit_trainer.py"), ["_compute_loss", "_get_sigmas"], glb)
    bf = _ref_functions(os.path.join(tdir, "base_trainer.py"), ["forward_loss", "convert_img_shapes_to_latent"], glb)
    ids_fn = _ref_functions(os.path.join(tdir, "flux_kontext_trainer.py"), ["_prepare_latent_image_ids"], glb)["_prepare_latent_image_ids"]
    FK = type("FluxKontextLoraTrainer", (), {"_prepare_latent_image_ids": staticmethod(ids_fn)})
    ff = _ref_functions(os.path.join(tdir, "flux_kontext_trainer.py"), ["_compute_loss_shared_mode", "_compute_loss_multi_resolution_mode"],
                        dict(glb, FluxKontextLoraTrainer=FK))

    def stub(dit, criterion):
        s = _t.SimpleNamespace()
        s.accelerator = _t.SimpleNamespace(device=torch.device("cpu"))
        s.weight_dtype = torch.float32
        s.scheduler = _Sched()
        s.dit = _Rec(dit)
        s.criterion = criterion
        s.vae_scale_factor = 8
        s.forward_loss = lambda *a, **k: bf["forward_loss"](s, *a, **k)
        s._get_sigmas = lambda *a, **k: qf["_get_sigmas"](s, *a, **k)
        s._prepare_latent_image_ids = ids_fn
        s.convert_img_shapes_to_latent = lambda *a, **k: bf["convert_img_shapes_to_latent"](s, *a, **k)
        return s

    out = {}
    # ---------------- Qwen: _compute_loss ----------------
    model = ref.QwenImageTransformer2DModel(**TINY, guidance_embeds=False).eval()
    fill_weights(model, seed=1)
    g = torch.Generator().manual_seed(101)
    B, T = 2, 5
    shapes = [(1, 4, 6), (1, 4, 6)]
    emb = dict(image_latents=torch.randn(B, 24, 64, generator=g).half(), control_latents=torch.randn(B, 24, 64, generator=g).half(),
               prompt_embeds=(torch.randn(B, T, 512, generator=g) * 4).half(), prompt_embeds_mask=torch.ones(B, T, dtype=torch.int64),
               img_shapes=[shapes] * B)
    s = stub(model, mse_mod.MseLoss())
    torch.manual_seed(4242)
    loss = qf["_compute_loss"](s, emb)
    torch.manual_seed(4242)                 # replay the two draws of the body in its order: randn_like(image_latents), rand(B)
    noise = torch.randn_like(emb["image_latents"].float())
    u = torch.rand(size=(B,), device="cpu")
    kw, pred = s.dit.calls[0]
    oracle = O.OracleQwenDiT(**TINY)
    oracle.load_state_dict(model.state_dict(), strict=True)
    lo, po = O.qwen_compute_loss(oracle, emb, noise, u, torch.float32, return_pred=True)
    print("qwen _compute_loss (reference body) vs oracle step caller: loss %.3e pred %.3e" % (
        abs(lo.item() - loss.item()), (po - pred[:, :24]).abs().max().item()))
    assert abs(lo.item() - loss.item()) < 1e-6 and (po - pred[:, :24]).abs().max() < 1e-5
    out.update({"qwen.image_latents": emb["image_latents"], "qwen.control_latents": emb["control_latents"],
                "qwen.prompt_embeds": emb["prompt_embeds"], "qwen.noise": noise, "qwen.u": u,
                "qwen.dit_hidden_states": kw["hidden_states"].detach(), "qwen.dit_timestep": kw["timestep"].detach(),
                "qwen.pred": pred, "qwen.loss": loss.detach().reshape(1), "qwen.w_checksum": weight_checksum(model)})
    # ---------------- FLUX: _compute_loss_shared_mode ----------------
    cfg = dict(FLUX_TINY, guidance_embeds=True)
    fm = reff.FluxTransformer2DModel(**cfg).eval()
    fill_weights(fm, seed=3)
    g = torch.Generator().manual_seed(103)
    B, hh, ww, T = 2, 4, 6, 7
    S_t = hh * ww
    ctl = ids_fn(B, hh, ww, "cpu", torch.float32); ctl[:, 0] = 1
    femb = dict(image_latents=torch.randn(B, S_t, 64, generator=g), control_latents=torch.randn(B, S_t, 64, generator=g),
                control_ids=ctl, text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(B, cfg["pooled_projection_dim"], generator=g),
                prompt_embeds=torch.randn(B, T, cfg["joint_attention_dim"], generator=g), image=torch.zeros(B, 3, hh * 16, ww * 16),
                noise=torch.randn(B, S_t, 64, generator=g), timestep=torch.tensor([0.7109, 0.1611]))
    s = stub(fm, mse_mod.MseLoss())
    loss = ff["_compute_loss_shared_mode"](s, femb)
    kw, pred = s.dit.calls[0]
    fo = FO.OracleFluxDiT(**cfg)
    fo.load_state_dict(fm.state_dict(), strict=True)
    lo, po = FO.flux_compute_loss(fo, dict(femb, latent_hw=(hh, ww)), femb["noise"], femb["timestep"], torch.float32, return_pred=True)
    print("flux _compute_loss_shared_mode (reference body) vs oracle: loss %.3e pred %.3e" % (
        abs(lo.item() - loss.item()), (po - pred[:, :S_t]).abs().max().item()))
    assert abs(lo.item() - loss.item()) < 1e-6 and (po - pred[:, :S_t]).abs().max() < 1e-5
    out.update({"flux." + k: v for k, v in femb.items() if k != "image"})
    out.update({"flux.dit_hidden_states": kw["hidden_states"].detach(), "flux.dit_img_ids": kw["img_ids"].detach(),
                "flux.dit_guidance": kw["guidance"].detach(), "flux.pred": pred, "flux.loss": loss.detach().reshape(1),
                "flux.w_checksum": weight_checksum(fm)})
    # ---------------- FLUX: _compute_loss_multi_resolution_mode (ragged: one square, one non-square sample) ----------------
    cm = refc.FluxTransformer2DModel(**cfg).eval()
    fill_weights(cm, seed=3)
    g = torch.Generator().manual_seed(107)
    px = [[(3, 64, 96), (3, 64, 96)], [(3, 80, 48), (3, 48, 80)]]      # per sample: target + control in PIXELS -> 4x6, 4x6 | 5x3, 3x5 tokens
    lat = [[(h // 16, w // 16) for _, h, w in sh] for sh in px]
    n_t = [lat[i][0][0] * lat[i][0][1] for i in range(2)]
    n_c = [sum(a * b for a, b in lat[i][1:]) for i in range(2)]
    il = torch.zeros(2, max(n_t), 64); cl = torch.zeros(2, max(n_c), 64)
    noises = []
    for i in range(2):
        il[i, : n_t[i]] = torch.randn(n_t[i], 64, generator=g)
        cl[i, : n_c[i]] = torch.randn(n_c[i], 64, generator=g)
        noises.append(torch.randn(n_t[i], 64, generator=g))
    T = 7
    memb = dict(image_latents=il, control_latents=cl, text_ids=torch.zeros(T, 3), img_shapes=px,
                pooled_prompt_embeds=torch.randn(2, cfg["pooled_projection_dim"], generator=g),
                prompt_embeds=torch.randn(2, T, cfg["joint_attention_dim"], generator=g), noise=noises,
                timestep=[torch.tensor([0.7109]), torch.tensor([0.1611])])
    s = stub(cm, am_mod.AttentionMaskMseLoss())
    loss, aux = ff["_compute_loss_multi_resolution_mode"](s, memb, return_pred=True)
    fo = FO.OracleFluxDiT(**cfg)
    fo.load_state_dict(cm.state_dict(), strict=True)
    samples = [dict(image_latents=il[i, : n_t[i]], control_latents=cl[i, : n_c[i]], hw=lat[i][0], control_hw=lat[i][1:], noise=noises[i],
                    t=memb["timestep"][i].reshape(())) for i in range(2)]
    lo, po = FO.flux_compute_loss_multires(fo, samples, dict(text_ids=memb["text_ids"], pooled_prompt_embeds=memb["pooled_prompt_embeds"],
                                                             prompt_embeds=memb["prompt_embeds"]), torch.float32, return_pred=True)
    print("flux _compute_loss_multi_resolution_mode (reference body) vs oracle: loss %.3e pred %.3e" % (
        abs(lo.item() - loss.item()), (po - aux["model_pred"]).abs().max().item()))
    assert abs(lo.item() - loss.item()) < 1e-6 and (po - aux["model_pred"]).abs().max() < 1e-5
    out.update({"mr.image_latents": il, "mr.control_latents": cl, "mr.pooled_prompt_embeds": memb["pooled_prompt_embeds"],
                "mr.prompt_embeds": memb["prompt_embeds"], "mr.noise0": noises[0], "mr.noise1": noises[1],
                "mr.timestep": torch.cat(memb["timestep"]), "mr.px_shapes": torch.tensor(px),
                "mr.dit_hidden_states": aux["latent_model_input"].detach(), "mr.dit_img_ids": aux["latent_ids"].detach(),
                "mr.dit_attention_mask": aux["full_attention_mask"].to(torch.uint8), "mr.pred": aux["model_pred"].detach(),
                "mr.loss": loss.detach().reshape(1), "mr.w_checksum": weight_checksum(cm)})
    save_file({k: v.contiguous().clone() for k, v in out.items()}, os.path.join(HERE, "ref_step_callers.safetensors"),
              metadata={"producer": "bodies of the reference's step callers executed on a stub self (make_golden.py --step)",
                        "qwen_cfg": repr(TINY), "flux_cfg": repr(cfg), "qwen_weights": "fill_weights seed 1", "flux_weights": "fill_weights seed 3"})
    print("step callers: wrote ref_step_callers.safetensors")


def main():
    from oracle import qwen_dit as O

    ref = import_reference_qwen()
    torch.manual_seed(0)
    model = ref.QwenImageTransformer2DModel(**TINY, guidance_embeds=False).eval()
    fill_weights(model, seed=1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # ---------------- forward ----------------
    inp, img_shapes, txt_lens = tiny_inputs()
    x = inp["hidden_states"].clone().requires_grad_(True)
    out = model(hidden_states=x, encoder_hidden_states=inp["encoder_hidden_states"],
                encoder_hidden_states_mask=inp["mask"], timestep=inp["timestep"],
                img_shapes=img_shapes, txt_seq_lens=txt_lens, return_dict=False)[0]
    tgt = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    loss = ((out - tgt) ** 2).mean()
    wq = model.transformer_blocks[0].attn.to_q.weight
    wq.requires_grad_(True)
    gx, gw = torch.autograd.grad(loss, [x, wq])

    oracle = O.OracleQwenDiT(**TINY)
    missing, unexpected = oracle.load_state_dict(sd, strict=True)
    xo = inp["hidden_states"].clone().requires_grad_(True)
    oo = oracle(hidden_states=xo, encoder_hidden_states=inp["encoder_hidden_states"],
                encoder_hidden_states_mask=inp["mask"], timestep=inp["timestep"],
                img_shapes=img_shapes, txt_seq_lens=txt_lens)[0]
    lo = ((oo - tgt) ** 2).mean()
    ogx, ogw = torch.autograd.grad(lo, [xo, oracle.transformer_blocks[0].attn.to_q.weight])
    print("oracle vs reference: fwd max|d| = %.3e  gx %.3e  gw %.3e" % (
        (oo - out).abs().max().item(), (ogx - gx).abs().max().item(), (ogw - gw).abs().max().item()))
    assert (oo - out).abs().max() < 1e-5 and (ogx - gx).abs().max() < 1e-6

    fwd = {"w.checksum": weight_checksum(model)}
    fwd.update({"in.hidden_states": inp["hidden_states"], "in.encoder_hidden_states": inp["encoder_hidden_states"],
                "in.timestep": inp["timestep"], "in.mask": inp["mask"], "in.target": tgt,
                "out.sample": out.detach().contiguous()})
    save_file(fwd, os.path.join(HERE, "qwen_tiny_fwd.safetensors"),
              metadata={"img_shapes": repr(img_shapes[0]), "txt_len": str(txt_lens[0]), "cfg": repr(TINY)})
    save_file({"grad.hidden_states": gx.contiguous(), "grad.blocks0_to_q_weight": gw.contiguous(),
               "loss": loss.detach().reshape(1)}, os.path.join(HERE, "qwen_tiny_grad.safetensors"))

    # ---------------- RoPE tables of the reference ----------------
    rope = {}
    pe = ref.QwenEmbedRope(theta=10000, axes_dim=[16, 56, 56], scale_rope=True)
    cases = {"a": ([[(1, 4, 6), (1, 4, 6)]], [5]), "b": ([[(1, 16, 12), (1, 16, 12)]], [7]),
             "c": ([[(1, 6, 4), (1, 8, 8), (1, 2, 10)]], [9])}
    for name, (shapes, tl) in cases.items():
        v, t = pe(shapes, tl, device=torch.device("cpu"))
        ov, ot = O.qwen_rope_tables(shapes[0], tl[0], (16, 56, 56))
        assert torch.allclose(torch.view_as_real(v), torch.view_as_real(ov), atol=1e-6), name
        assert torch.allclose(torch.view_as_real(t), torch.view_as_real(ot), atol=1e-6), name
        rope[f"{name}.vid"] = torch.view_as_real(v).clone().contiguous()
        rope[f"{name}.txt"] = torch.view_as_real(t).clone().contiguous()
    # apply_rotary_emb_qwen known-answer: x=[0..7], angle pi/2 -> [-1,0,-3,2,-5,4,-7,6]
    xk = torch.arange(8.0).view(1, 1, 1, 8)
    fk = torch.polar(torch.ones(1, 4), torch.full((1, 4), torch.pi / 2))
    rope["kat.out"] = ref.apply_rotary_emb_qwen(xk, fk, use_real=False).contiguous()
    save_file(rope, os.path.join(HERE, "qwen_rope.safetensors"), metadata={k: repr(v) for k, v in cases.items()})

    # ---------------- LoRA training step (oracle only: peft absent) ----------------
    names = O.add_lora(oracle, r=4, lora_alpha=8, adapter_name="lora_edit", seed=7)
    fill_weights(oracle, seed=2)
    g = torch.Generator().manual_seed(11)
    B, S_t = 2, 24
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half().float(),
               control_latents=torch.randn(B, S_t, 64, generator=g).half().float(),
               prompt_embeds=(torch.randn(B, 5, 512, generator=g) * 4).half().float(),
               prompt_embeds_mask=torch.ones(B, 5, dtype=torch.int64),
               img_shapes=[[(1, 4, 6), (1, 4, 6)]] * B)
    noise = torch.randn(B, S_t, 64, generator=g)
    u = torch.tensor([0.7109, 0.1611])
    loss, pred = O.qwen_compute_loss(oracle, emb, noise, u, torch.float32, return_pred=True)
    loss.backward()
    step = {"in.image_latents": emb["image_latents"], "in.control_latents": emb["control_latents"],
            "in.prompt_embeds": emb["prompt_embeds"], "in.noise": noise, "in.u": u,
            "out.loss": loss.detach().reshape(1), "out.pred": pred.detach().contiguous(),
            "w.checksum": weight_checksum(oracle)}
    for n, p in oracle.named_parameters():
        if "lora" in n:
            step["g." + n] = p.grad.detach().clone().contiguous()
    save_file(step, os.path.join(HERE, "qwen_tiny_lora_step.safetensors"),
              metadata={"targets": repr(names), "r": "4", "lora_alpha": "8", "adapter": "lora_edit",
                        "pinned": "oracle-only (peft unavailable offline): parity unpinned for LoRA half"})
    # ---------------- FLUX: reference transformer_flux.py vs oracle ----------------
    from oracle import flux_dit as FO
    from common import FLUX_TINY
    reff = import_reference_flux()
    for ge in (False, True):
        cfg = dict(FLUX_TINY, guidance_embeds=ge)
        fm = reff.FluxTransformer2DModel(**cfg).eval()
        fill_weights(fm, seed=3)
        fo = FO.OracleFluxDiT(**cfg)
        fo.load_state_dict(fm.state_dict(), strict=True)
        g = torch.Generator().manual_seed(21)
        B, hh, ww, T = 2, 4, 6, 7
        S_t = hh * ww
        x = torch.randn(B, 2 * S_t, 64, generator=g)
        pe = torch.randn(B, T, cfg["joint_attention_dim"], generator=g)
        pooled = torch.randn(B, cfg["pooled_projection_dim"], generator=g)
        tt = torch.tensor([0.7109, 0.1611])
        gd = torch.ones(B) if ge else None
        lat = FO.prepare_latent_image_ids(hh, ww)
        ctl = lat.clone(); ctl[:, 0] = 1
        img_ids = torch.cat([lat, ctl], 0)
        txt_ids = torch.zeros(T, 3)
        xr = x.clone().requires_grad_(True)
        out = fm(hidden_states=xr, encoder_hidden_states=pe, pooled_projections=pooled, timestep=tt, img_ids=img_ids, txt_ids=txt_ids,
                 guidance=gd, joint_attention_kwargs={}, return_dict=False)[0]
        tgt = torch.randn(out.shape, generator=g)
        lossr = ((out - tgt) ** 2).mean()
        gxr, gwr = torch.autograd.grad(lossr, [xr, fm.single_transformer_blocks[0].attn.to_q.weight.requires_grad_(True)])
        xo = x.clone().requires_grad_(True)
        oo = fo(hidden_states=xo, encoder_hidden_states=pe, pooled_projections=pooled, timestep=tt, img_ids=img_ids, txt_ids=txt_ids,
                guidance=gd)[0]
        lo2 = ((oo - tgt) ** 2).mean()
        gxo, gwo = torch.autograd.grad(lo2, [xo, fo.single_transformer_blocks[0].attn.to_q.weight])
        print("flux(guidance=%s) oracle vs reference: fwd %.3e gx %.3e gw %.3e" % (ge, (oo - out).abs().max().item(),
              (gxo - gxr).abs().max().item(), (gwo - gwr).abs().max().item()))
        assert (oo - out).abs().max() < 1e-5 and (gxo - gxr).abs().max() < 1e-6
        if ge:
            save_file({"in.hidden_states": x, "in.encoder_hidden_states": pe, "in.pooled": pooled, "in.timestep": tt,
                       "in.img_ids": img_ids, "in.txt_ids": txt_ids, "in.target": tgt, "out.sample": out.detach().contiguous(),
                       "grad.hidden_states": gxr.contiguous(), "grad.single0_to_q_weight": gwr.contiguous(),
                       "w.checksum": weight_checksum(fm)}, os.path.join(HERE, "flux_tiny_fwd.safetensors"),
                      metadata={"cfg": repr(cfg), "weights": "common.fill_weights seed 3"})
    # ---------------- FLUX multi-resolution: reference transformer_flux_custom.py vs oracle ----------------
    try:
        refc = importlib.import_module("qflux.models.transformer_flux_custom")
        cfg = dict(FLUX_TINY, guidance_embeds=True)
        cm = refc.FluxTransformer2DModel(**cfg).eval()
        fill_weights(cm, seed=3)
        fo = FO.OracleFluxDiT(**cfg)
        fo.load_state_dict(cm.state_dict(), strict=True)
        g = torch.Generator().manual_seed(41)
        B, T = 2, 7
        lens = [2 * 4 * 6, 4 * 6 + 3 * 5]            # sample 0: 4x6 target + 4x6 control ; sample 1: 4x6 + 3x5
        S_max = max(lens)
        x = torch.zeros(B, S_max, 64); idb = torch.zeros(B, S_max, 3)
        full = torch.ones(B, T + S_max, dtype=torch.bool)
        shapes = [[(4, 6), (4, 6)], [(4, 6), (3, 5)]]
        for b in range(B):
            x[b, : lens[b]] = torch.randn(lens[b], 64, generator=g)
            ii = []
            for j, (hh, ww) in enumerate(shapes[b]):
                q = FO.prepare_latent_image_ids(hh, ww); q[:, 0] = j; ii.append(q)
            idb[b, : lens[b]] = torch.cat(ii, 0)
            full[b, T + lens[b]:] = False
        pe = torch.randn(B, T, cfg["joint_attention_dim"], generator=g); pooled = torch.randn(B, cfg["pooled_projection_dim"], generator=g)
        tt = torch.tensor([0.7109, 0.1611]); gd = torch.ones(B); txt_ids = torch.zeros(T, 3)
        xr = x.clone().requires_grad_(True)
        out = cm(hidden_states=xr, encoder_hidden_states=pe, pooled_projections=pooled, timestep=tt, img_ids=idb, txt_ids=txt_ids,
                 guidance=gd, joint_attention_kwargs={}, return_dict=False, attention_mask=full)[0]
        tgt = torch.randn(out.shape, generator=g)
        (gxr,) = torch.autograd.grad(((out - tgt) ** 2).mean(), [xr])
        xo = x.clone().requires_grad_(True)
        oo = fo(hidden_states=xo, encoder_hidden_states=pe, pooled_projections=pooled, timestep=tt, img_ids=idb, txt_ids=txt_ids,
                guidance=gd, attention_mask=full)[0]
        (gxo,) = torch.autograd.grad(((oo - tgt) ** 2).mean(), [xo])
        print("flux multi-res oracle vs reference custom model: fwd %.3e gx %.3e; padded output max %.1e" % (
            (oo - out).abs().max().item(), (gxo - gxr).abs().max().item(), out[1, lens[1]:].abs().max().item()))
        assert (oo - out).abs().max() < 1e-5 and (gxo - gxr).abs().max() < 1e-6
        save_file({"in.hidden_states": x, "in.encoder_hidden_states": pe, "in.pooled": pooled, "in.timestep": tt, "in.img_ids": idb,
                   "in.txt_ids": txt_ids, "in.attention_mask": full.to(torch.uint8), "in.target": tgt,
                   "out.sample": out.detach().contiguous(), "grad.hidden_states": gxr.contiguous(), "w.checksum": weight_checksum(cm)},
                  os.path.join(HERE, "flux_tiny_multires.safetensors"), metadata={"cfg": repr(cfg), "weights": "common.fill_weights seed 3"})
    except Exception as e:  # noqa: BLE001
        print("multi-res reference check FAILED:", repr(e))
        raise
    # ---------------- Qwen multi-resolution: reference transformer_qwen_custom.py vs oracle ----------------
    qc = importlib.import_module("qflux.models.transformer_qwen_custom")
    import contextlib, io
    cmq = qc.QwenImageTransformer2DModel(**TINY).eval()
    fill_weights(cmq, seed=1)
    oq = O.OracleQwenDiT(**TINY)
    oq.load_state_dict(cmq.state_dict(), strict=True)
    g = torch.Generator().manual_seed(43)
    B, T = 2, 6
    shapes_b = [[(1, 4, 6), (1, 4, 6)], [(1, 4, 6), (1, 3, 5)]]      # per-sample shape lists (target + control)
    lens = [sum(f * h * w for f, h, w in sh) for sh in shapes_b]     # 48, 39
    txt_lens = [6, 4]                                                # ragged text too (exercises the placement quirk)
    S_max = max(lens)
    x = torch.zeros(B, S_max, 64)
    full = torch.zeros(B, T + S_max, dtype=torch.bool)
    for b in range(B):
        x[b, : lens[b]] = torch.randn(lens[b], 64, generator=g)
        full[b, : txt_lens[b]] = True
        full[b, T: T + lens[b]] = True
    pe = torch.randn(B, T, TINY["joint_attention_dim"], generator=g)
    tt = torch.tensor([0.7109, 0.1611])
    xr = x.clone().requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):      # the reference's forward_batched prints debug lines
        outq = cmq(hidden_states=xr, encoder_hidden_states=pe, encoder_hidden_states_mask=None, timestep=tt, img_shapes=shapes_b,
                   txt_seq_lens=txt_lens, return_dict=False, attention_mask=full)[0]
    tgt = torch.randn(outq.shape, generator=g)
    (gxr,) = torch.autograd.grad(((outq - tgt) ** 2).mean(), [xr])
    xo = x.clone().requires_grad_(True)
    oo = oq(hidden_states=xo, encoder_hidden_states=pe, timestep=tt, img_shapes=shapes_b, txt_seq_lens=txt_lens, attention_mask=full)[0]
    (gxo,) = torch.autograd.grad(((oo - tgt) ** 2).mean(), [xo])
    print("qwen multi-res oracle vs reference custom model: fwd %.3e gx %.3e; padded output max %.1e" % (
        (oo - outq).abs().max().item(), (gxo - gxr).abs().max().item(), outq[1, lens[1]:].abs().max().item()))
    assert (oo - outq).abs().max() < 1e-5 and (gxo - gxr).abs().max() < 1e-6
    save_file({"in.hidden_states": x, "in.encoder_hidden_states": pe, "in.timestep": tt, "in.attention_mask": full.to(torch.uint8),
               "in.target": tgt, "in.shapes": torch.tensor([[list(t_) for t_ in sh] for sh in shapes_b]), "in.txt_lens": torch.tensor(txt_lens),
               "out.sample": outq.detach().contiguous(), "grad.hidden_states": gxr.contiguous(), "w.checksum": weight_checksum(cmq)},
              os.path.join(HERE, "qwen_tiny_multires.safetensors"), metadata={"cfg": repr(TINY), "weights": "common.fill_weights seed 1"})
    # ---------------- criteria: the reference's OWN loss classes (importable without diffusers) ----------------
    for m_ in ("qflux.losses",):
        if m_ not in sys.modules:
            pk = types.ModuleType(m_); pk.__path__ = [os.path.join(REF, "src", "qflux", "losses")]; sys.modules[m_] = pk
    mse_mod = importlib.import_module("qflux.losses.mse_loss")
    em_mod = importlib.import_module("qflux.losses.edit_mask_loss")
    am_mod = importlib.import_module("qflux.losses.attention_mask_loss")
    g = torch.Generator().manual_seed(77)
    pr = torch.randn(2, 12, 8, generator=g); tg = torch.randn(2, 12, 8, generator=g)
    em = (torch.rand(2, 12, generator=g) > 0.5).float()
    am = torch.ones(2, 12, dtype=torch.bool); am[0, 9:] = False
    pix = (torch.rand(2, 64, 96, generator=g) > 0.7).float()
    out = {"pred": pr, "target": tg, "edit_mask": em, "attention_mask": am.to(torch.uint8), "pixel_mask": pix,
           "mse": mse_mod.MseLoss()(model_pred=pr, target=tg).reshape(1),
           "mask_edit_2_1": em_mod.MaskEditLoss(2.0, 1.0)(model_pred=pr, target=tg, edit_mask=em).reshape(1),
           "mask_edit_none": em_mod.MaskEditLoss(2.0, 1.0)(model_pred=pr, target=tg, edit_mask=None).reshape(1),
           "latent_mask": em_mod.map_mask_to_latent(pix)}
    AM = am_mod.AttentionMaskMseLoss
    try:
        out["attn_mask_mse"] = AM()(model_pred=pr, target=tg, attention_mask=am, edit_mask=None).reshape(1)
    except TypeError:
        out["attn_mask_mse"] = AM(reduction="mean")(model_pred=pr, target=tg, attention_mask=am, edit_mask=None).reshape(1)
    assert abs(O.mask_edit_loss(pr, tg, em).item() - out["mask_edit_2_1"].item()) < 1e-6
    assert torch.equal(O.map_mask_to_latent(pix), out["latent_mask"])
    assert abs(FO.attention_mask_mse_loss(pr, tg, am).item() - out["attn_mask_mse"].item()) < 1e-6
    assert abs(O.mse_loss(pr, tg).item() - out["mse"].item()) < 1e-6
    save_file({k: v.contiguous().clone() for k, v in out.items()}, os.path.join(HERE, "losses.safetensors"))
    print("criteria: oracle == reference MseLoss / MaskEditLoss / AttentionMaskMseLoss / map_mask_to_latent")
    make_f1_f3_fixtures()
    make_step_caller_fixtures()
    print("wrote golden vectors to", HERE)


if __name__ == "__main__":
    if "--f1f3" in sys.argv:      # only the cache / LoRA-file pins (does not need the diffusers shim)
        from common import FLUX_TINY, TINY  # noqa: F401
        make_f1_f3_fixtures()
    elif "--step" in sys.argv:    # only the executed step-caller pins (needs the diffusers shim for the reference's DiT files)
        import_reference_qwen()
        make_step_caller_fixtures()
    else:
        main()
