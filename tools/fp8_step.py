import os, sys, time, torch, json
sys.path.insert(0, "qwen-image-finetune_amd")
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0); torch.manual_seed(1234)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=60)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=1e-4)
emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, 384, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
def t(n=10):
    for _ in range(3): step.train_step(emb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = step.train_step(emb)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, l.item()
out = {}
modes = sys.argv[1:] or ["none", "mxfp8"]
# mode[:unfused] -- "mxfp8-fb:unfused" switches the quantising GEMM epilogues off (stand-alone quantisation passes for every operand)
p0 = dit.lora_store.pflat.detach().clone()
for mode_ in modes:
    mode, _, opt = mode_.partition(":")
    os.environ["QFX_FP8_FUSED_QUANT"] = "0" if opt == "unfused" else "1"
    dit.quantize_trunk(None if mode == "none" else mode)
    # every mode starts from the same adapter weights / optimizer state and sees the same noise: losses are comparable
    dit.lora_store.pflat.copy_(p0); step._m = None; step.global_step = 0
    torch.manual_seed(7)
    ms, l = t()
    mode = mode_
    out[mode] = {"ms_per_step": round(ms, 2), "images_per_s": round(1e3 / ms, 2), "loss": round(l, 4)}
    print(mode, out[mode], flush=True)
json.dump(out, open("gpurun_out/fp8_step.json", "w"), indent=1)
