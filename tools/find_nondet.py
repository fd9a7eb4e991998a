#!/usr/bin/env python
"""Locate the first launch of the backward program whose effect differs between two identical passes: after EVERY call (all on one
stream, synchronised) every arena tensor + the flat LoRA gradient is check-summed; pass 2 repeats pass 1 from the same forward.
    python tools/find_nondet.py [--blocks 2] [--res 1024|512]"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--blocks", type=int, default=2); ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--res", type=int, default=1024, help="1024: cfg #4 (S = 8576); 512: the headline shape (S = 2432)")
args = ap.parse_args()
import test_fullsize_cfgs_gpu as T
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
from qflux_amd import ops
hip = T._qwen_full(args.blocks, None)
emb, noise, u = T._emb_1024()
if args.res == 512:
    g_ = torch.Generator().manual_seed(5)
    emb = dict(image_latents=torch.randn(1, 1024, 64, generator=g_).half(), control_latents=torch.randn(1, 1024, 64, generator=g_).half(),
               prompt_embeds=(torch.randn(1, 384, 3584, generator=g_) * 4).half(), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
    noise = torch.randn(1, 1024, 64, generator=g_); u = torch.tensor([0.37])
hip.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    for n, p in hip.named_parameters():
        if "lora_B" in n:
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(9)).to(p.device) * 1e-2)
step = QwenLoraTrainStep(hip)
st = hip.lora_store
step.forward_backward(emb, noise=noise, u=u); step.zero_grad()
plan = list(hip._plans.values())[0]

def flat(prefix, obj, out):
    if isinstance(obj, torch.Tensor): out.append((prefix, obj))
    elif isinstance(obj, dict):
        for k, v in obj.items(): flat(f"{prefix}.{k}", v, out)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj): flat(f"{prefix}[{i}]", v, out)
tens = []
flat("A", plan.A, tens)
seen = set(); uniq = []
for n, t in tens:
    key = (t.data_ptr(), t.numel(), t.dtype)
    if key in seen or t.numel() == 0: continue
    seen.add(key); uniq.append((n, t))
uniq.append(("gflat", st.gflat))
print(len(uniq), "tensors,", sum(t.numel() * t.element_size() for _, t in uniq) / 1e9, "GB")

def sums():
    out = []
    for n, t in uniq:
        v = t.contiguous().view(torch.uint8) if not t.is_contiguous() else t.view(torch.uint8)
        v = v.view(-1)
        k = v.numel() // 8 * 8
        out.append(int(v[:k].view(torch.int64).sum().item()) if k else 0)
    return out

stream = torch.cuda.current_stream().cuda_stream
FWD_TRACES = []
def one_pass():
    for n, t in uniq:                  # identical leftovers: every arena tensor starts a pass zeroed
        t.zero_()
    packed, target, pe, t_in, S_t = step._prepare(emb, noise=noise, u=u) if hasattr(step, "_prepare") else (None,) * 5
    A = plan.A
    plan._copy_rows(A["in_img"].view(plan.B, plan.S_i, -1), packed)
    A["in_txt"].view(plan.B, plan.T, -1).copy_(pe); A["t"].copy_(t_in.reshape(plan.B).float())
    hip.refresh_lora_operands()
    torch.cuda.synchronize()
    ftrace = [sums()]
    for ent in plan.fwd.calls:
        fn, a = ent[0], ent[1]
        if fn is None: a()
        else: assert fn(*a, stream) == 0, fn.__name__
        torch.cuda.synchronize()
        ftrace.append(sums())
    global FWD_TRACES
    FWD_TRACES.append(ftrace)
    pred = A["out"].view(plan.B, plan.S_i, -1)
    loss, dpred = ops.mse_loss_fwd_bwd(pred, target, S_t)
    plan._copy_rows(plan.A["dpred"].view(plan.B, plan.S_i, -1), dpred)
    hip._lora.ensure_grads(); st.gflat.zero_()
    torch.cuda.synchronize()
    trace = [sums()]
    for ent in plan.bwd.calls:
        fn, a = ent[0], ent[1]
        if fn is None:
            a()
        else:
            rc = fn(*a, stream); assert rc == 0, fn.__name__
        torch.cuda.synchronize()
        trace.append(sums())
    return trace
try:
    ref = one_pass()
except TypeError as e:
    print("prepare signature:", e); raise
for ps in range(1, args.passes):
    cur = one_pass()
    prev, ref = ref, cur          # compare consecutive serialised passes
    first = None; bad = []
    for i in range(1, len(cur)):
        # tensors this call wrote in either pass (checksum moved) ...
        wrote = [j for j in range(len(uniq)) if cur[i][j] != cur[i - 1][j] or prev[i][j] != prev[i - 1][j]]
        # ... must come out the same (every earlier write did, so its inputs are the same)
        bad = [uniq[j][0] for j in wrote if cur[i][j] != prev[i][j] and uniq[j][0] != "gflat"]   # (the weight-gradient launches add with fp32 atomics: order-dependent in the last bit)
        if bad:
            first = i; break
    if first is None:
        print(f"pass {ps}: identical"); continue
    ent = plan.bwd.calls[first - 1]
    name = "py" if ent[0] is None else ent[0].__name__ + ("@side" if len(ent) > 2 else "")
    print(f"pass {ps}: first difference after call #{first - 1} = {name}; tensors: {bad[:8]}")

# ---- forward program: consecutive passes
for ps in range(1, len(FWD_TRACES)):
    a_, b_ = FWD_TRACES[ps - 1], FWD_TRACES[ps]
    fd = next((i for i in range(1, len(a_)) if a_[i] != b_[i]), None)
    if fd is None: print(f"forward pass {ps}: identical")
    else:
        ent = plan.fwd.calls[fd - 1]
        print(f"forward pass {ps}: first difference after call #{fd - 1} = {'py' if ent[0] is None else ent[0].__name__}; tensors:", [uniq[j][0] for j in range(len(uniq)) if a_[fd][j] != b_[fd][j]][:6])
# ---- where inside the offending tensors do two passes differ?
if first is not None and bad:
    names = dict(uniq)
    def run_until(k):
        for n, t in uniq: t.zero_()
        packed, target, pe, t_in, S_t = step._prepare(emb, noise=noise, u=u)
        pred = plan.run_forward(packed, pe, t_in)
        loss, dpred = ops.mse_loss_fwd_bwd(pred, target, S_t)
        plan._copy_rows(plan.A["dpred"].view(plan.B, plan.S_i, -1), dpred)
        hip._lora.ensure_grads(); st.gflat.zero_()
        for ent in plan.bwd.calls[:k]:
            fn, a = ent[0], ent[1]
            if fn is None: a()
            else: assert fn(*a, stream) == 0
        torch.cuda.synchronize()
        return {n: names[n].clone() for n in bad}
    x, y = run_until(first), run_until(first)
    for n in bad:
        a_, b_ = x[n], y[n]
        d = (a_ != b_)
        print(n, tuple(a_.shape), "differing elements:", int(d.sum()))
        if a_.dim() >= 2:
            d2 = d.reshape(-1, a_.shape[-1])
            rows = d2.any(1).nonzero().flatten()
            cols = d2.any(0).nonzero().flatten()
            print("  rows:", rows[:24].tolist(), "... n =", rows.numel(), " per-row count (first):", d2[rows[:8]].sum(1).tolist())
            print("  cols: n =", cols.numel(), cols[:32].tolist())
            r0 = rows[0].item()
            cc = d2[r0].nonzero().flatten()[:8]
            print("  row", r0, "cols", cc.tolist(), "A", a_.reshape(-1, a_.shape[-1])[r0, cc].float().tolist(), "B", b_.reshape(-1, a_.shape[-1])[r0, cc].float().tolist())
