"""Low-precision trunk: MX-FP8 quantiser and block-scaled FP8 GEMM vs an fp8-emulating reference in plain PyTorch
(torch.float8_e4m3fn + power-of-two block scales restated from the OCP MX specification)."""
import ctypes as C_
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _ops():
    from qflux_amd import ops
    return ops


def mx_quant_ref(x):
    """OCP MX: per 32 consecutive K elements, e = floor(log2(amax)) - 8, q = RNE_e4m3(clamp(v / 2^e, +-448)); scale byte e + 127."""
    M, K = x.shape
    v = x.float().view(M, K // 32, 32)
    amax = v.abs().amax(-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-38))) - 8, torch.full_like(amax, -127.0)).clamp(-127, 127)
    q = (v * torch.pow(2.0, -e)).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(M, K).view(torch.uint8), (e + 127).to(torch.uint8).view(M, K // 32)


@pytest.mark.parametrize("M,K,scale", [(64, 128, 1.0), (300, 3072, 4.0), (2432, 3072, 0.02), (77, 12288, 30.0)])
def test_quant_mxfp8_is_bit_exact(M, K, scale):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * scale * torch.exp(torch.randn(M, 1, generator=g))).to(BF)
    x[0, :32] = 0                       # an all-zero block
    x[1, 5] = 300.0                     # a block dominated by one outlier
    if M > 2:
        x[2, 64:96] = 2.0 ** -120       # tiny values: exponent clamps at the E8M0 minimum
    q, s = ops.quant_mxfp8(x.to(DEV))
    qr, sr = mx_quant_ref(x)
    s_rm = ops.mxfp8_scales_rowmajor(s).cpu()          # the library keeps the scales tile-major [K/128, M, 4]
    assert torch.equal(s_rm, sr), (s_rm.int() - sr.int()).abs().max()
    assert torch.equal(q.cpu(), qr), ((q.cpu() != qr).sum().item(), q.numel())
    # dequantised error stays within the e4m3 half-ulp of the block's scale
    d = ops.mxfp8_dequant(q.cpu(), s.cpu())
    blk = x.float().view(M, K // 32, 32)
    err = (d.view(M, K // 32, 32) - blk).abs().amax(-1)
    assert (err <= blk.abs().amax(-1) * 2.0 ** -3 + 1e-30).all()


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 384, 512), (2432, 3072, 3072), (384, 12288, 3072)])
def test_gemm_mxfp8_matches_fp8_emulation(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(BF).to(DEV)
    b = (torch.randn(N, K, generator=g) * 0.02 * torch.exp(0.5 * torch.randn(1, K, generator=g))).to(BF).to(DEV)
    bias = torch.randn(N, generator=g).to(BF).to(DEV)
    aq, asc = ops.quant_mxfp8(a)
    bq, bsc = ops.quant_mxfp8(b)
    ref = ops.mxfp8_dequant(aq, asc).double() @ ops.mxfp8_dequant(bq, bsc).double().t() + bias.double()
    out = ops.gemm_mxfp8(aq, asc, bq, bsc, bias=bias)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    # and against the un-quantised product: the quantisation error itself (reported, loosely bounded)
    full = a.double() @ b.double().t() + bias.double()
    qerr = ((ref - full).abs().max() / full.abs().max()).item()
    print(f"mxfp8 gemm {M}x{N}x{K}: kernel vs fp8 emulation rel {err:.2e}; fp8 emulation vs bf16 operands rel {qerr:.2e}")
    assert err < 6e-3          # one bf16 rounding of the output
    assert qerr < 6e-2


def test_gemm_mxfp8_epilogues_and_lora_segment():
    """bias + bf16 mid-rounding + bf16 LoRA K-extension + the four epilogues behave as in the bf16 kernel."""
    from qflux_amd import _lib as L
    ops = _ops()
    M, N, K, R = 300, 256, 256, 64
    g = torch.Generator().manual_seed(9)
    a = torch.randn(M, K, generator=g).to(BF).to(DEV)
    b = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    a2 = (torch.randn(M, R, generator=g) * 0.1).to(BF).to(DEV)
    b2 = (torch.randn(N, R, generator=g) * 0.1).to(BF).to(DEV)
    bias = torch.randn(N, generator=g).to(BF).to(DEV)
    aux = torch.randn(M, N, generator=g).to(BF).to(DEV)
    gate = torch.randn(1, N, generator=g).to(BF).to(DEV)
    aq, asc = ops.quant_mxfp8(a)
    bq, bsc = ops.quant_mxfp8(b)
    base = (ops.mxfp8_dequant(aq, asc) @ ops.mxfp8_dequant(bq, bsc).t() + bias.float()).to(BF).float()     # base output rounded first
    y = (base + a2.float() @ b2.float().t()).to(BF).float()
    out = ops.gemm_mxfp8(aq, asc, bq, bsc, bias=bias, a2=a2, b2=b2)
    assert ((out.float() - y).abs().max() / y.abs().max()).item() < 1e-2
    out2 = torch.empty(M, N, dtype=BF, device=DEV)
    h = ops.gemm_mxfp8(aq, asc, bq, bsc, bias=bias, a2=a2, b2=b2, epi=L.EPI_GELU, out2=out2)
    assert torch.equal(h, out)
    gl = torch.nn.functional.gelu(out.float(), approximate="tanh")
    assert ((out2.float() - gl).abs().max() / gl.abs().max()).item() < 1e-2
    o = ops.gemm_mxfp8(aq, asc, bq, bsc, bias=bias, a2=a2, b2=b2, epi=L.EPI_GATE_RES, aux=aux, gate=gate)
    want = aux.float() + (gate.float() * out.float()).to(BF).float()
    assert ((o.float() - want).abs().max() / want.abs().max()).item() < 1e-2
    o = ops.gemm_mxfp8(aq, asc, bq, bsc, bias=bias, epi=L.EPI_DGELU, aux=aux)
    xg = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(xg, approximate="tanh").sum().backward()
    want = ((ops.mxfp8_dequant(aq, asc) @ ops.mxfp8_dequant(bq, bsc).t() + bias.float()).to(BF).float() * xg.grad)
    assert ((o.float() - want).abs().max() / want.abs().max()).item() < 1e-2


FULLW = dict(patch_size=2, in_channels=64, out_channels=16, attention_head_dim=128, num_attention_heads=8, joint_attention_dim=1024,
             axes_dims_rope=(16, 56, 56))


@pytest.mark.parametrize("mode", ["mxfp8", "mxfp8-fb"])
def test_mxfp8_trunk_step_matches_fp8_emulating_oracle(mode):
    """model.quantize_trunk("mxfp8"): forward GEMMs of the block linears in MX-FP8, backward in bf16 -- one LoRA training step of a
    2-block DiT of width 1024 (the narrowest width whose linears are eligible: K % 128 == 0, K >= 1024) against the oracle with
    fp8-emulated linears (oracle/mxfp8.py); and against the un-quantised bf16 oracle to show the size of the fp8 effect."""
    import os
    import sys
    from oracle import mxfp8 as QX
    from oracle import qwen_dit as O
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(num_layers=2, **FULLW)
    g = torch.Generator(device=DEV).manual_seed(21)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.03).to(p.dtype))
    hip.add_adapter(LoraConfig(r=8, lora_alpha=8), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(p.device) * 1e-2)
    gg = torch.Generator().manual_seed(4)
    side, T = 12, 40
    S_t = side * side
    emb = dict(image_latents=torch.randn(2, S_t, 64, generator=gg).half().float(), control_latents=torch.randn(2, S_t, 64, generator=gg).half().float(),
               prompt_embeds=(torch.randn(2, T, 1024, generator=gg) * 4).half().float(), prompt_embeds_mask=torch.ones(2, T, dtype=torch.int64),
               img_shapes=[[(1, side, side), (1, side, side)]] * 2)
    noise, u = torch.randn(2, S_t, 64, generator=gg), torch.tensor([0.7109, 0.1611])
    sd = {k: v.cpu() for k, v in hip.state_dict().items()}
    res = {}
    for tag in ("fp8", "bf16"):
        oracle = O.OracleQwenDiT(num_layers=2, **FULLW)
        O.add_lora(oracle, r=8, lora_alpha=8, adapter_name="default")
        oracle.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        for n, p in oracle.named_parameters():
            if "lora" not in n:
                p.data = p.data.to(BF)
        if tag == "fp8":
            nq = QX.quantize_oracle(oracle, backward=(mode == "mxfp8-fb"))
            assert nq == 2 * 12 + 1        # 12 block linears per block + txt_in (K = 1024); img_in / proj_out / modulation stay bf16
        loss_o, pred_o = O.qwen_compute_loss(oracle, emb, noise, u, BF, return_pred=True)
        loss_o.float().backward()
        res[tag] = (loss_o.item(), pred_o.detach().float(), {n: p.grad.float() for n, p in oracle.named_parameters() if "lora" in n and p.grad is not None})
    hip.quantize_trunk(mode)
    step = QwenLoraTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, u=u).item()
    plan = list(hip._plans.values())[0]
    from qflux_amd import _lib as L
    n_fp8 = 0
    for c in plan.fwd.calls:
        if c[0] is not None and c[0].__name__ == "qfx_gemm_mxfp8":
            n_fp8 += 1
        elif c[0] is not None and c[0].__name__ == "qfx_gemm_mxfp8_grouped":
            n_fp8 += c[1][1]
    names = [c[0].__name__ for c in plan.fwd.calls if c[0] is not None]
    assert n_fp8 == 2 * 12 + 1 - 3 and "qfx_quant_mxfp8" in names     # last block: text out-proj + text MLP are dead compute
    bwd_fp8 = sum(1 for c in plan.bwd.calls if c[0] is not None and "gemm_mxfp8" in c[0].__name__)
    assert (bwd_fp8 > 0) == (mode == "mxfp8-fb")
    pred_h = plan.A["out"].view(2, -1, 64)[:, :S_t].float().cpu()
    hg = {n: p.grad.float().cpu() for n, p in hip.named_parameters() if "lora" in n}

    def rel(a, b):
        return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    e8, e16 = rel(pred_h, res["fp8"][1]), rel(pred_h, res["bf16"][1])
    gap = rel(res["fp8"][1], res["bf16"][1])
    gw = max(rel(hg[n], gr) for n, gr in res["fp8"][2].items())
    print(f"mxfp8 trunk: loss hip {loss_h:.5f} oracle-fp8 {res['fp8'][0]:.5f} oracle-bf16 {res['bf16'][0]:.5f}; pred rel hip~fp8 {e8:.4f} "
          f"hip~bf16 {e16:.4f} fp8~bf16 {gap:.4f}; worst LoRA grad rel vs fp8 oracle {gw:.4f}")
    assert abs(loss_h - res["fp8"][0]) / abs(res["fp8"][0]) < 1e-2
    assert e8 < 3e-2 and gw < 6e-2
    assert e8 < gap      # the HIP path sits closer to the fp8-emulating oracle than the bf16 one: the quantisation is really applied
    # switching back restores the bf16 trunk bit-exactly
    hip.quantize_trunk(None)
    step.zero_grad()
    l2 = step.forward_backward(emb, noise=noise, u=u).item()
    assert abs(l2 - res["bf16"][0]) / abs(res["bf16"][0]) < 5e-3


def test_gemm_mxfp8_grouped_persistent_all_epilogues():
    """The warp-specialised persistent kernel with MX-FP8 operands: grouped image + text problems, every epilogue, LoRA K-extension,
    C row map -- against the 128x128 reference kernel of the same library on the same operands (and the fp8 emulation)."""
    import ctypes as C
    from qflux_amd import _lib as L
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    Mi, Mt, N, K, R = 2048, 384, 3072, 1024, 64
    out = {}
    for epi in (L.EPI_NONE, L.EPI_GELU, L.EPI_GATE_RES, L.EPI_DGELU):
        res = []
        for persistent in (True, False):
            fs, keep, outs = [], [], []
            for M in (Mi, Mt):
                gg = torch.Generator().manual_seed(M + epi)
                a = torch.randn(M, K, generator=gg).to(BF).to(DEV)
                b = (torch.randn(N, K, generator=gg) * 0.03).to(BF).to(DEV)
                a2 = (torch.randn(M, R, generator=gg) * 0.1).to(BF).to(DEV)
                b2 = (torch.randn(N, R, generator=gg) * 0.1).to(BF).to(DEV)
                bias = torch.randn(N, generator=gg).to(BF).to(DEV)
                aux = torch.randn(M, N, generator=gg).to(BF).to(DEV)
                gate = torch.randn(1, N, generator=gg).to(BF).to(DEV)
                aq, asc = ops.quant_mxfp8(a)
                bq, bsc = ops.quant_mxfp8(b)
                y = torch.zeros(M, N, dtype=BF, device=DEV)
                y2 = torch.zeros(M, N, dtype=BF, device=DEV)
                f = L.GemmFp8Args()
                q = f.g
                q.A1, q.B1, q.lda1, q.ldb1, q.K1 = aq.data_ptr(), bq.data_ptr(), K, K, K
                q.A2, q.B2, q.lda2, q.ldb2, q.K2 = a2.data_ptr(), b2.data_ptr(), R, R, R
                q.M, q.N, q.bias, q.C, q.ldc = M, N, bias.data_ptr(), y.data_ptr(), N
                q.rows_per_batch, q.epi = M, epi
                if epi == L.EPI_GELU:
                    q.C2, q.ldc2 = y2.data_ptr(), N
                if epi in (L.EPI_GATE_RES, L.EPI_DGELU):
                    q.aux, q.ldaux = aux.data_ptr(), N
                if epi == L.EPI_GATE_RES:
                    q.gate, q.gate_bstride = gate.data_ptr(), N
                f.sa, f.sb = asc.data_ptr(), bsc.data_ptr()
                fs.append(f); keep.append((a, b, a2, b2, bias, aux, gate, aq, asc, bq, bsc)); outs.append((y, y2))
            if persistent:
                arr = (L.GemmFp8Args * 2)(*fs)
                L.check(L.lib.qfx_gemm_mxfp8_grouped(arr, 2, ops.stream_ptr()), "grouped")
            else:
                # force the 128x128 kernel: problems below the persistent threshold are routed there; split the rows to stay below it
                for f, (y, y2) in zip(fs, outs):
                    f.g.seg2_plain = 0
                    rows = f.g.M
                    for r0 in range(0, rows, 256):
                        sub = L.GemmFp8Args()
                        C.memmove(C.byref(sub), C.byref(f), C.sizeof(L.GemmFp8Args))
                        n = min(256, rows - r0)
                        sub.g.M, sub.g.rows_per_batch = n, n
                        sub.g.A1 = f.g.A1 + r0 * K
                        sub.g.A2 = f.g.A2 + r0 * R * 2
                        sub.g.C = f.g.C + r0 * N * 2
                        if f.g.C2:
                            sub.g.C2 = f.g.C2 + r0 * N * 2
                        if f.g.aux:
                            sub.g.aux = f.g.aux + r0 * N * 2
                        # tile-major scales cannot be sliced by rows: re-quantise the row block
                        blk = keep[fs.index(f)][0][r0:r0 + n]
                        sq = ops.quant_mxfp8(blk.contiguous())
                        sub.g.A1, sub.sa = sq[0].data_ptr(), sq[1].data_ptr()
                        L.check(L.lib.qfx_gemm_mxfp8(C.byref(sub), ops.stream_ptr()), "single")
                        torch.cuda.synchronize()
            torch.cuda.synchronize()
            res.append([(y.float().cpu(), y2.float().cpu()) for y, y2 in outs])
        for (ya, y2a), (yb, y2b) in zip(res[0], res[1]):
            e = ((ya - yb).abs().max() / yb.abs().max()).item()
            assert e < 1e-2, (epi, e)
            if epi == L.EPI_GELU:
                assert ((y2a - y2b).abs().max() / y2b.abs().max()).item() < 1e-2
        out[epi] = True
    assert len(out) == 4


def test_flux_mxfp8_trunk_step_matches_fp8_emulating_oracle():
    """FLUX double + single block of width 1024 with the MX-FP8 trunk (forward + dX GEMMs): the single block's proj_out runs as one
    MX-FP8 contraction over the kept [attn | gelu(mlp)] buffer; its backward as three -- d(attn) and d(mlp) against row ranges of
    proj_out^T, d(norm_x) over the concatenated K of q/k/v/proj_mlp with the adapters as the bf16 K-extension."""
    import os
    from oracle import flux_dit as FO
    from oracle import mxfp8 as QX
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg = dict(patch_size=1, in_channels=64, out_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=8,
               joint_attention_dim=1024, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    g = torch.Generator(device=DEV).manual_seed(5)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.03).to(p.dtype))
    hip.add_adapter(LoraConfig(r=8, lora_alpha=8), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(p.device) * 1e-2)
    oracle = FO.OracleFluxDiT(**cfg)
    O.add_lora(oracle, r=8, lora_alpha=8, adapter_name="default")
    oracle.load_state_dict({k: v.float().cpu() for k, v in hip.state_dict().items()}, strict=True)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    gg = torch.Generator().manual_seed(3)
    B, h, w, T = 2, 10, 12, 24
    S_t = h * w
    ctl = FO.prepare_latent_image_ids(h, w)
    ctl[:, 0] = 1
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=gg).half(), control_latents=torch.randn(B, S_t, 64, generator=gg).half(),
               control_ids=ctl, text_ids=torch.zeros(T, 3), latent_hw=(h, w), pooled_prompt_embeds=torch.randn(B, 64, generator=gg).half(),
               prompt_embeds=torch.randn(B, T, 1024, generator=gg).half())
    noise = torch.randn(B, S_t, 64, generator=gg).to(BF)
    t = torch.tensor([0.7109, 0.1611]).to(BF)
    emb_o = dict(emb, control_latents=emb["control_latents"].to(BF))
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
    with torch.no_grad():
        _, pred_b = FO.flux_compute_loss(oracle, emb_o, noise, t, BF, return_pred=True)          # bf16 oracle
    step = FluxKontextTrainStep(hip)
    step.forward_backward(emb, noise=noise, t=t)
    plan = list(hip._plans.values())[0]
    e_bf = rel(plan.A["out"].view(B, -1, 64)[:, :S_t].float().cpu(), pred_b)                     # bf16 HIP path vs bf16 oracle
    step.zero_grad()
    nq = QX.quantize_oracle(oracle, backward=True)
    assert nq == 12 + 1 + 5      # double block linears, context_embedder, single-block q/k/v + proj_mlp + proj_out (K = 5D, one contraction)
    loss_o, pred_o = FO.flux_compute_loss(oracle, emb_o, noise, t, BF, return_pred=True)
    loss_o.float().backward()
    gap = rel(pred_o, pred_b)
    hip.quantize_trunk("mxfp8-fb")
    loss_h = step.forward_backward(emb, noise=noise, t=t).item()
    plan = list(hip._plans.values())[0]
    pred_h = plan.A["out"].view(B, -1, 64)[:, :S_t].float().cpu()
    e = rel(pred_h, pred_o)
    print(f"flux: bf16 hip~oracle {e_bf:.4f}, fp8~bf16 oracle gap {gap:.4f}, hip fp8 ~ bf16 oracle {rel(pred_h, pred_b):.4f}")
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    gw = max(((p.grad.float().cpu() - og[n].float()).abs().max() / (og[n].float().abs().max() + 1e-12)).item()
             for n, p in hip.named_parameters() if "lora" in n and og[n] is not None)
    n_fp8 = 0
    for c in plan.fwd.calls:
        if c[0] is not None and c[0].__name__ == "qfx_gemm_mxfp8":
            n_fp8 += 1
        elif c[0] is not None and c[0].__name__ == "qfx_gemm_mxfp8_grouped":
            n_fp8 += c[1][1]
    assert n_fp8 == nq       # the same 18 linears run on the scaled MFMA in the HIP forward
    names = [c[0].__name__ for c in plan.bwd.calls if c[0] is not None]
    dq = [i for i, n_ in enumerate(names) if n_ == "qfx_attn_bwd_dq"]          # [single block, double block]
    assert len(dq) == 2
    head, tail = names[:dq[0]], names[dq[0]:dq[1]]
    # single block: d(attn) / d(mlp) against two row ranges of proj_out^T before its attention backward (the only bf16 GEMM up to
    # there is the K = 64 output projection), then the q/k/v/proj_mlp contraction over the concatenated K, all on the scaled MFMA
    assert head.count("qfx_gemm_mxfp8") == 2 and head.count("qfx_gemm_bf16") == 1 and "qfx_gemm_grouped" not in head, head
    k = tail.index("qfx_gemm_mxfp8")
    assert tail[k - 2:k] == ["qfx_quant_mxfp8"] * 2 and "qfx_gemm_bf16" not in tail[:k] and "qfx_gemm_grouped" not in tail[:k], tail
    print(f"flux mxfp8-fb: loss {loss_h:.5f} / {loss_o.item():.5f}, pred rel {e:.4f}, worst LoRA grad rel {gw:.4f}")
    # width-1024 weights of std 0.03 make this a noisy net: the fp8 trunk moves the prediction by `gap` (8 % of its maximum, the
    # bf16 paths agree to 1 %).  The HIP path must sit well inside that distance from the fp8-emulating oracle -- element flips at
    # fp8 rounding boundaries (bf16 inputs that differ in the last bit) are the residual.
    assert e_bf < 2e-2 and abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2
    assert e < 0.6 * gap and e < rel(pred_h, pred_b) and gw < 6e-2


@pytest.mark.parametrize("epi", ["none", "gelu", "dgelu"])
def test_gemm_mxfp8_quantising_epilogue_is_bit_identical_to_a_separate_pass(epi):
    """qfx_gemm_fp8_args.cq: the output the next GEMM contracts over leaves the persistent kernel's epilogue as MX-FP8 bytes + tile-
    major scales that equal qfx_quant_mxfp8 of the bf16 output bit for bit; cq_only suppresses the bf16 copy (the buffer keeps its
    previous contents) without changing the quantised image; the small-problem kernel refuses the request."""
    from qflux_amd import _lib as L
    ops = _ops()
    E = {"none": L.EPI_NONE, "gelu": L.EPI_GELU, "dgelu": L.EPI_DGELU}[epi]
    Mi, Mt, N, K = 2048, 384, 3072, 1024
    gg = torch.Generator().manual_seed(5)

    def build(cq_only):
        fs, keep = [], []
        for M in (Mi, Mt):
            g2 = torch.Generator().manual_seed(M)
            a = torch.randn(M, K, generator=g2).to(BF).to(DEV)
            b = (torch.randn(N, K, generator=g2) * 0.03).to(BF).to(DEV)
            bias = torch.randn(N, generator=g2).to(BF).to(DEV)
            aux = torch.randn(M, N, generator=g2).to(BF).to(DEV)
            aq, asc = ops.quant_mxfp8(a)
            bq, bsc = ops.quant_mxfp8(b)
            y = torch.full((M, N), 7.0, dtype=BF, device=DEV)
            y2 = torch.full((M, N), 7.0, dtype=BF, device=DEV)
            oq = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
            osc = torch.zeros(N // 128, M, 4, dtype=torch.uint8, device=DEV)
            f = L.GemmFp8Args()
            q = f.g
            q.A1, q.B1, q.lda1, q.ldb1, q.K1 = aq.data_ptr(), bq.data_ptr(), K, K, K
            q.M, q.N, q.bias, q.C, q.ldc = M, N, bias.data_ptr(), y.data_ptr(), N
            q.rows_per_batch, q.epi = M, E
            if E == L.EPI_GELU:
                q.C2, q.ldc2 = y2.data_ptr(), N
            if E == L.EPI_DGELU:
                q.aux, q.ldaux = aux.data_ptr(), N
            f.sa, f.sb = asc.data_ptr(), bsc.data_ptr()
            f.cq, f.cs, f.ldcq, f.cq_rows, f.cq_only = oq.data_ptr(), osc.data_ptr(), N, M, int(cq_only)
            fs.append(f); keep.append((a, b, bias, aux, aq, asc, bq, bsc, y, y2, oq, osc))
        arr = (L.GemmFp8Args * 2)(*fs)
        L.check(L.lib.qfx_gemm_mxfp8_grouped(arr, 2, ops.stream_ptr()), "grouped")
        torch.cuda.synchronize()
        return keep

    full = build(False)
    only = build(True)
    for kf, ko in zip(full, only):
        y, y2, oq, osc = kf[8], kf[9], kf[10], kf[11]
        consumed = y2 if E == L.EPI_GELU else y          # the tensor the next GEMM contracts over
        rq, rs = ops.quant_mxfp8(consumed)
        assert torch.equal(oq, rq) and torch.equal(osc.reshape(-1), rs.reshape(-1))
        assert torch.equal(ko[10], oq) and torch.equal(ko[11], osc)                       # same image with cq_only
        untouched = ko[9] if E == L.EPI_GELU else ko[8]
        assert bool((untouched == 7.0).all())                                            # ... and no bf16 copy of it
        if E == L.EPI_GELU:
            assert torch.equal(ko[8], y)                                                  # h is still written
    # a problem below the persistent threshold cannot honour the request
    f = L.GemmFp8Args()
    a = torch.randn(128, K, generator=gg).to(BF).to(DEV)
    b = torch.randn(128, K, generator=gg).to(BF).to(DEV)
    aq, asc = ops.quant_mxfp8(a)
    bq, bsc = ops.quant_mxfp8(b)
    y = torch.zeros(128, 128, dtype=BF, device=DEV)
    oq = torch.zeros(128, 128, dtype=torch.uint8, device=DEV)
    osc = torch.zeros(1, 128, 4, dtype=torch.uint8, device=DEV)
    f.g.A1, f.g.B1, f.g.lda1, f.g.ldb1, f.g.K1, f.g.M, f.g.N, f.g.C, f.g.ldc, f.g.rows_per_batch = aq.data_ptr(), bq.data_ptr(), K, K, K, 128, 128, y.data_ptr(), 128, 128
    f.sa, f.sb, f.cq, f.cs, f.ldcq, f.cq_rows = asc.data_ptr(), bsc.data_ptr(), oq.data_ptr(), osc.data_ptr(), 128, 128
    import ctypes as C
    assert L.lib.qfx_gemm_mxfp8(C.byref(f), ops.stream_ptr()) == -2        # QFX_EUNSUPPORTED


def test_layernorm_kernels_emit_bit_identical_mxfp8_images():
    """qfx_ln_modulate_fwd_batch (yq), qfx_ln_down_fwd (ln.yq) and qfx_ln_modulate_bwd_batch (dygq): the MX-FP8 image written on the
    fly equals qfx_quant_mxfp8 of the bf16 output of the same launch, bytes and tile-major scales."""
    import ctypes as C
    from qflux_amd import _lib as L
    ops = _ops()
    D = 1024
    g = torch.Generator().manual_seed(11)
    for rows in (2048, 384):
        x = (torch.randn(rows, D, generator=g) * 2).to(BF).to(DEV)
        mod = (torch.randn(1, 3 * D, generator=g) * 0.3).to(BF).to(DEV)
        # forward, row-per-wave kernel
        y = torch.empty(rows, D, dtype=BF, device=DEV)
        yq = torch.zeros(rows, D, dtype=torch.uint8, device=DEV)
        ys = torch.zeros(D // 128, rows, 4, dtype=torch.uint8, device=DEV)
        a = (L.LnFwdArgs * 1)()
        a[0].x, a[0].shift, a[0].scale, a[0].mod_bstride, a[0].y = x.data_ptr(), mod[:, :D].data_ptr(), mod[:, D:2 * D].data_ptr(), 3 * D, y.data_ptr()
        a[0].rows, a[0].D, a[0].rows_per_batch, a[0].eps = rows, D, rows, 1e-6
        a[0].yq, a[0].ys, a[0].ldyq, a[0].ys_rows = yq.data_ptr(), ys.data_ptr(), D, rows
        L.check(L.lib.qfx_ln_modulate_fwd_batch(a, 1, ops.stream_ptr()), "ln fwd")
        rq, rs = ops.quant_mxfp8(y)
        assert torch.equal(yq, rq) and torch.equal(ys.reshape(-1), rs.reshape(-1))
        # forward, fused LN + down kernel (plain LayerNorm rows: W_hi = NULL)
        y2 = torch.empty(rows, D, dtype=BF, device=DEV)
        yq.zero_(); ys.zero_()
        d = (L.LnDownArgs * 1)()
        C.memmove(C.byref(d[0].ln), C.byref(a[0]), C.sizeof(L.LnFwdArgs))
        d[0].ln.y = y2.data_ptr()
        L.check(L.lib.qfx_ln_down_fwd(d, 1, ops.stream_ptr()), "ln_down")
        rq, rs = ops.quant_mxfp8(y2)
        assert torch.equal(yq, rq) and torch.equal(ys.reshape(-1), rs.reshape(-1))
        # backward: dyg = gate * dx
        dy = torch.randn(rows, D, generator=g).to(BF).to(DEV)
        dres = torch.randn(rows, D, generator=g).to(BF).to(DEV)
        dx = torch.empty(rows, D, dtype=BF, device=DEV)
        dyg = torch.empty(rows, D, dtype=BF, device=DEV)
        yq.zero_(); ys.zero_()
        b = (L.LnBwdArgs * 1)()
        b[0].dy, b[0].x, b[0].scale, b[0].mod_bstride = dy.data_ptr(), x.data_ptr(), mod[:, D:2 * D].data_ptr(), 3 * D
        b[0].dres, b[0].gate, b[0].gate_bstride, b[0].dx, b[0].dyg = dres.data_ptr(), mod[:, 2 * D:].data_ptr(), 3 * D, dx.data_ptr(), dyg.data_ptr()
        b[0].rows, b[0].D, b[0].rows_per_batch, b[0].eps = rows, D, rows, 1e-6
        b[0].dygq, b[0].dygs, b[0].lddygq, b[0].dygs_rows = yq.data_ptr(), ys.data_ptr(), D, rows
        L.check(L.lib.qfx_ln_modulate_bwd_batch(b, 1, ops.stream_ptr()), "ln bwd")
        rq, rs = ops.quant_mxfp8(dyg)
        assert torch.equal(yq, rq) and torch.equal(ys.reshape(-1), rs.reshape(-1))


def test_lora_down_emits_bit_identical_mxfp8_image_of_its_input():
    """qfx_lora_down_args.xq: the rank-r down projection also writes the MX-FP8 image of the X it reads -- whole operand (attention
    output) and column sections of a wider operand with a joint-buffer row remap (q / k / v sections of dqkv) -- equal to
    qfx_quant_mxfp8 of the same rows; U / ext are unchanged by the option."""
    from qflux_amd import _lib as L
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    D, R, T, S_i = 1024, 16, 24, 232
    S = T + S_i
    joint = torch.randn(2 * S, 3 * D, generator=g).to(BF).to(DEV)          # [B*S, 3D], image rows follow the text rows per sample
    A = torch.randn(R, D, generator=g) * 0.05
    a_hi = A.to(BF); a_lo = (A - a_hi.float()).to(BF)
    a_hi, a_lo = a_hi.to(DEV), a_lo.to(DEV)
    M = 2 * S_i
    img = torch.cat([joint[b * S + T:(b + 1) * S] for b in range(2)])        # compact image rows
    xq = torch.zeros(M, 3 * D, dtype=torch.uint8, device=DEV)
    xs = torch.zeros(3 * D // 128, M, 4, dtype=torch.uint8, device=DEV)
    outs = []
    for with_q in (False, True):
        ext = torch.zeros(M, 64, dtype=BF, device=DEV)
        for sec in range(3):
            a = L.LoraDownArgs()
            a.X, a.ldx, a.M, a.K = joint[:, sec * D:].data_ptr(), 3 * D, M, D
            a.W_hi, a.W_lo, a.ldw, a.R = a_hi.data_ptr(), a_lo.data_ptr(), D, R
            a.ext, a.ld_ext, a.group_R, a.group_stride = ext.data_ptr(), 64, R, 0
            a.rows_per_batch, a.x_batch_rows, a.x_row_off = S_i, S, T
            if with_q:
                a.xq, a.xs, a.ldxq, a.xs_rows, a.xq_kb0 = xq.data_ptr() + sec * D, xs.data_ptr(), 3 * D, M, sec * D // 32
            L.check(L.lib.qfx_lora_down(C_.byref(a), ops.stream_ptr()), "down")
        torch.cuda.synchronize()
        outs.append(ext.clone())
    assert torch.equal(outs[0], outs[1])
    rq, rs = ops.quant_mxfp8(img.contiguous())
    assert torch.equal(xq, rq) and torch.equal(xs.reshape(-1), rs.reshape(-1))
