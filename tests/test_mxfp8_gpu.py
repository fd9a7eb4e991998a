"""Low-precision trunk: MX-FP8 quantiser and block-scaled FP8 GEMM vs an fp8-emulating reference in plain PyTorch
(torch.float8_e4m3fn + power-of-two block scales restated from the OCP MX specification)."""
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
    assert torch.equal(s.cpu(), sr), (s.cpu().int() - sr.int()).abs().max()
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
