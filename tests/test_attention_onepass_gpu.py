"""Round 6: the ONE-PASS attention backward qfx_attn_bwd_fused (csrc/qfx_attn_bwd1.hip: dK, dV and dQ from a single sweep over the score
tiles, dQ accumulated across the 256-key blocks of a head in a fixed order through turn counters) against
  * an fp32 autograd reference of the joint SDPA (transformer_qwenimage.py:329-337), plain mode;
  * the two-pass pair qfx_attn_bwd_dq + qfx_attn_bwd_dkv in fused mode (QK RMSNorm + RoPE backward epilogues, rank-r projections of the
    q / k / v adapters): same arithmetic, only the fp32 summation order of dQ differs;
and run to run (bit-identical outputs, counters back at zero).  Shapes: the headline (S = 2432, 24 heads: one round of 240 blocks), two
rounds (batch 2), ragged tails on both the key and the query side, additive / -inf key masks, a single key block, cfg #4 (S = 8576: 34 key
blocks per head, 4 rounds of 7 heads)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _ops():
    from qflux_amd import ops
    return ops


def _rel(x, y):
    return ((x.float() - y.float()).abs().max() / (y.float().abs().max() + 1e-12)).item()


def _setup(S, H, Bn, mask, R, fused, seed=0):
    from qflux_amd import _lib as L
    ops = _ops()
    dh = 128
    D = H * dh
    S_pad = (S + 63) // 64 * 64
    T = 48 if S > 64 else 16
    g = torch.Generator(device=DEV).manual_seed(1000 * seed + S + 3 * H)
    qkv = torch.randn(Bn, S, 3 * D, device=DEV, generator=g).to(BF)
    dO = torch.randn(Bn, S, D, device=DEV, generator=g).to(BF)
    ld = 3 * D
    O = torch.zeros(Bn, S, D, dtype=BF, device=DEV)
    lse2 = torch.zeros(Bn, H, S_pad, device=DEV)
    dsum = torch.zeros(Bn, H, S_pad, device=DEV)
    dqkv = torch.zeros_like(qkv)
    a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                      O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:],
                      lddq=ld, lddk=ld, lddv=ld)
    keep = [qkv, dO, O, lse2, dsum, dqkv]
    kmask = None
    if mask:
        kmask = torch.zeros(Bn, S, device=DEV)
        kmask[:, S - S // 5:] = -1e4 if mask == 1 else float("-inf")
        a.key_mask = kmask.data_ptr()
        keep.append(kmask)
    a.T = T
    L.check(L.lib.qfx_attn_fwd(C.byref(a), ops.stream_ptr()), "qfx_attn_fwd")
    parts = []
    if fused:
        sqk = torch.randn(Bn, S, 2 * D, device=DEV, generator=g).to(BF)
        ang = torch.rand(S, dh // 2, device=DEV, generator=g) * 6.28
        rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
        ws = [(1 + 0.1 * torch.randn(dh, device=DEV, generator=g)).to(BF) for _ in range(4)]
        a.qk_saved, a.ld_saved, a.rope, a.rope_bstride = sqk.data_ptr(), 2 * D, rope.data_ptr(), 0
        a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = (t.data_ptr() for t in ws)
        a.norm_flags, a.norm_eps = 0, 1e-6
        keep += [sqk, rope, ws]
        if R:
            for slot in (1, 2, 3):
                wts = [(torch.randn(R, D, device=DEV, generator=g) * 0.1).to(BF) for _ in range(2)]
                part = torch.zeros(H, Bn * S, R, device=DEV)
                hl = a.hl[slot]
                hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = part.data_ptr(), Bn * S * R, R, 0, R
                w = [L.head_fragment_image(wts[0], wts[1], dh), L.head_fragment_image(wts[1], wts[0], dh)]
                hl.w_pk[0], hl.w_pk[1] = (t.data_ptr() for t in w)
                parts.append(part)
                keep.append(w)
    return a, dict(qkv=qkv, dO=dO, dqkv=dqkv, dsum=dsum, parts=parts, kmask=kmask, D=D, dh=dh, S_pad=S_pad), keep


def _run_pair(a, t, which):
    from qflux_amd import _lib as L
    ops = _ops()
    t["dqkv"].zero_(); t["dsum"].zero_()
    for p in t["parts"]:
        p.zero_()
    if which == "fused":
        L.check(L.lib.qfx_attn_bwd_fused(C.byref(a), ops.stream_ptr()), "qfx_attn_bwd_fused")
    else:
        L.check(L.lib.qfx_attn_bwd_dq(C.byref(a), ops.stream_ptr()), "qfx_attn_bwd_dq")
        L.check(L.lib.qfx_attn_bwd_dkv(C.byref(a), ops.stream_ptr()), "qfx_attn_bwd_dkv")
    torch.cuda.synchronize()
    return t["dqkv"].float().clone(), t["dsum"].clone(), [p.clone() for p in t["parts"]]


CASES = [(2432, 24, 1, 0, 16), (2432, 24, 2, 0, 16), (333, 2, 2, 0, 16), (333, 2, 2, 1, 0), (200, 3, 2, 2, 32), (64, 1, 1, 0, 0), (257, 2, 1, 1, 16),
         (1000, 4, 1, 0, 16), (700, 5, 3, 2, 16), (8576, 24, 1, 0, 16)]


@pytest.mark.parametrize("S,H,Bn,mask,R", CASES)
def test_onepass_backward_matches_autograd_and_the_two_pass_kernels(S, H, Bn, mask, R):
    ops = _ops()
    # ---- plain mode: against fp32 autograd of SDPA, and not worse than the two-pass pair
    a, t, keep = _setup(S, H, Bn, mask, 0, fused=False)
    ws = ops.attn_bwd_fused_workspace(a)
    assert ws is not None, "the one-pass backward must exist for dh = 128, S >= 64"
    two = _run_pair(a, t, "two")
    one = _run_pair(a, t, "fused")
    assert int(ws[1].abs().max()) == 0                       # every turn counter is back at zero
    D, dh = t["D"], t["dh"]
    if S <= 2432:
        q, k, v = (t["qkv"][:, :, i * D:(i + 1) * D].float().view(Bn, S, H, dh).transpose(1, 2).detach().requires_grad_(True) for i in range(3))
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        if t["kmask"] is not None:
            sc = sc + t["kmask"][:, None, None, :]
        o = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(Bn, S, D)
        o.backward(t["dO"].float())
        ref = torch.cat([x.grad.transpose(1, 2).reshape(Bn, S, D) for x in (q, k, v)], -1)
        for i, nm in enumerate(("dq", "dk", "dv")):
            e1, e2 = _rel(one[0][:, :, i * D:(i + 1) * D], ref[:, :, i * D:(i + 1) * D]), _rel(two[0][:, :, i * D:(i + 1) * D], ref[:, :, i * D:(i + 1) * D])
            assert e1 < 8e-3 and e1 <= 1.25 * e2 + 1e-4, (nm, e1, e2)
    assert torch.isfinite(one[0]).all()
    assert _rel(one[1][:, :, :S], two[1][:, :, :S]) < 1e-5                  # dsum (fp32; prep kernel vs the dQ kernel's own)
    # dK, dV: the same MFMA chains as the two-pass dK/dV kernel
    assert _rel(one[0][:, :, D:], two[0][:, :, D:]) < 6e-3
    assert _rel(one[0][:, :, :D], two[0][:, :, :D]) < 6e-3                  # dQ: fp32 sum over key blocks, then ONE bf16 rounding
    # ---- fused mode (QK-norm + RoPE backward epilogues, rank-r projections) against the two-pass pair
    a, t, keep = _setup(S, H, Bn, mask, R, fused=True, seed=1)
    ws = ops.attn_bwd_fused_workspace(a)
    two = _run_pair(a, t, "two")
    one = _run_pair(a, t, "fused")
    assert torch.isfinite(one[0]).all()
    assert _rel(one[0], two[0]) < 6e-3
    for p1, p2 in zip(one[2], two[2]):
        assert _rel(p1, p2) < 3e-3
    assert int(ws[1].abs().max()) == 0


@pytest.mark.parametrize("S,H,Bn", [(2432, 24, 1), (2432, 24, 2), (1000, 4, 1), (8576, 24, 1)])
def test_onepass_backward_is_bit_reproducible(S, H, Bn):
    """The cross-CU dQ accumulation is ORDERED (turn counters), not atomic: 6 launches on identical inputs into re-zeroed outputs give
    bit-identical dQ / dK / dV / rank-r sums, also when the accumulator workspace starts from different garbage."""
    ops = _ops()
    a, t, keep = _setup(S, H, Bn, 0, 16, fused=True, seed=2)
    ws = ops.attn_bwd_fused_workspace(a)
    ref = None
    for rep in range(6):
        ws[0].fill_(float(rep) * 1e30)          # the first key block at every tile overwrites: stale contents must not matter
        cur = _run_pair(a, t, "fused")
        flat = [cur[0], cur[1]] + cur[2]
        if ref is None:
            ref = flat
        else:
            for i, (x, y) in enumerate(zip(flat, ref)):
                assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), f"output {i} differs between launches (rep {rep})"
        assert int(ws[1].abs().max()) == 0


def test_onepass_backward_argument_checks():
    from qflux_amd import _lib as L
    ops = _ops()
    a, t, keep = _setup(333, 2, 1, 0, 0, fused=False)
    assert L.lib.qfx_attn_bwd_fused(C.byref(a), ops.stream_ptr()) == L.QFX_EINVAL          # no workspace
    a.dh = 64
    nb1, nb2 = C.c_int64(-1), C.c_int64(-1)
    assert L.lib.qfx_attn_bwd_fused_workspace(C.byref(a), C.byref(nb1), C.byref(nb2)) == L.QFX_EUNSUPPORTED and nb1.value == 0 and nb2.value == 0
    assert L.lib.qfx_attn_tune(b"fwd64=2") == L.QFX_EINVAL and L.lib.qfx_attn_tune(b"nope=1") == L.QFX_EINVAL
    assert L.lib.qfx_attn_tune(b"fwd64=1p,dq64=0,fwd_waves=8") == 0 and L.lib.qfx_attn_tune(None) == 0
