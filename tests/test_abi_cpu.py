"""CPU: the C-ABI library loads and exports every symbol include/qfx.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    return os.path.exists(os.path.join(ROOT, "qwen-image-finetune_amd", "qflux_amd", "libqfx.so"))


def test_header_symbols_are_exported_and_bound():
    if not _built():
        import __graft_entry__ as g
        g.build()
    from qflux_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qfx.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(qfx_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert _lib.lib.qfx_abi_version() == _lib.ABI_VERSION == 7
    assert _lib.lib.qfx_build_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    """Rejected arguments return QFX_EINVAL before any launch (safe on a CPU-only host)."""
    import ctypes as C
    from qflux_amd import _lib
    g = _lib.GemmArgs()
    assert _lib.lib.qfx_gemm_bf16(C.byref(g), None) == -1
    a = _lib.AttnArgs()
    assert _lib.lib.qfx_attn_fwd(C.byref(a), None) == -1
    with pytest.raises(_lib.QfxError):
        _lib.check(-1, "x")
    # qfx_gemm_tune: exact comma-split tile names, all-or-nothing, no launch involved
    t = _lib.lib.qfx_gemm_tune
    assert t(b"1256x1280", None) == -1 and t(b"256x128,", None) == -1 and t(b"256x128,17x3", None) == -1
    assert t(b"256x128,160x192", None) == 0 and t(None, b"1,1.09,0.965") == 0 and t(None, b"1,1.09,0.965,7") == -1
    assert t(b"all", None) == 0
    # qfx_lora_head_reduce: descriptors that do not cover what the kernel writes / reads are rejected, never launched
    r = _lib.LoraHeadReduceArgs()
    r.part, r.part_hstride, r.ld_part, r.H, r.M, r.R = 0x1000, 64 * 16, 16, 4, 64, 16
    r.group_R, r.group_stride, r.rows_per_batch = 16, 0, 64
    r.Ut_hi, r.Ut_lo, r.ld_ut = 0x2000, 0x3000, 63                      # ld_ut < M
    assert _lib.lib.qfx_lora_head_reduce(C.byref(r), 1, None) == -1
    r.Ut_hi, r.Ut_lo, r.ext, r.ld_ext = 0, 0, 0x4000, 47                 # ld_ext < 3 * group_R
    assert _lib.lib.qfx_lora_head_reduce(C.byref(r), 1, None) == -1
    r.ld_ext, r.part_hstride = 48, 63 * 16                              # a head slab shorter than the rows read from it
    assert _lib.lib.qfx_lora_head_reduce(C.byref(r), 1, None) == -1
    # ABI 7 (round 6): the attention policy entry parses completely before anything changes; the split-K lever's values go through the
    # same all-or-nothing table as the tile names; the one-pass backward refuses what it cannot run, without a device
    tune = _lib.lib.qfx_attn_tune
    assert tune(b"fwd64=2") == -1 and tune(b"dq64=1p") == -1 and tune(b"fwd64=1,nope=3") == -1 and tune(b"fwd_waves=6") == -1
    assert tune(b"fwd64=1p,dq64=0,fwd_waves=8") == 0 and tune(b"fwd64=auto,dq64=auto,fwd_waves=0") == 0 and tune(None) == 0 and tune(b"") == 0
    assert t(b"splitk=2", None) == -1 and t(b"splitk_mink=100", None) == -1 and t(b"splitk_bias=3", b"1,1.09") == -1
    assert t(b"splitk_bias=3", b"1,1.09,0.965") == 0 and t(b"splitk=0", None) == 0
    ws = _lib.lib.qfx_lora_grad_ws_floats
    assert ws(512, 3072, 16) == 0 and ws(2048, 3072, 16) == 4 * 24 * 2048 and ws(2432, 3072, 48) == 5 * 24 * 3 * 2048 and ws(2048, 3072, 17) == 0
    a.B, a.S, a.S_pad, a.H, a.dh = 1, 2432, 2432, 24, 64
    nb1, nb2 = C.c_int64(-1), C.c_int64(-1)
    assert _lib.lib.qfx_attn_bwd_fused_workspace(C.byref(a), C.byref(nb1), C.byref(nb2)) == _lib.QFX_EUNSUPPORTED and (nb1.value, nb2.value) == (0, 0)
    assert _lib.lib.qfx_attn_bwd_fused(C.byref(a), None) == _lib.QFX_EUNSUPPORTED
    g2 = _lib.LoraGradArgs()
    assert _lib.lib.qfx_lora_grad(C.byref(g2), None) == -1
    g2.Vt_hi, g2.Vt_lo, g2.ldvt, g2.R, g2.r_valid, g2.group_R = 0x1000, 0x2000, 2048, 16, 16, 16
    g2.X, g2.ldx, g2.M, g2.K, g2.G, g2.g_sr, g2.g_sc, g2.rows_per_batch = 0x3000, 3072, 2048, 3072, 0x4000, 3072, 1, 2048
    g2.ws, g2.ws_count, g2.ws_floats = 0x5000, 0x6000, 4 * 24 * 2048 - 1            # one float short of qfx_lora_grad_ws_floats(2048, 3072, 16)
    assert _lib.lib.qfx_lora_grad(C.byref(g2), None) == -1


def test_map_mask_to_latent_host_helper_matches_reference_vectors():
    """Host-side plumbing of the edit-mask criterion (no GPU needed): pixel mask -> packed-latent token mask."""
    import os
    import torch
    from safetensors.torch import load_file
    from qflux_amd.trainer import map_mask_to_latent
    t = load_file(os.path.join(os.path.dirname(__file__), "golden", "losses.safetensors"))
    assert torch.equal(map_mask_to_latent(t["pixel_mask"]), t["latent_mask"])


def test_sampling_schedule_host_logic():
    """FlowMatchEulerSchedule (host side of the sampling loop): shifted sigmas are decreasing in (0,1], end with 0, the Euler step
    reproduces x + (s_next - s) v, calculate_shift interpolates linearly (custom_flowmatch_scheduler.py:20-30)."""
    import torch
    from qflux_amd.sampling import FlowMatchEulerSchedule, calculate_shift
    assert abs(calculate_shift(256) - 0.5) < 1e-12 and abs(calculate_shift(4096) - 1.15) < 1e-12
    sch = FlowMatchEulerSchedule()
    ts = sch.set_timesteps(8, 1024)
    assert ts.shape == (8,) and sch.sigmas.shape == (9,) and sch.sigmas[-1] == 0
    assert torch.all(sch.sigmas[:-1] > 0) and torch.all(sch.sigmas[:-1] <= 1) and torch.all(sch.sigmas[1:] < sch.sigmas[:-1])
    assert abs(float(sch.sigmas[0]) - 1.0) < 1e-6     # t=1 is a fixed point of the time shift
    x, v = torch.randn(2, 3, 4), torch.randn(2, 3, 4)
    assert torch.allclose(sch.step(v, 0.7, 0.4, x), x + (0.4 - 0.7) * v, atol=1e-6)


def test_lr_schedules_host_logic():
    """get_scheduler multipliers (diffusers.optimization semantics used at base_trainer.py:900-916)."""
    from qflux_amd.trainer import get_scheduler
    c = get_scheduler("constant_with_warmup", num_warmup_steps=4)
    assert [c(s) for s in (0, 2, 4, 100)] == [0.0, 0.5, 1.0, 1.0]
    lin = get_scheduler("linear", 2, 10)
    assert abs(lin(1) - 0.5) < 1e-12 and abs(lin(6) - 0.5) < 1e-12 and lin(10) == 0.0
    cos = get_scheduler("cosine", 0, 10)
    assert abs(cos(0) - 1.0) < 1e-12 and abs(cos(5) - 0.5) < 1e-12 and abs(cos(10)) < 1e-12
    assert get_scheduler("constant")(7) == 1.0


def test_optimizer_config_mapping_host_logic():
    """optimizer.class_path / init_args of the reference's YAMLs (base_trainer.py:884-909) -> fused-step keyword arguments:
    every optimizer block that occurs under /root/reference/configs is covered (values copied as data, not the files)."""
    from qflux_amd.trainer import optimizer_kwargs_from_config as f
    # configs/face_seg_config.yaml:56-59 (and 13 more YAMLs): Adam8bit -> Adam with fp32 moments, no weight decay
    kw = f("bitsandbytes.optim.Adam8bit", {"lr": 1e-4, "betas": [0.9, 0.999]})
    assert kw == {"lr": 1e-4, "betas": (0.9, 0.999), "optimizer": "adam8bit", "weight_decay": 0.0}
    kw = f("torch.optim.AdamW", {"lr": 1e-4, "weight_decay": 0.01, "betas": [0.9, 0.999], "eps": 1e-8})
    assert kw["optimizer"] == "adamw" and kw["weight_decay"] == 0.01 and kw["eps"] == 1e-8
    # configs/face_seg_flux_kontext_fp16_prodigy.yaml:41-47
    kw = f("prodigyopt.Prodigy", {"lr": 1.0, "use_bias_correction": True, "safeguard_warmup": True, "weight_decay": 0.01})
    assert kw["optimizer"] == "prodigy" and kw["optimizer_args"] == {"use_bias_correction": True, "safeguard_warmup": True} and kw["lr"] == 1.0
    with pytest.raises(NotImplementedError):
        f("torch.optim.SGD", {"lr": 0.1})
    with pytest.raises(NotImplementedError):
        f("torch.optim.AdamW", {"lr": 1e-4, "maximize": True})


def test_ctypes_structs_match_the_c_header_layout(tmp_path):
    """Compile include/qfx.h with gcc and compare sizeof / field offsets of every argument struct with the ctypes mirrors
    (an ABI drift between the header and qflux_amd/_lib.py would otherwise only show up as wrong results on the GPU)."""
    import ctypes as C
    import os
    import subprocess
    from qflux_amd import _lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = {"qfx_gemm_args": L.GemmArgs, "qfx_lora_down_args": L.LoraDownArgs, "qfx_lora_grad_args": L.LoraGradArgs,
             "qfx_lora_pack_args": L.LoraPackArgs, "qfx_attn_args": L.AttnArgs, "qfx_ln_fwd_args": L.LnFwdArgs, "qfx_ln_bwd_args": L.LnBwdArgs,
             "qfx_prodigy_args": L.ProdigyArgs, "qfx_cond_lora_args": L.CondLoraArgs, "qfx_head_lora": L.HeadLora,
             "qfx_lora_head_reduce_args": L.LoraHeadReduceArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "qfx.h"', "int main(void) {"]
    for cname, ct in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == len(pairs)
    for line in out:
        parts = line.split()
        ct = pairs[parts[0]]
        want = [C.sizeof(ct)] + [getattr(ct, f).offset for f, _ in ct._fields_]
        assert [int(v) for v in parts[1:]] == want, (parts[0], parts[1:], want)


def test_pos_embed_stand_in_matches_pinned_tables():
    """dit.pos_embed(video_fhw, txt_seq_lens, device) (qwen_image_edit_trainer.py:734) vs the oracle tables that are pinned against the
    reference's QwenEmbedRope; the holder adds no state-dict keys (the reference's tables are plain attributes)."""
    import torch
    from oracle.qwen_dit import qwen_rope_tables
    from qflux_amd.rope import QwenEmbedRope
    pe = QwenEmbedRope(theta=10000, axes_dim=[8, 28, 28], scale_rope=True)
    for shapes, lens in ((((1, 4, 6), (1, 4, 6)), [5, 3]), (((1, 6, 4), (1, 8, 8), (1, 2, 10)), [9]), (((1, 3, 5),), [1, 7])):
        vid, txt = pe([[list(s) for s in shapes]], lens, device=torch.device("cpu"))
        v0, t0 = qwen_rope_tables(shapes, max(lens), (8, 28, 28))
        assert vid.dtype == torch.complex64 and vid.shape == v0.shape and txt.shape == t0.shape
        assert torch.equal(vid, v0) and torch.equal(txt, t0)
    assert not [k for k in vars(pe) if isinstance(getattr(pe, k), torch.Tensor)]


def test_fragment_image_layout_statements():
    """The two weight layouts the ABI-6 kernels read (include/qfx.h: qfx_lora_pack_args.A_hl / Bt_hl and A_fr) as the header states
    them, element by element, on the CPU: the torch helpers the GPU tests compare qfx_lora_pack against are themselves checked
    against the formulas of the header here."""
    import torch
    from qflux_amd import _lib as L
    BF = torch.bfloat16
    Rp, dh, H = 32, 64, 3
    hi = torch.arange(Rp * H * dh, dtype=torch.float32).reshape(Rp, H * dh).to(BF)
    lo = (hi.float() * 0.5).to(BF)
    img = L.head_fragment_image(hi, lo, dh)
    assert img.numel() == 2 * Rp * H * dh
    for (j, c) in ((0, 0), (17, 5), (31, H * dh - 1), (9, 2 * dh + 37)):
        h, dd = divmod(c, dh)
        ks, db, g, r = dd // 32, (dd // 16) % 2, (dd // 4) % 4, dd % 4
        for sel, t in ((0, hi), (1, lo)):
            off = ((((h * (Rp // 16) + j // 16) * (dh // 32) + ks) * 2 + sel) * 64 + 16 * g + j % 16) * 8 + 4 * db + r
            assert img[off] == t[j, c], (j, c, sel)
    R, K = 48, 256
    a = torch.arange(R * K, dtype=torch.float32).reshape(R, K).to(BF)
    b = (a.float() + 0.25).to(BF)
    fr = L.down_fragment_image(a, b)
    for (j, k) in ((0, 0), (16, 33), (47, 255), (23, 100)):
        off = ((k // 32 * (R // 16) + j // 16) * 64 + 16 * ((k % 32) // 8) + j % 16) * 8 + k % 8
        assert fr[off] == a[j, k] and fr[R * K + off] == b[j, k], (j, k)
    # a fragment (16 rows x 32 columns) is one contiguous 512-element piece
    piece = fr[(3 * (R // 16) + 1) * 512:(3 * (R // 16) + 2) * 512].reshape(64, 8)      # k-step 3, row group 1
    assert torch.equal(piece[16 * 2 + 5], a[16 + 5, 96 + 16:96 + 24])                       # lane (g = 2, li = 5)
