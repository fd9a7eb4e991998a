"""ctypes binding of libqfx.so (include/qfx.h).  The product path has NO fallback: if the HIP
library is missing or does not export a symbol this module raises at import time."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QFX_LIB_PATH") or os.path.join(_HERE, "libqfx.so")   # QFX_LIB_PATH: lab builds (tools/build_variants.py)

c_u16p = C.c_void_p
c_f32p = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("A1", C.c_void_p), ("B1", C.c_void_p), ("lda1", C.c_int64), ("ldb1", C.c_int64), ("K1", C.c_int32),
        ("A2", C.c_void_p), ("B2", C.c_void_p), ("lda2", C.c_int64), ("ldb2", C.c_int64), ("K2", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32),
        ("bias", C.c_void_p),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("C2", C.c_void_p), ("ldc2", C.c_int64),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("gate", C.c_void_p), ("gate_bstride", C.c_int64),
        ("rows_per_batch", C.c_int32),
        ("a_batch_rows", C.c_int32), ("a_row_off", C.c_int32),
        ("c_batch_rows", C.c_int32), ("c_row_off", C.c_int32),
        ("epi", C.c_int32), ("row_mask", C.c_void_p), ("aux_unmapped", C.c_int32), ("seg2_plain", C.c_int32),
    ]


class QuantArgs(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("M", C.c_int32), ("K", C.c_int32),
                ("Q", C.c_void_p), ("ldq", C.c_int64), ("S", C.c_void_p), ("lds", C.c_int64),
                ("rows_per_batch", C.c_int32), ("x_batch_rows", C.c_int32), ("x_row_off", C.c_int32)]


class GemmFp8Args(C.Structure):
    _fields_ = [("g", GemmArgs), ("sa", C.c_void_p), ("ldsa", C.c_int64), ("sb", C.c_void_p), ("ldsb", C.c_int64),
                ("cq", C.c_void_p), ("cs", C.c_void_p), ("ldcq", C.c_int64), ("cq_rows", C.c_int32), ("cq_only", C.c_int32)]


class LoraDownArgs(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64), ("M", C.c_int32), ("K", C.c_int32),
        ("W_hi", C.c_void_p), ("W_lo", C.c_void_p), ("ldw", C.c_int64), ("R", C.c_int32),
        ("U", C.c_void_p), ("ldu", C.c_int64),
        ("ext", C.c_void_p), ("ld_ext", C.c_int64),
        ("Ut_hi", C.c_void_p), ("Ut_lo", C.c_void_p), ("ld_ut", C.c_int64),
        ("group_R", C.c_int32), ("group_stride", C.c_int32),
        ("rows_per_batch", C.c_int32), ("x_batch_rows", C.c_int32), ("x_row_off", C.c_int32),
        ("xq", C.c_void_p), ("xs", C.c_void_p), ("ldxq", C.c_int64), ("xs_rows", C.c_int32), ("xq_kb0", C.c_int32),
    ]


class LoraGradArgs(C.Structure):
    _fields_ = [
        ("Vt_hi", C.c_void_p), ("Vt_lo", C.c_void_p), ("ldvt", C.c_int64), ("R", C.c_int32), ("r_valid", C.c_int32), ("group_R", C.c_int32),
        ("X", C.c_void_p), ("ldx", C.c_int64), ("M", C.c_int32), ("K", C.c_int32),
        ("G", C.c_void_p), ("G1", C.c_void_p), ("G2", C.c_void_p), ("g_sr", C.c_int64), ("g_sc", C.c_int64),
        ("rows_per_batch", C.c_int32), ("x_batch_rows", C.c_int32), ("x_row_off", C.c_int32),
        ("out_scale", C.c_float),
        ("ws", C.c_void_p), ("ws_count", C.c_void_p), ("ws_floats", C.c_int64),
    ]


class LnFwdArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("shift", C.c_void_p), ("scale", C.c_void_p), ("mod_bstride", C.c_int64), ("y", C.c_void_p),
                ("rows", C.c_int32), ("D", C.c_int32), ("rows_per_batch", C.c_int32), ("eps", C.c_float),
                ("yq", C.c_void_p), ("ys", C.c_void_p), ("ldyq", C.c_int64), ("ys_rows", C.c_int32), ("pad_", C.c_int32)]


class LnDownArgs(C.Structure):
    _fields_ = [("ln", LnFwdArgs), ("W_hi", C.c_void_p), ("W_lo", C.c_void_p), ("ldw", C.c_int64), ("R", C.c_int32),
                ("ext", C.c_void_p), ("ld_ext", C.c_int64), ("Ut_hi", C.c_void_p), ("Ut_lo", C.c_void_p), ("ld_ut", C.c_int64),
                ("group_R", C.c_int32), ("group_stride", C.c_int32), ("W_fr", C.c_void_p)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("scale", C.c_void_p), ("mod_bstride", C.c_int64),
                ("dres", C.c_void_p), ("gate", C.c_void_p), ("gate_bstride", C.c_int64), ("dx", C.c_void_p), ("dyg", C.c_void_p),
                ("row_mask", C.c_void_p), ("rows", C.c_int32), ("D", C.c_int32), ("rows_per_batch", C.c_int32), ("eps", C.c_float),
                ("dygq", C.c_void_p), ("dygs", C.c_void_p), ("lddygq", C.c_int64), ("dygs_rows", C.c_int32), ("pad_", C.c_int32)]


class ModGradArgs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("ld_dy", C.c_int64), ("x", C.c_void_p), ("ld_x", C.c_int64),
                ("dxo", C.c_void_p), ("ld_dxo", C.c_int64), ("y", C.c_void_p), ("ld_y", C.c_int64),
                ("dshift", C.c_void_p), ("dscale", C.c_void_p), ("dgate", C.c_void_p), ("out_bstride", C.c_int64),
                ("row_mask", C.c_void_p), ("rows", C.c_int32), ("D", C.c_int32), ("rows_per_batch", C.c_int32), ("eps", C.c_float)]


class LoraPackArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("r", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("scale", C.c_float),
        ("A_hi", C.c_void_p), ("A_lo", C.c_void_p), ("ld_a", C.c_int64),
        ("Bt_hi", C.c_void_p), ("Bt_lo", C.c_void_p), ("ld_bt", C.c_int64),
        ("We", C.c_void_p), ("ld_we", C.c_int64),
        ("WeT", C.c_void_p), ("ld_wet", C.c_int64),
        ("Rp", C.c_int32), ("Kext", C.c_int32),
        ("A_hl", C.c_void_p), ("Bt_hl", C.c_void_p), ("hl_dh", C.c_int32), ("reserved", C.c_int32),
        ("A_fr", C.c_void_p), ("fr_row0", C.c_int32), ("fr_nf", C.c_int32),
    ]


class CondLoraArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("B", C.c_int32), ("K", C.c_int32), ("apply_silu", C.c_int32), ("na", C.c_int32), ("r", C.c_int32),
                ("N", C.c_int32), ("A", C.c_void_p), ("Bm", C.c_void_p), ("scale", C.c_void_p), ("u", C.c_void_p),
                ("y", C.c_void_p), ("ldy", C.c_int64), ("g", C.c_void_p), ("ldg", C.c_int64), ("dA", C.c_void_p), ("dB", C.c_void_p),
                ("du", C.c_void_p), ("dx", C.c_void_p)]


class ProdigyArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("s", C.c_void_p),
                ("p0", C.c_void_p), ("n", C.c_int64), ("state", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("beta3", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("d0", C.c_float), ("d_coef", C.c_float), ("growth_rate", C.c_float),
                ("use_bias_correction", C.c_int32), ("safeguard_warmup", C.c_int32), ("decouple", C.c_int32),
                ("gnorm_sq", C.c_void_p), ("max_norm", C.c_float), ("grad_scale", C.c_float)]


PRODIGY_STATE = 12


class HeadLora(C.Structure):
    _fields_ = [("w_pk", C.c_void_p * 2),
                ("part", C.c_void_p), ("part_hstride", C.c_int64), ("ld_part", C.c_int32), ("c0", C.c_int32), ("R", C.c_int32),
                ("reserved", C.c_int32)]


class LoraHeadReduceArgs(C.Structure):
    _fields_ = [("part", C.c_void_p), ("part_hstride", C.c_int64), ("ld_part", C.c_int32), ("H", C.c_int32),
                ("M", C.c_int32), ("R", C.c_int32),
                ("ext", C.c_void_p), ("ld_ext", C.c_int64),
                ("Ut_hi", C.c_void_p), ("Ut_lo", C.c_void_p), ("ld_ut", C.c_int64),
                ("group_R", C.c_int32), ("group_stride", C.c_int32),
                ("rows_per_batch", C.c_int32), ("x_batch_rows", C.c_int32), ("x_row_off", C.c_int32), ("reserved", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64),
        ("Qt", C.c_void_p), ("Kt", C.c_void_p), ("Vt", C.c_void_p),
        ("O", C.c_void_p), ("ldo", C.c_int64),
        ("lse2", C.c_void_p), ("dsum", C.c_void_p),
        ("dO", C.c_void_p), ("lddo", C.c_int64), ("dOt", C.c_void_p),
        ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p), ("lddq", C.c_int64), ("lddk", C.c_int64), ("lddv", C.c_int64),
        ("key_mask", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("S_pad", C.c_int32), ("H", C.c_int32), ("dh", C.c_int32), ("scale", C.c_float),
        ("qk_saved", C.c_void_p), ("ld_saved", C.c_int64), ("rope", C.c_void_p), ("rope_bstride", C.c_int64),
        ("wq_txt", C.c_void_p), ("wk_txt", C.c_void_p), ("wq_img", C.c_void_p), ("wk_img", C.c_void_p),
        ("T", C.c_int32), ("norm_flags", C.c_int32), ("norm_eps", C.c_float),
        ("hl", HeadLora * 4),
        ("dq_acc", C.c_void_p), ("dq_turn", C.c_void_p),
    ]


ABI_VERSION = 7        # QFX_ABI_VERSION
QFX_OK, QFX_EINVAL, QFX_EUNSUPPORTED = 0, -1, -2
MAX_BATCH = 8          # QFX_MAX_BATCH
MAX_LN_BATCH = 4       # QFX_MAX_LN_BATCH
EPI_NONE, EPI_GELU, EPI_GATE_RES, EPI_DGELU = 0, 1, 2, 3

# name -> (restype, argtypes); every symbol include/qfx.h declares
_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "qfx_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "qfx_gemm_grouped": (C.c_int, [C.POINTER(GemmArgs), _i32, _vp]),
    "qfx_quant_mxfp8": (C.c_int, [C.POINTER(QuantArgs), _vp]),
    "qfx_gemm_mxfp8": (C.c_int, [C.POINTER(GemmFp8Args), _vp]),
    "qfx_gemm_mxfp8_grouped": (C.c_int, [C.POINTER(GemmFp8Args), C.c_int32, _vp]),
    "qfx_lora_down": (C.c_int, [C.POINTER(LoraDownArgs), _vp]),
    "qfx_lora_down_batch": (C.c_int, [C.POINTER(LoraDownArgs), C.c_int32, _vp]),
    "qfx_lora_grad": (C.c_int, [C.POINTER(LoraGradArgs), _vp]),
    "qfx_lora_grad_batch": (C.c_int, [C.POINTER(LoraGradArgs), C.c_int32, _vp]),
    "qfx_lora_grad_ws_floats": (C.c_int64, [_i32, _i32, _i32]),
    "qfx_lora_pack": (C.c_int, [_vp, _i32, _i32, _vp]),
    "qfx_ln_modulate_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _f, _vp]),
    "qfx_ln_modulate_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _f, _vp, _vp]),
    "qfx_ln_modulate_fwd_batch": (C.c_int, [C.POINTER(LnFwdArgs), C.c_int32, _vp]),
    "qfx_ln_down_fwd": (C.c_int, [C.POINTER(LnDownArgs), C.c_int32, _vp]),
    "qfx_ln_modulate_bwd_batch": (C.c_int, [C.POINTER(LnBwdArgs), C.c_int32, _vp]),
    "qfx_mod_grad": (C.c_int, [C.POINTER(ModGradArgs), _vp]),
    "qfx_mod_grad_batch": (C.c_int, [C.POINTER(ModGradArgs), C.c_int32, _vp]),
    "qfx_gate_mul": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "qfx_rmsnorm_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _vp]),
    "qfx_mod_gemv": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "qfx_mod_gemv_t": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "qfx_timestep_embed": (C.c_int, [_vp, _i32, _i32, _f, _f, _vp, _vp]),
    "qfx_cond_lora_fwd": (C.c_int, [C.POINTER(CondLoraArgs), _vp]),
    "qfx_cond_lora_bwd": (C.c_int, [C.POINTER(CondLoraArgs), _vp]),
    "qfx_cast_f32_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "qfx_silu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "qfx_add3_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "qfx_qk_norm_rope_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _i32, _i64, _vp]),
    "qfx_qk_norm_rope_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _i32, _i64, _vp]),
    "qfx_transpose_heads": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "qfx_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "qfx_attn_bwd_prep": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "qfx_attn_bwd_dq": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "qfx_attn_bwd_dkv": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "qfx_attn_bwd_fused": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "qfx_attn_bwd_fused_workspace": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "qfx_attn_tune": (C.c_int, [C.c_char_p]),
    "qfx_mse_loss_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f, _vp]),
    "qfx_mse_token_weighted_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f, _f, _vp]),
    "qfx_flowmatch_prepare": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "qfx_sumsq": (C.c_int, [_vp, _i64, _vp, _vp]),
    "qfx_sumsq_det": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp]),
    "qfx_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _vp, _f, _f, _vp]),
    "qfx_prodigy_init_state": (C.c_int, [_vp, C.c_double, _vp]),
    "qfx_prodigy_step": (C.c_int, [C.POINTER(ProdigyArgs), _vp]),
    "qfx_stream_create_cu_masked": (C.c_int, [_i32, C.POINTER(C.c_void_p)]),
    "qfx_stream_destroy": (C.c_int, [_vp]),
    "qfx_debug_where": (C.c_int, [_vp, _i32, _vp]),
    "qfx_gemm_tune": (C.c_int, [C.c_char_p, C.c_char_p]),
    "qfx_lora_head_reduce": (C.c_int, [C.POINTER(LoraHeadReduceArgs), _i32, _vp]),
    "qfx_debug_tr_read": (C.c_int, [_vp, _vp, _vp]),
    "qfx_abi_version": (C.c_int, []),
    "qfx_build_arch": (C.c_char_p, []),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP library is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "at the repo root (needs hipcc). qflux_amd has no CPU / eager fallback by design.")
    # torch FIRST: the PyTorch-ROCm wheel bundles its own libamdhip64 and libqfx.so's HIP dependency must resolve to THAT instance.
    # Loaded before torch, libqfx.so pulls in the system runtime, the process ends up with two HIP runtimes and every launch on a
    # torch stream fails with hipErrorNoDevice (seen when pytest's stale-library rebuild imported the package ahead of torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.qfx_abi_version() != ABI_VERSION:
        raise ImportError("libqfx.so ABI version mismatch")
    return lib


lib = _load()


class QfxError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise QfxError(f"{what} failed with code {rc}" + (" (HIP error %d)" % (-rc - 1000) if rc <= -1000 else ""))


def head_fragment_image(hi, lo, dh):
    """The head-fragment order of a [Rp, H*dh] bf16 hi/lo split (include/qfx.h, qfx_lora_pack_args.A_hl / Bt_hl): what
    qfx_lora_pack writes for the attention epilogues (qfx_head_lora.w_pk).  Layout statement for tests and tools -- the training
    path's images come from qfx_lora_pack."""
    import torch
    Rp, C_ = hi.shape
    assert Rp % 16 == 0 and dh % 32 == 0 and C_ % dh == 0
    j = torch.arange(Rp, device=hi.device).view(Rp, 1)
    c = torch.arange(C_, device=hi.device).view(1, C_)
    h, dd = c // dh, c % dh
    ks, db, g, r = dd // 32, (dd // 16) % 2, (dd // 4) % 4, dd % 4
    base = ((h * (Rp // 16) + j // 16) * (dh // 32) + ks) * 2
    img = torch.zeros(2 * Rp * C_, dtype=hi.dtype, device=hi.device)
    for sel, t in ((0, hi), (1, lo)):
        off = ((base + sel) * 64 + 16 * g + j % 16) * 8 + 4 * db + r
        img[off.reshape(-1)] = t.reshape(-1)
    return img


def down_fragment_image(hi, lo):
    """MFMA-fragment order of a [R, K] bf16 hi / lo split (include/qfx.h, qfx_lora_pack_args.A_fr): what qfx_lora_pack writes for
    qfx_ln_down_args.W_fr.  Layout statement for tests -- the training path's images come from qfx_lora_pack."""
    import torch
    R, K = hi.shape
    assert R % 16 == 0 and K % 32 == 0
    j = torch.arange(R, device=hi.device).view(R, 1)
    k = torch.arange(K, device=hi.device).view(1, K)
    off = (((k // 32) * (R // 16) + j // 16) * 64 + 16 * ((k % 32) // 8) + j % 16) * 8 + k % 8
    img = torch.zeros(2 * R * K, dtype=hi.dtype, device=hi.device)
    img[off.reshape(-1)] = hi.reshape(-1)
    img[R * K + off.reshape(-1)] = lo.reshape(-1)
    return img
