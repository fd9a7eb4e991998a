"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/qfx.h).
These are what the per-kernel parity tests call; the model builds cached argument structs instead
(models/transformer_qwenimage.py) so that a training step is a flat list of C calls."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L

lib = L.lib
BF = torch.bfloat16


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _bf(t, name):
    assert t.dtype == BF and t.is_cuda, f"{name}: bf16 CUDA tensor required"
    return t


def gemm(a1, b1, *, bias=None, a2=None, b2=None, out=None, epi=L.EPI_NONE, out2=None, aux=None, gate=None,
         rows_per_batch=None, a_map=(0, 0), c_map=(0, 0), M=None, c_rows=None, row_mask=None):
    """C = a1 @ b1.T (+ a2 @ b2.T) + bias -> epilogue.  a1 [*,K1] (row stride = a1.stride(0)), b1 [N,K1]."""
    _bf(a1, "a1"); _bf(b1, "b1")
    M = a1.shape[0] if M is None else M
    N, K1 = b1.shape
    if out is None:
        out = torch.empty(M if c_rows is None else c_rows, N, dtype=BF, device=a1.device)
    g = L.GemmArgs()
    g.A1, g.B1, g.lda1, g.ldb1, g.K1 = _p(a1), _p(b1), a1.stride(0), b1.stride(0), K1
    if a2 is not None:
        g.A2, g.B2, g.lda2, g.ldb2, g.K2 = _p(a2), _p(b2), a2.stride(0), b2.stride(0), b2.shape[1]
    g.M, g.N = M, N
    g.bias = _p(bias)
    g.C, g.ldc = _p(out), out.stride(0)
    if out2 is not None:
        g.C2, g.ldc2 = _p(out2), out2.stride(0)
    if aux is not None:
        g.aux, g.ldaux = _p(aux), aux.stride(0)
    if gate is not None:
        g.gate, g.gate_bstride = _p(gate), gate.stride(0)
    g.rows_per_batch = M if rows_per_batch is None else rows_per_batch
    g.a_batch_rows, g.a_row_off = a_map
    g.c_batch_rows, g.c_row_off = c_map
    g.epi = epi
    g.row_mask = _p(row_mask)
    L.check(lib.qfx_gemm_bf16(C.byref(g), stream_ptr()), "qfx_gemm_bf16")
    return out


def gemm_grouped(arg_list):
    """arg_list: list of (a1, b1, out, kwargs) -> one grouped launch (kwargs as for gemm's struct fields)."""
    gs = []
    for (a1, b1, out, kw) in arg_list:
        g = L.GemmArgs()
        g.A1, g.B1, g.lda1, g.ldb1, g.K1 = _p(a1), _p(b1), a1.stride(0), b1.stride(0), b1.shape[1]
        g.M, g.N = kw.get("M", a1.shape[0]), b1.shape[0]
        g.bias = _p(kw.get("bias"))
        g.C, g.ldc = _p(out), out.stride(0)
        g.rows_per_batch = g.M
        g.epi = kw.get("epi", L.EPI_NONE)
        gs.append(g)
    arr = (L.GemmArgs * len(gs))(*gs)
    L.check(lib.qfx_gemm_grouped(arr, len(gs), stream_ptr()), "qfx_gemm_grouped")


def lora_down(x, w_hi, w_lo, *, U=None, ext=None, Ut=None, group_R=None, group_stride=0, M=None, rows_per_batch=None, x_map=(0, 0)):
    """Ut = (Ut_hi, Ut_lo) bf16 [R, ld] zero-initialised transposed split outputs (optional)."""
    a = L.LoraDownArgs()
    M = x.shape[0] if M is None else M
    R, K = w_hi.shape
    a.X, a.ldx, a.M, a.K = _p(x), x.stride(0), M, K
    a.W_hi, a.W_lo, a.ldw, a.R = _p(w_hi), _p(w_lo), w_hi.stride(0), R
    if U is not None:
        a.U, a.ldu = _p(U), U.stride(0)
    if ext is not None:
        a.ext, a.ld_ext = _p(ext), ext.stride(0)
    if Ut is not None:
        a.Ut_hi, a.Ut_lo, a.ld_ut = _p(Ut[0]), _p(Ut[1]), Ut[0].stride(0)
    a.group_R = R if group_R is None else group_R
    a.group_stride = group_stride
    a.rows_per_batch = M if rows_per_batch is None else rows_per_batch
    a.x_batch_rows, a.x_row_off = x_map
    L.check(lib.qfx_lora_down(C.byref(a), stream_ptr()), "qfx_lora_down")


def lora_grad(Vt, X, G, g_sr, g_sc, *, M, r_valid=None, group_R=None, K=None, rows_per_batch=None, x_map=(0, 0), out_scale=1.0, deterministic=True):
    """Vt = (Vt_hi, Vt_lo) bf16 [R, ld>=roundup(M,32)] ; G a tensor or a tuple of up to 3 tensors (fused targets)."""
    a = L.LoraGradArgs()
    Gs = G if isinstance(G, (tuple, list)) else (G,)
    R = Vt[0].shape[0]
    a.Vt_hi, a.Vt_lo, a.ldvt, a.R = _p(Vt[0]), _p(Vt[1]), Vt[0].stride(0), R
    a.group_R = R // len(Gs) if group_R is None else group_R
    a.r_valid = a.group_R if r_valid is None else r_valid
    a.X, a.ldx, a.M, a.K = _p(X), X.stride(0), M, (X.shape[1] if K is None else K)
    a.G = _p(Gs[0]); a.G1 = _p(Gs[1]) if len(Gs) > 1 else None; a.G2 = _p(Gs[2]) if len(Gs) > 2 else None
    a.g_sr, a.g_sc = g_sr, g_sc
    a.rows_per_batch = M if rows_per_batch is None else rows_per_batch
    a.x_batch_rows, a.x_row_off = x_map
    a.out_scale = out_scale
    if deterministic:      # ABI 7: chunk partials through scratch, summed in chunk order (False: the fp32 atomics of rounds 1-5)
        ws = torch.empty(max(int(lib.qfx_lora_grad_ws_floats(a.M, a.K, a.R)), 4), dtype=torch.float32, device=X.device)
        cnt = torch.zeros((a.K + 127) // 128, dtype=torch.int32, device=X.device)
        a.ws, a.ws_count, a.ws_floats = _p(ws), _p(cnt), ws.numel()
    L.check(lib.qfx_lora_grad(C.byref(a), stream_ptr()), "qfx_lora_grad")
    if deterministic:
        assert int(cnt.abs().max()) == 0      # (synchronises: this helper is for tests)


def pack_descs_tensor(descs, device):
    """list[LoraPackArgs] -> uint8 device tensor holding the C array."""
    arr = (L.LoraPackArgs * len(descs))(*descs)
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def lora_pack(desc_tensor, n, max_dim):
    L.check(lib.qfx_lora_pack(desc_tensor.data_ptr(), n, max_dim, stream_ptr()), "qfx_lora_pack")


def ln_modulate_fwd(x, shift, scale, rows_per_batch, eps=1e-6, out=None):
    rows, D = x.shape
    out = torch.empty_like(x) if out is None else out
    assert shift.stride(0) == scale.stride(0)
    L.check(lib.qfx_ln_modulate_fwd(_p(x), _p(shift), _p(scale), shift.stride(0), _p(out), rows, D, rows_per_batch, eps,
                                    stream_ptr()), "qfx_ln_modulate_fwd")
    return out


def ln_modulate_bwd(dy, x, scale, rows_per_batch, dres=None, gate=None, eps=1e-6, want_dyg=False, row_mask=None):
    rows, D = x.shape
    dx = torch.empty_like(x)
    dyg = torch.empty_like(x) if want_dyg else None
    L.check(lib.qfx_ln_modulate_bwd(_p(dy), _p(x), _p(scale), scale.stride(0), _p(dres), _p(gate),
                                    gate.stride(0) if gate is not None else 0, _p(dx), _p(dyg), rows, D, rows_per_batch, eps,
                                    _p(row_mask), stream_ptr()), "qfx_ln_modulate_bwd")
    return dx, dyg


def gate_mul(dx, gate, rows_per_batch):
    rows, D = dx.shape
    out = torch.empty_like(dx)
    L.check(lib.qfx_gate_mul(_p(dx), _p(gate), gate.stride(0), _p(out), rows, D, rows_per_batch, stream_ptr()), "qfx_gate_mul")
    return out


def rmsnorm_fwd(x, w, eps=1e-6):
    rows, D = x.shape
    out = torch.empty_like(x)
    L.check(lib.qfx_rmsnorm_fwd(_p(x), _p(w), _p(out), rows, D, eps, stream_ptr()), "qfx_rmsnorm_fwd")
    return out


def ptr_table(tensors, device):
    return torch.tensor([t.data_ptr() if t is not None else 0 for t in tensors], dtype=torch.int64, device=device)


def mod_gemv(temb, weights, biases, apply_silu=True):
    B, K = temb.shape
    N = weights[0].shape[0]
    wt = ptr_table(weights, temb.device)
    bt = ptr_table(biases, temb.device) if biases is not None else None
    out = torch.empty(len(weights), B, N, dtype=BF, device=temb.device)
    L.check(lib.qfx_mod_gemv(_p(temb), B, K, _p(wt), _p(bt), len(weights), N, int(apply_silu), _p(out), stream_ptr()), "qfx_mod_gemv")
    return out


def mod_gemv_tables(temb, wt, bt, nmat, N, apply_silu=True):
    """qfx_mod_gemv with prepared device pointer tables (wt / bt: int64 tensors of nmat device pointers)."""
    B, K = temb.shape
    out = torch.empty(nmat, B, N, dtype=BF, device=temb.device)
    L.check(lib.qfx_mod_gemv(_p(temb), B, K, _p(wt), _p(bt), nmat, N, int(apply_silu), _p(out), stream_ptr()), "qfx_mod_gemv")
    return out


def mod_gemv_t(dy, weights=None, out=None, table=None):
    """out[b, k] += sum_mat dy[mat, b, :] @ W_mat  (fp32 [B, K]); dy bf16 [nmat, B, N] contiguous; weights: list of [N, K] bf16
    tensors, or table = (prepared device pointer table, K)."""
    nmat, B, N = dy.shape
    if table is None:
        K = weights[0].shape[1]
        wt = ptr_table(weights, dy.device)
    else:
        wt, K = table
    out = torch.zeros(B, K, dtype=torch.float32, device=dy.device) if out is None else out
    L.check(lib.qfx_mod_gemv_t(_p(dy), B, N, K, _p(wt), nmat, _p(out), stream_ptr()), "qfx_mod_gemv_t")
    return out


def timestep_embed(t, dim=256, scale=1000.0, pre_scale=1.0):
    out = torch.empty(t.shape[0], dim, dtype=BF, device=t.device)
    L.check(lib.qfx_timestep_embed(_p(t.float().contiguous()), t.shape[0], dim, scale, pre_scale, _p(out), stream_ptr()),
            "qfx_timestep_embed")
    return out


def add3(a, b, c=None):
    out = torch.empty_like(a)
    L.check(lib.qfx_add3_bf16(_p(a), _p(b), _p(c), _p(out), a.numel(), stream_ptr()), "qfx_add3_bf16")
    return out


def qk_norm_rope(qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, dh, eps=1e-6, backward=False, flags=0, rope_bstride=0):
    fn = lib.qfx_qk_norm_rope_bwd if backward else lib.qfx_qk_norm_rope_fwd
    L.check(fn(_p(qkv), _p(saved), _p(rope), _p(wq_txt), _p(wk_txt), _p(wq_img), _p(wk_img), B, S, T, H, dh, eps, flags, rope_bstride, stream_ptr()),
            "qfx_qk_norm_rope")


def transpose_heads(x, ld, B, S, S_pad, H, dh, out=None):
    """x: pointer-carrying tensor whose element 0 is [b=0,s=0,h=0,d=0]; row stride ld."""
    out = torch.empty(B, H, dh, S_pad, dtype=BF, device=x.device) if out is None else out
    L.check(lib.qfx_transpose_heads(_p(x), ld, _p(out), B, S, S_pad, H, dh, stream_ptr()), "qfx_transpose_heads")
    return out


def attn_args(B, S, S_pad, H, dh, scale, **kw):
    a = L.AttnArgs()
    a.B, a.S, a.S_pad, a.H, a.dh, a.scale = B, S, S_pad, H, dh, scale
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return a


def attn_call(name, a):
    L.check(getattr(lib, name)(C.byref(a), stream_ptr()), name)


def emit_attn_backward(p, a, A):
    """Emit the attention backward of one block into launch program `p`: the two-pass pair qfx_attn_bwd_dq + qfx_attn_bwd_dkv, or the
    one-pass qfx_attn_bwd_fused (csrc/qfx_attn_bwd1.hip) with its workspace kept once per plan in the arena `A`.  QFX_ATTN_BWD =
    2pass | 1pass | auto; auto (default) = whichever measured faster for the shape: the two-pass pair everywhere as of round 6
    (profiles/r06_attn_onepass.json: the ordered fp32 dQ accumulation across key blocks costs what the saved recompute returns)."""
    import os
    mode = os.environ.get("QFX_ATTN_BWD", "auto")
    if mode == "1pass":
        if "dq_ws" not in A:
            A["dq_ws"] = attn_bwd_fused_workspace(a)
        ws = A["dq_ws"]
        if ws is not None:
            a.dq_acc, a.dq_turn = ws[0].data_ptr(), ws[1].data_ptr()
            p.c(lib.qfx_attn_bwd_fused, C.byref(a))
            return "1pass"
    if os.environ.get("QFX_ATTN_BWD_CONC", "0") == "1":
        # Round 6 lever (default OFF): the two kernels are independent once dsum = rowsum(dO * O) exists (the dQ kernel normally publishes
        # it for dK / dV), and neither fills whole rounds of the 256 CUs at S = 2432 (456 blocks of 2 rounds, 912 of 4): dsum by the
        # standalone pass, dQ on a second stream with its own copy of the statistics buffer, dK / dV on the main stream, join.
        import ctypes
        st2 = side_stream(torch.device("cuda", torch.cuda.current_device()), 0, slot=1)
        a2 = type(a)()
        ctypes.memmove(ctypes.byref(a2), ctypes.byref(a), ctypes.sizeof(a))
        if "dsum_conc" not in A:
            A["dsum_conc"] = torch.empty(a.B * a.H * a.S_pad + 1024, dtype=torch.float32, device=st2.device)
        a2.dsum = A["dsum_conc"].data_ptr()
        ev_f, ev_j = Event(), Event()
        p.keep += [a2, ev_f, ev_j, st2]
        p.c(lib.qfx_attn_bwd_prep, C.byref(a))

        def fork(ev=ev_f, s=st2):
            ev.record(torch.cuda.current_stream())
            ev.wait(s)

        def join(ev=ev_j, s=st2):
            ev.record(s)
            ev.wait(torch.cuda.current_stream())
        p.py(fork)
        p.c_on(st2, lib.qfx_attn_bwd_dq, C.byref(a2))
        p.c(lib.qfx_attn_bwd_dkv, C.byref(a))
        p.py(join)
        return "2pass-concurrent"
    p.c(lib.qfx_attn_bwd_dq, C.byref(a))
    p.c(lib.qfx_attn_bwd_dkv, C.byref(a))
    return "2pass"


def attn_tune(spec: str):
    """Kernel-selection policy of the attention entry points (qfx_attn_tune): "fwd64=0|1|1p|auto,dq64=0|1|auto,fwd_waves=0|4|8"."""
    L.check(lib.qfx_attn_tune(spec.encode() if spec else None), "qfx_attn_tune")


def attn_bwd_fused_workspace(a, device=None):
    """Allocates (once per shape, by the caller) the workspace of qfx_attn_bwd_fused for the shape in `a` and points a.dq_acc / a.dq_turn
    at it.  Returns (acc fp32 tensor, turn int32 tensor) -- keep them alive -- or None where the one-pass backward does not exist."""
    nb_acc, nb_turn = C.c_int64(0), C.c_int64(0)
    rc = lib.qfx_attn_bwd_fused_workspace(C.byref(a), C.byref(nb_acc), C.byref(nb_turn))
    if rc != 0 or nb_acc.value == 0:
        return None
    device = device or torch.device("cuda", torch.cuda.current_device())
    acc = torch.empty(nb_acc.value // 4, dtype=torch.float32, device=device)
    turn = torch.zeros(nb_turn.value // 4, dtype=torch.int32, device=device)      # zero before the first launch; launches leave it zero
    a.dq_acc, a.dq_turn = acc.data_ptr(), turn.data_ptr()
    return acc, turn


def mse_loss_fwd_bwd(pred, target, S_t, gscale=1.0, want_grad=True):
    B, S_all, Cc = pred.shape
    loss = torch.zeros((), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    L.check(lib.qfx_mse_loss_fwd_bwd(_p(pred), _p(target), _p(loss), _p(dpred), B, S_all, S_t, Cc, gscale, stream_ptr()),
            "qfx_mse_loss_fwd_bwd")
    return loss, dpred


def mse_token_weighted_fwd_bwd(pred, target, token_w, S_t, inv_denom, gscale=1.0, want_grad=True):
    B, S_all, Cc = pred.shape
    loss = torch.zeros((), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    L.check(lib.qfx_mse_token_weighted_fwd_bwd(_p(pred), _p(target), _p(token_w), _p(loss), _p(dpred), B, S_all, S_t, Cc, inv_denom,
                                               gscale, stream_ptr()), "qfx_mse_token_weighted_fwd_bwd")
    return loss, dpred


def flowmatch_prepare(x0, noise, ctrl, sigma, mode=0):
    B, S_t, Cc = x0.shape
    S_c = ctrl.shape[1] if ctrl is not None else 0
    packed = torch.empty(B, S_t + S_c, Cc, dtype=BF, device=x0.device)
    target = torch.empty(x0.shape, dtype=BF, device=x0.device)   # x0 may hold fp16 bits (mode 1); outputs are always bf16
    L.check(lib.qfx_flowmatch_prepare(_p(x0), _p(noise), _p(ctrl), _p(sigma), _p(packed), _p(target), B, S_t, S_c, Cc, mode, stream_ptr()),
            "qfx_flowmatch_prepare")
    return packed, target


def sumsq(g, out):
    L.check(lib.qfx_sumsq(_p(g), g.numel(), _p(out), stream_ptr()), "qfx_sumsq")


def sumsq_det(g, out, partials):
    """Deterministic sum of squares (fixed reduction order): out[0] is overwritten; partials = fp32 workspace."""
    L.check(lib.qfx_sumsq_det(_p(g), g.numel(), _p(out), _p(partials), partials.numel(), stream_ptr()), "qfx_sumsq_det")


_hip = None


class Event:
    """Stream-ordering event for the launch programs' fork / join points.  Default: a torch.cuda.Event.  QFX_EVENT_NOFENCE=1 (round-6
    lever): a raw HIP event created with hipEventDisableTiming | hipEventDisableSystemFence -- the record then carries no system-scope
    release (no L2 write-back for the host's sake) in the middle of the main stream; device-side ordering between streams is all these
    events are used for."""

    def __init__(self):
        global _hip
        self.raw = None
        if os.environ.get("QFX_EVENT_NOFENCE", "0") == "1":
            if _hip is None:
                _hip = C.CDLL("libamdhip64.so")
                _hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
                _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
                _hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
                _hip.hipEventDestroy.argtypes = [C.c_void_p]
            h = C.c_void_p()
            rc = _hip.hipEventCreateWithFlags(C.byref(h), 0x2 | 0x20000000)
            if rc != 0 or not h.value:
                raise L.QfxError(f"hipEventCreateWithFlags failed with code {rc}")
            self.raw = h
        else:
            self.ev = torch.cuda.Event()

    def record(self, stream):
        if self.raw is None:
            self.ev.record(stream)
        else:
            rc = _hip.hipEventRecord(self.raw, C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise L.QfxError(f"hipEventRecord failed with code {rc}")

    def wait(self, stream):
        """Make `stream` wait for this event."""
        if self.raw is None:
            stream.wait_event(self.ev)
        else:
            rc = _hip.hipStreamWaitEvent(C.c_void_p(stream.cuda_stream), self.raw, 0)
            if rc != 0:
                raise L.QfxError(f"hipStreamWaitEvent failed with code {rc}")

    def __del__(self):
        if getattr(self, "raw", None) is not None and _hip is not None:
            _hip.hipEventDestroy(self.raw)
            self.raw = None


_side_streams = {}


def side_stream(device, n_cus=16, slot=0):
    """Process-wide CU-masked side stream of `device` (qfx_stream_create_cu_masked) as a torch stream object; n_cus = 0 or a driver
    that refuses the mask -> an ordinary lowest-priority stream.  slot > 0: further streams, at the main stream's priority (peer work)."""
    key = (torch.device(device).index or 0, int(n_cus), int(slot))
    if key not in _side_streams:
        st = None
        if n_cus > 0:
            out = C.c_void_p()
            with torch.cuda.device(key[0]):
                rc = lib.qfx_stream_create_cu_masked(int(n_cus), C.byref(out))
            if rc == 0 and out.value:
                st = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", key[0]))
        if st is None:
            st = torch.cuda.Stream(device=torch.device("cuda", key[0]), priority=1 if slot == 0 else 0)
        _side_streams[key] = st
    return _side_streams[key]


def prodigy_init_state(state, d0=1e-6):
    """state: fp64[PRODIGY_STATE] device tensor (d, d_max, d_numerator, d_denom, d_hat, k, scratch...)."""
    L.check(lib.qfx_prodigy_init_state(_p(state), float(d0), stream_ptr()), "qfx_prodigy_init_state")


def prodigy_step(p, g, exp_avg, exp_avg_sq, s, p0, state, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0.0,
                 decouple=True, use_bias_correction=False, safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf"),
                 gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    a = L.ProdigyArgs(_p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), _p(s), _p(p0), p.numel(), _p(state), lr, betas[0], betas[1],
                      0.0 if beta3 is None else beta3, eps, weight_decay, d0, d_coef, growth_rate, int(use_bias_correction),
                      int(safeguard_warmup), int(decouple), _p(gnorm_sq), max_norm, grad_scale)
    L.check(lib.qfx_prodigy_step(a, stream_ptr()), "qfx_prodigy_step")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    L.check(lib.qfx_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, bc1, bc2, _p(gnorm_sq), max_norm,
                               grad_scale, stream_ptr()), "qfx_adamw_step")


# ---------------------------------------------------------------------------------------------- MX-FP8 (low-precision trunk)
def quant_mxfp8(x, M=None, rows_per_batch=None, x_map=(0, 0), out=None):
    """x [*, K] bf16 (row stride x.stride(0)) -> (q uint8 [M, K], s uint8 [K/128, M, 4] tile-major scales) in the OCP MX-FP8
    format (qfx.h)."""
    _bf(x, "x")
    M = x.shape[0] if M is None else M
    K = x.shape[1]
    q, s = out if out is not None else (torch.empty(M, K, dtype=torch.uint8, device=x.device),
                                        torch.empty(K // 128, M, 4, dtype=torch.uint8, device=x.device))
    a = L.QuantArgs()
    a.X, a.ldx, a.M, a.K = _p(x), x.stride(0), M, K
    a.Q, a.ldq, a.S, a.lds = _p(q), q.stride(0), _p(s), 0
    a.rows_per_batch = M if rows_per_batch is None else rows_per_batch
    a.x_batch_rows, a.x_row_off = x_map
    L.check(lib.qfx_quant_mxfp8(C.byref(a), stream_ptr()), "qfx_quant_mxfp8")
    return q, s


def mxfp8_scales_rowmajor(s):
    """tile-major scale bytes [K/128, M, 4] -> [M, K/32]."""
    return s.permute(1, 0, 2).reshape(s.shape[1], -1)


def mxfp8_dequant(q, s):
    """Host/torch view of an MX-FP8 operand as fp32 (test helper; not on the product path)."""
    v = q.view(torch.float8_e4m3fn).float().view(q.shape[0], -1, 32)
    sc = torch.pow(2.0, mxfp8_scales_rowmajor(s).float() - 127.0).unsqueeze(-1)
    return (v * sc).view(q.shape[0], -1)


def gemm_mxfp8(aq, asc, bq, bsc, *, bias=None, a2=None, b2=None, out=None, epi=L.EPI_NONE, out2=None, aux=None, gate=None,
               rows_per_batch=None, c_map=(0, 0), c_rows=None, row_mask=None):
    """C = dequant(aq, asc) @ dequant(bq, bsc).T (+ a2 @ b2.T in bf16) + bias -> epilogue (same contract as ops.gemm)."""
    M, K1 = aq.shape
    N = bq.shape[0]
    if out is None:
        out = torch.empty(M if c_rows is None else c_rows, N, dtype=BF, device=aq.device)
    f = L.GemmFp8Args()
    g = f.g
    g.A1, g.B1, g.lda1, g.ldb1, g.K1 = _p(aq), _p(bq), aq.stride(0), bq.stride(0), K1
    if a2 is not None:
        g.A2, g.B2, g.lda2, g.ldb2, g.K2 = _p(a2), _p(b2), a2.stride(0), b2.stride(0), b2.shape[1]
    g.M, g.N = M, N
    g.bias = _p(bias)
    g.C, g.ldc = _p(out), out.stride(0)
    if out2 is not None:
        g.C2, g.ldc2 = _p(out2), out2.stride(0)
    if aux is not None:
        g.aux, g.ldaux = _p(aux), aux.stride(0)
    if gate is not None:
        g.gate, g.gate_bstride = _p(gate), gate.stride(0)
    g.rows_per_batch = M if rows_per_batch is None else rows_per_batch
    g.c_batch_rows, g.c_row_off = c_map
    g.epi = epi
    g.row_mask = _p(row_mask)
    f.sa, f.ldsa, f.sb, f.ldsb = _p(asc), 0, _p(bsc), 0
    L.check(lib.qfx_gemm_mxfp8(C.byref(f), stream_ptr()), "qfx_gemm_mxfp8")
    return out
