"""Host-side RoPE table construction (plumbing: runs once per shape signature and is cached).

Follows QwenEmbedRope (src/qflux/models/transformer_qwenimage.py:159-254): per-image frame index
= position in the sample's image list, centred h/w positions (scale_rope=True), text positions
start at max(h//2, w//2) over the images and use the same index on all three axes; only the first
sample's shape list is used for the whole batch (:206-207).  Output is the JOINT table in the
kernels' layout: [T + S_i, dh/2, 2] fp32 (cos, sin), text rows first (concat order :324-326).
"""
from __future__ import annotations

import functools

import torch


def _axis(pos: torch.Tensor, dim: int, theta: float) -> torch.Tensor:
    inv = 1.0 / torch.pow(torch.tensor(float(theta)), torch.arange(0, dim, 2).to(torch.float32).div(dim))
    ang = torch.outer(pos.to(torch.float32), inv)
    return torch.polar(torch.ones_like(ang), ang)


@functools.lru_cache(maxsize=64)
def qwen_joint_rope(img_shapes: tuple, txt_len: int, axes_dim: tuple, theta: float = 10000.0) -> torch.Tensor:
    vids = []
    max_vid_index = 0
    for idx, (frame, height, width) in enumerate(img_shapes):
        fpos = torch.arange(idx, idx + frame)
        hpos = torch.cat([torch.arange(-(height - height // 2), 0), torch.arange(0, height // 2)])
        wpos = torch.cat([torch.arange(-(width - width // 2), 0), torch.arange(0, width // 2)])
        ff = _axis(fpos, axes_dim[0], theta).view(frame, 1, 1, -1).expand(frame, height, width, -1)
        fh = _axis(hpos, axes_dim[1], theta).view(1, height, 1, -1).expand(frame, height, width, -1)
        fw = _axis(wpos, axes_dim[2], theta).view(1, 1, width, -1).expand(frame, height, width, -1)
        vids.append(torch.cat([ff, fh, fw], dim=-1).reshape(frame * height * width, -1))
        max_vid_index = max(height // 2, width // 2, max_vid_index)
    tpos = torch.arange(max_vid_index, max_vid_index + txt_len)
    txt = torch.cat([_axis(tpos, d, theta) for d in axes_dim], dim=1)
    joint = torch.cat([txt] + vids, dim=0)
    return torch.view_as_real(joint).contiguous().float()


def normalize_img_shapes(img_shapes) -> tuple:
    """Accept [[(f,h,w),...]]*B, [(f,h,w),...] or (f,h,w); return the first sample's list as a tuple of tuples."""
    s = img_shapes
    if isinstance(s, (list, tuple)) and len(s) and isinstance(s[0], (list, tuple)) and len(s[0]) and isinstance(s[0][0], (list, tuple)):
        s = s[0]
    if isinstance(s, (list, tuple)) and len(s) == 3 and all(isinstance(v, int) for v in s):
        s = [s]
    return tuple(tuple(int(v) for v in fhw) for fhw in s)


class QwenEmbedRope:
    """Call-compatible stand-in for the module the reference keeps at `dit.pos_embed` (transformer_qwenimage.py:159-254; called as
    `self.dit.pos_embed([img_shapes[b]], [txt_seq_lens[b]], device=device)` at qwen_image_edit_trainer.py:734): returns
    (vid_freqs [S_i, sum(axes)/2] complex64, txt_freqs [max(txt_seq_lens), ...] complex64).  Holds no parameters or buffers, like the
    reference's (its tables are plain attributes, so the state dict has no pos_embed keys).  Host-side table construction; the
    kernels consume the same table through qwen_joint_rope()."""

    def __init__(self, theta: int, axes_dim, scale_rope: bool = True):
        if not scale_rope:
            raise NotImplementedError("scale_rope=False is not used by the reference model (transformer_qwenimage.py:549)")
        self.theta, self.axes_dim, self.scale_rope = theta, tuple(axes_dim), scale_rope

    def __call__(self, video_fhw, txt_seq_lens, device=None):
        shapes = normalize_img_shapes(video_fhw)          # only the first sample's list is used (:206-207)
        T = int(max(txt_seq_lens))
        joint = torch.view_as_complex(qwen_joint_rope(shapes, T, self.axes_dim, float(self.theta)))
        txt, vid = joint[:T].contiguous(), joint[T:].contiguous()
        if device is not None:
            txt, vid = txt.to(device), vid.to(device)
        return vid, txt

    forward = __call__
