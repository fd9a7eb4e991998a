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
