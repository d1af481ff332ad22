"""Synthetic bursts (after the reference's own burst
generator test_opencv/main.cpp:1877-1913: one scene, per-frame random shifts
in U(-5,5) LR pixels, down-sampling, crops).

Scene = band-limited noise + hard edges on an HR grid of (s*W+64*s) x (s*H+64*s);
frame k = scene shifted by a sub-pixel offset t_k (bilinear), box-averaged by s,
plus signal-dependent noise (variance alpha*I + beta), then either kept as a
12-bit gray frame or mosaicked to a 12-bit RGGB raw frame
(u16 = round(I * white + black)).  Frame 0 is the reference (zero shift).

torch is used only as an array library (works on cpu and on cuda); nothing here
is part of the measured path.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


def _scene(hr_h: int, hr_w: int, gen: torch.Generator, device) -> torch.Tensor:
    """[3, H, W] float in [0.05, 0.95]."""
    def noise(div):
        h, w = max(hr_h // div, 2), max(hr_w // div, 2)
        n = torch.rand(1, 1, h, w, generator=gen, device=device)
        return F.interpolate(n, size=(hr_h, hr_w), mode="bicubic", align_corners=False)[0, 0]

    lum = 0.5 * noise(32) + 0.3 * noise(8) + 0.2 * noise(3)
    # hard edges: random axis-aligned boxes and a diagonal stripe pattern
    yy, xx = torch.meshgrid(torch.arange(hr_h, device=device), torch.arange(hr_w, device=device), indexing="ij")
    nbox = 24
    bx = torch.rand(nbox, 4, generator=gen, device=device)
    for i in range(nbox):
        x0, y0 = int(bx[i, 0] * hr_w), int(bx[i, 1] * hr_h)
        w, h = int(20 + bx[i, 2] * hr_w * 0.15), int(20 + bx[i, 3] * hr_h * 0.15)
        lum[y0:y0 + h, x0:x0 + w] = lum[y0:y0 + h, x0:x0 + w] * 0.5 + (0.15 if i % 2 else 0.6)
    stripes = (((xx + 2 * yy) // 23) % 2).float()
    lum = lum * (0.85 + 0.15 * stripes)
    lum = (lum - lum.min()) / (lum.max() - lum.min() + 1e-9)
    chroma = torch.stack([noise(16), noise(16), noise(16)]) - 0.5
    rgb = (0.1 + 0.8 * lum)[None] + 0.15 * chroma
    return rgb.clamp(0.05, 0.95)


def _shifted(scene: torch.Tensor, tx: float, ty: float) -> torch.Tensor:
    """scene sampled at (x + tx, y + ty), bilinear (tx,ty in scene pixels)."""
    ix, iy = math.floor(tx), math.floor(ty)
    fx, fy = tx - ix, ty - iy
    s = torch.roll(scene, shifts=(-iy, -ix), dims=(1, 2))
    s10 = torch.roll(s, shifts=-1, dims=2)
    s01 = torch.roll(s, shifts=-1, dims=1)
    s11 = torch.roll(s01, shifts=-1, dims=2)
    return (1 - fx) * (1 - fy) * s + fx * (1 - fy) * s10 + (1 - fx) * fy * s01 + fx * fy * s11


def _shifted_crop(scene: torch.Tensor, tx: float, ty: float, m: int, hh: int, ww: int) -> torch.Tensor:
    """_shifted(scene, tx, ty)[:, m:m + hh, m:m + ww] without moving the whole scene four times: while the shift stays inside
    the margin m the rolled scene's wrap-around never reaches the crop, so the four neighbours are plain slices -- the same
    elements in the same arithmetic, bit for bit (a 16-frame 4K burst: 27 -> 15 s on the CPU)."""
    ix, iy = math.floor(tx), math.floor(ty)
    if not (m + iy >= 0 and m + ix >= 0 and m + iy + hh + 1 <= scene.shape[1] and m + ix + ww + 1 <= scene.shape[2]):
        return _shifted(scene, tx, ty)[:, m:m + hh, m:m + ww]
    fx, fy = tx - ix, ty - iy
    y0, x0 = m + iy, m + ix
    s = scene[:, y0:y0 + hh, x0:x0 + ww]
    s10 = scene[:, y0:y0 + hh, x0 + 1:x0 + 1 + ww]
    s01 = scene[:, y0 + 1:y0 + 1 + hh, x0:x0 + ww]
    s11 = scene[:, y0 + 1:y0 + 1 + hh, x0 + 1:x0 + 1 + ww]
    return (1 - fx) * (1 - fy) * s + fx * (1 - fy) * s10 + (1 - fx) * fy * s01 + fx * fy * s11


def _rotated(scene: torch.Tensor, tx: float, ty: float, angle_deg: float) -> torch.Tensor:
    """scene sampled at c + R(angle) (p - c) + (tx, ty), bilinear, c = image centre (the rotation stress variant of
    SURVEY.md section 8d; the reference's generator rotates its crops the same way, test_opencv/main.cpp:1896-1907)."""
    _, h, w = scene.shape
    a = math.radians(angle_deg)
    ca, sa = math.cos(a), math.sin(a)
    dev = scene.device
    yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32),
                            indexing="ij")
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    dx, dy = xx - cx, yy - cy
    qx = cx + ca * dx - sa * dy + tx
    qy = cy + sa * dx + ca * dy + ty
    grid = torch.stack([(qx + 0.5) / w * 2 - 1, (qy + 0.5) / h * 2 - 1], -1)[None]
    return F.grid_sample(scene[None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0]


def make_burst(width: int, height: int, frames: int, scale: int = 2, mono: bool = False, seed: int = 1234,
               device="cpu", max_shift: float = 5.0, noise: bool = True, alpha: float = 1e-4, beta: float = 1e-6,
               black: float = 256.0, white: float = 4095.0 - 256.0, shift_seed: Optional[int] = None,
               first_is_reference: bool = True,
               angles_deg: Optional[List[float]] = None,
               keep: Optional[Sequence[int]] = None) -> Tuple[List[torch.Tensor], torch.Tensor, torch.Tensor]:
    """Returns (raw frames [H,W] int16 holding u16 bit patterns, shifts [N,2] in LR px, ground truth [3,sH,sW]).

    ``seed`` fixes the scene; ``shift_seed`` (default: same stream) fixes the per-frame shifts and
    noise, so ranks of a sharded burst can draw different frames of the SAME scene.  ``angles_deg`` (one per frame)
    additionally rotates frame k about the frame centre (needs cfg.preAlign beyond a degree or two).  ``keep``: only
    these frame numbers are rendered (the others come back as None) while the random stream advances as if all were -- a
    rank of a sharded burst gets exactly the frames the one-GPU burst has at those positions."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    s = scale
    m = 32 * s
    hr_h, hr_w = s * height + 2 * m, s * width + 2 * m
    scene = _scene(hr_h, hr_w, gen, device)
    if shift_seed is not None:
        gen = torch.Generator(device=device)
        gen.manual_seed(shift_seed)
    shifts = (torch.rand(frames, 2, generator=gen, device=device) * 2 - 1) * max_shift
    if first_is_reference:
        shifts[0] = 0
    out = []
    yy, xx = torch.meshgrid(torch.arange(height, device=device), torch.arange(width, device=device), indexing="ij")
    cfa_idx = ((yy % 2) + (xx % 2))  # RGGB: (0,0)->R=0, (0,1)/(1,0)->G=1, (1,1)->B=2
    keep_set = None if keep is None else set(int(k) for k in keep)
    for k in range(frames):
        if keep_set is not None and k not in keep_set:
            if noise:  # the draw the frame would have made
                torch.randn((3, height, width), generator=gen, device=device)
            out.append(None)
            continue
        tx, ty = float(shifts[k, 0]) * s, float(shifts[k, 1]) * s
        ang = float(angles_deg[k]) if angles_deg is not None else 0.0
        if ang == 0.0:
            sh = _shifted_crop(scene, tx, ty, m, s * height, s * width)
        else:
            sh = _rotated(scene, tx, ty, ang)[:, m:m + s * height, m:m + s * width]
        lr = F.avg_pool2d(sh[None], s)[0] if s > 1 else sh
        if noise:
            lr = lr + torch.randn(lr.shape, generator=gen, device=device) * torch.sqrt(alpha * lr + beta)
        if mono:
            img = 0.299 * lr[0] + 0.587 * lr[1] + 0.114 * lr[2]
        else:
            img = torch.gather(lr, 0, cfa_idx[None])[0]
        raw = torch.round(img * white + black).clamp(0, 4095).to(torch.int16)
        out.append(raw.contiguous())
    gt = scene[:, m:m + s * height, m:m + s * width].contiguous()
    return out, shifts, gt
