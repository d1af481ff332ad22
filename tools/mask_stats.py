#!/usr/bin/env python3
"""Certainty (robustness-mask) statistics of the headline bench's synthetic burst: what share of the mask texels is
saturated (all channels == 1), and what share of wave footprints of the fuse kernel would see only saturated texels."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_frame_super_resolution_amd.pipeline import default_config
from multi_frame_super_resolution_amd.synth import make_burst
from tests.burst_compare import run_hip

W, H, N, s = 3840, 2160, 16, 2      # the headline burst (bench.py: seed 1236)
if len(sys.argv) > 1:
    W, H = (int(v) for v in sys.argv[1].split("x"))
if len(sys.argv) > 2:
    N = int(sys.argv[2])
frames, _, _ = make_burst(W, H, N, scale=s, mono=False, seed=1236, device="cpu")
cfg = default_config(W, H, N, s, False)
h = run_hip(cfg, frames)
tot = {}
for k in range(N):
    if k == cfg.reference:
        continue
    m = h["masks"][k][..., :3]
    sat = np.all(m == 1.0, axis=-1)
    eq = (m[..., 0] == m[..., 1]) & (m[..., 1] == m[..., 2])
    print(f"frame {k}: mask shape {m.shape} mean {m.mean():.4f} texels all-1: {sat.mean():.4f}; channels equal: {eq.mean():.4f}; zero: {np.all(m==0,axis=-1).mean():.4f}")
    # footprints: a wave of the tile kernel = 64 strips of 4 HR px; try 64x1 strips (65 cells x 2 rows) and 16x4 (17 cells x 3 rows)
    for (cw, ch) in ((66, 2), (18, 3), (34, 2), (3, 2)):
        hh, ww = sat.shape
        a = sat[: hh // ch * ch, : ww // cw * cw].reshape(hh // ch, ch, ww // cw, cw)
        share = float(a.all(axis=(1, 3)).mean())
        tot.setdefault((cw, ch), []).append(share)
        print(f"    windows {cw}x{ch} all saturated: {share:.4f}")
print("mean over the moved frames (the reference frame's mask is all 1: the share of a 16-frame burst's (wave, frame) pairs is (15 x this + 1) / 16):")
for (cw, ch), v in tot.items():
    m = float(np.mean(v))
    print(f"    footprint {cw} x {ch} texels: {m:.4f} of the moved frames' windows saturated -> {(m * (N - 1) + 1) / N:.4f} of the burst's")
