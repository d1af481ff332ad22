#!/usr/bin/env python3
"""Statistics of the certainty mask on the bench burst: how often are all certainty texels a wave
of the fuse kernel touches (66 columns x 2 rows x 3 channels) one value?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_frame_super_resolution_amd import synth  # noqa: E402
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config, view_as_tensor  # noqa: E402

dev = torch.device("cuda:0")
W, H, N = 3840, 2160, int(os.environ.get("FRAMES", "5"))
cfg = default_config(W, H, N, scale=2)
frames, shifts, _ = synth.make_burst(W, H, N, seed=1234, device=dev)
pipe = BurstPipeline(cfg, dev)
pipe.reset_accumulators()
pipe.set_reference(frames[0])
pipe.add_frame(frames[0], True)
for k in range(1, N):
    pipe.add_frame(frames[k], False)
    m = view_as_tensor(pipe.debug_views()[1], 4, dev)[..., :3]  # [mh, mw, 3]
    mh, mw = m.shape[:2]
    one = (m == 1.0).all(-1)
    zero = (m == 0.0).all(-1)
    chan_eq = ((m[..., 0] == m[..., 1]) & (m[..., 1] == m[..., 2]))
    # wave footprint: 64 cells (+1 each side) x 2 mask rows
    mm = m[: mh // 2 * 2, : mw // 64 * 64].reshape(mh // 2, 2, mw // 64, 64, 3)
    lo, hi = mm.amin((1, 3, 4)), mm.amax((1, 3, 4))
    print(f"frame {k}: mask mean {m.mean().item():.3f}; texels all-channels == 1: {one.float().mean().item():.3f}, == 0: "
          f"{zero.float().mean().item():.3f}, channels equal: {chan_eq.float().mean().item():.3f}; "
          f"wave footprints with ONE value: {(lo == hi).float().mean().item():.3f}", flush=True)
