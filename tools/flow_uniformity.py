#!/usr/bin/env python3
"""How uniform are the rounded shifts of the bench's flow fields?  For every moved frame of the
4K x16 RGGB burst: fraction of 4-pixel strips with more than one rounded shift, and the fraction
of wave footprints (256x1 vs 32x16-interleaved HR pixels) that contain such a strip or more than
one shift parity."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_frame_super_resolution_amd import synth  # noqa: E402
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config, view_as_tensor  # noqa: E402

dev = torch.device("cuda:0")
W, H, N = 3840, 2160, int(os.environ.get("FRAMES", "6"))
cfg = default_config(W, H, N, scale=2)
frames, shifts, _ = synth.make_burst(W, H, N, seed=1234, device=dev)
pipe = BurstPipeline(cfg, dev)
pipe.reset_accumulators()
pipe.set_reference(frames[0])
pipe.add_frame(frames[0], True)
for k in range(1, N):
    pipe.add_frame(frames[k], False)
    flow = view_as_tensor(pipe.debug_views()[0], 2, dev)  # [fh, fw, 2], LR px units
    fh, fw = flow.shape[:2]
    # bilinear to HR like the texture fetch (align_corners=False == texel centres), close enough for statistics
    hr = F.interpolate(flow.permute(2, 0, 1)[None], size=(2 * H, 2 * W), mode="bilinear", align_corners=False)[0]
    r = torch.round(2 * hr).to(torch.int32)                       # [2, hrH, hrW]
    strips = r.view(2, 2 * H, 2 * W // 4, 4)
    nonuni = (strips != strips[..., :1]).any(-1).any(0)            # [hrH, hrW/4]
    par = (strips[..., 0] & 1)                                     # parity per strip (sx&1, sy&1)
    code = par[0] + 2 * par[1]
    def frac_waves(bad, code, sh, sw, interleave):
        hh, ww = bad.shape
        hh2, ww2 = hh // sh * sh, ww // sw * sw
        b = bad[:hh2, :ww2].reshape(hh2 // sh, sh, ww2 // sw, sw)
        c = code[:hh2, :ww2].reshape(hh2 // sh, sh, ww2 // sw, sw)
        if interleave:   # rows of one parity only
            b, c = b[:, ::2], c[:, ::2]
        anybad = b.any(1).any(-1)
        mixed = (c.amax((1, 3)) != c.amin((1, 3)))
        return anybad.float().mean().item(), (mixed & ~anybad).float().mean().item()
    a1, m1 = frac_waves(nonuni, code, 1, 64, False)
    a2, m2 = frac_waves(nonuni, code, 16, 8, True)
    print(f"frame {k}: true shift {shifts[k].tolist()} flow mean {flow.mean((0,1)).tolist()} std {flow.std((0,1)).tolist()} "
          f"non-uniform strips {nonuni.float().mean().item():.4f}; waves with one: 256x1 {a1:.3f} (+mixed parity {m1:.3f}), "
          f"32x16i {a2:.3f} (+mixed {m2:.3f})", flush=True)
