#!/usr/bin/env python3
"""The warp+fuse launch of the headline workload with forced certainty masks: all ones (every wave takes the saturated
pixel body) against 0.999999 (every wave takes the general body), and the share of saturated waves in the bench's burst."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_frame_super_resolution_amd import capi
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
from multi_frame_super_resolution_amd.synth import make_burst
from multi_frame_super_resolution_amd.pipeline import view_as_tensor

W, H, N, s = 3840, 2160, 16, int(os.environ.get("SCALE", "2"))
dev = torch.device("cuda:0")
seed = 1236
ref, _, _ = make_burst(W, H, 1, scale=s, mono=False, seed=seed, device=dev)
sh, _, _ = make_burst(W, H, N - 1, scale=s, mono=False, seed=seed, device=dev, shift_seed=seed + 100, first_is_reference=False)
frames = [ref[0]] + list(sh)
cfg = default_config(W, H, N, s, False)
STATS = os.environ.get("SAT_STATS", "1") != "0"
REPS = int(os.environ.get("SAT_REPS", "20"))
pipe = BurstPipeline(cfg, dev)
pipe.begin_burst()
pipe.set_reference(frames[0])
shares = []
for k in range(1, N if STATS else 2):
    pipe.add_frame(frames[k])
    pipe.flush()
    _, mt = pipe.frame_views(0)
    m = view_as_tensor(mt, 4, dev)[..., :3]
    sat = (m == 1.0).all(-1)
    hh, ww = sat.shape
    a = sat[: hh // 2 * 2, : ww // 64 * 64].reshape(hh // 2, 2, ww // 64, 64)
    shares.append((float(sat.float().mean()), float(a.all(1).all(-1).float().mean())))
print("per frame (texels saturated, 64x2 windows saturated):", " ".join(f"({a:.3f},{b:.3f})" for a, b in shares))
print("mean window share %.3f" % np.mean([b for _, b in shares]))

L = capi.lib()
hrW, hrH = W * s, H * s
fw, fh = W // 2, H // 2
_, _, kp_t, _ = pipe.debug_views()
kp = view_as_tensor(kp_t, 4, dev).contiguous()
g = torch.Generator(device="cpu").manual_seed(5)
yy, xx = torch.meshgrid(torch.arange(fh, dtype=torch.float32), torch.arange(fw, dtype=torch.float32), indexing="ij")
n = 4
raws = frames[1:1 + n]
flows = [torch.stack([1.3 - 0.9 * k + 0.001 * xx, -2.2 + 1.1 * k + 0.002 * yy], -1).contiguous().to(dev) for k in range(n)]
acc_i = torch.zeros(hrH, hrW, 3, device=dev)
acc_w = torch.zeros(hrH, hrW, 3, device=dev)
P = ctypes.c_void_p * n
T = capi.Tex2D * n
shs = T(*[capi.Tex2D(t.data_ptr(), fw * 8, fw, fh) for t in flows])
white, black = capi.f3((4095.0, 4095.0, 4095.0)), capi.f3((64.0, 64.0, 64.0))
L.set_cfa_pattern((ctypes.c_int32 * 4)(*cfg.cfa))
res = {}
for name, val in (("ones", 1.0), ("0.999999", 0.999999), ("ones", 1.0), ("0.999999", 0.999999)):
    masks = [torch.full((H // 2, W // 2, 4), val, device=dev) for _ in range(n)]
    def launch():
        L.accumulateSuperResFullN(n, P(*[t.data_ptr() for t in raws]), acc_i.data_ptr(), acc_w.data_ptr(),
                                  P(*[t.data_ptr() for t in masks]), capi.Tex2D(kp.data_ptr(), fw * 16, fw, fh), shs,
                                  white, black, W, H, s, hrW * 12, (W // 2) * 16, 0, None)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        launch()
    e1.record()
    torch.cuda.synchronize()
    print(f"masks = {name}: {e0.elapsed_time(e1) / REPS:.4f} ms per 4-frame launch (tile + margin)")
