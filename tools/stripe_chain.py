#!/usr/bin/env python3
"""The compute chain A of ONE rank of a G-rank stripes burst, measured on one GPU (DESIGN.md section 7's model term): reference
products for the rank's stripe (mfsr_burst_set_reference_rows), alignment of its N / G frames as one batch
(mfsr_burst_align_frames), all N frames fused onto its stripe in groups of four (mfsr_burst_fuse_rows), finish of the stripe
(mfsr_burst_finish_rows) -- the calls csrc/dist.cpp::process_stripes / stripes_back make, back to back on one stream, without
any exchange.  Prints ms per burst for G = 1, 2, 4, 8 (rank G // 2's stripe) and the implied upper bound of the speed-up
max-over-terms would allow.

    python3 tools/stripe_chain.py [workload: 4k16x2 | 4k16x4 | 8k64x2]
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_frame_super_resolution_amd import capi, synth
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config

WL = {"4k16x2": (3840, 2160, 16, 2), "4k16x4": (3840, 2160, 16, 4), "8k64x2": (7680, 4320, 64, 2)}
W, H, N, s = WL[sys.argv[1] if len(sys.argv) > 1 else "4k16x2"]
dev = torch.device("cuda:0")
distinct = min(N, 16)
frames, _, _ = synth.make_burst(W, H, distinct, scale=s, seed=1236, device=dev)
frames = [frames[k % distinct] if k else frames[0] for k in range(N)]
cfg = default_config(W, H, N, s, False)
pipe = BurstPipeline(cfg, dev)
L = pipe.L
st = torch.cuda.current_stream().cuda_stream
flows, masks = zip(*[pipe.new_frame_products() for _ in range(N)])
# every frame's products once (the other ranks' share, which the exchange would deliver)
pipe.set_reference(frames[0])
for k in range(N):
    pipe.align_frame(frames[k], k == 0, flows[k], masks[k])
torch.cuda.synchronize()
group = pipe.group_size()
res = {}
for G in (1, 2, 4, 8):
    r = G // 2
    plan = pipe.stripe_plan(G, r, 64)
    own = list(range(r, N, G))
    P = ctypes.c_void_p * len(own)
    raws = P(*[frames[k].data_ptr() for k in own])
    isref = (ctypes.c_int * len(own))(*[1 if k == 0 else 0 for k in own])
    fl = P(*[flows[k].data_ptr() for k in own])
    mk = P(*[masks[k].data_ptr() for k in own])

    def burst():
        L.burst_set_reference_rows(pipe._h, frames[0].data_ptr(), plan.rowBegin, plan.rowEnd, st)
        L.burst_align_frames(pipe._h, len(own), raws, isref, fl, flows[0].stride(0) * 4, mk, masks[0].stride(0) * 4, st)
        for k0 in range(0, N, group):
            ks = list(range(k0, min(k0 + group, N)))
            pipe.fuse_rows([frames[k] for k in ks], [flows[k] for k in ks], [masks[k] for k in ks], plan.rowBegin, plan.rowEnd, k0 == 0)
        pipe.finish_rows(plan.rowBegin, plan.rowEnd - plan.rowBegin)

    for _ in range(3):
        burst()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        burst()
    e1.record()
    torch.cuda.synchronize()
    res[G] = e0.elapsed_time(e1) / reps
    print(f"G={G}: rank {r} aligns {len(own)} frames, fuses {N} frames on HR rows [{plan.rowBegin}, {plan.rowEnd}): {res[G]:.3f} ms per burst", flush=True)
print(json.dumps({"workload": sys.argv[1] if len(sys.argv) > 1 else "4k16x2", "compute_chain_ms": {str(g): round(v, 3) for g, v in res.items()},
                  "speedup_bound_vs_G1": {str(g): round(res[1] / v, 2) for g, v in res.items()}}))
