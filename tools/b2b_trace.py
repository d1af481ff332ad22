#!/usr/bin/env python3
"""Host bursts back to back (no host synchronisation between them), for a kernel trace of the steady state:
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/b2b_trace.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_frame_super_resolution_amd import synth
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config

dev = torch.device("cuda:0")
W, H, N = 3840, 2160, 16
cfg = default_config(W, H, N, scale=2)
cfg.uploadRing = 32
frames, _, _ = synth.make_burst(W, H, N, seed=1236, device=dev)
host = [f.cpu().pin_memory() for f in frames]
pipe = BurstPipeline(cfg, dev)
outs = [torch.empty(2 * H, 2 * W, 3, dtype=torch.int16).pin_memory() for _ in range(2)]
for i in range(int(os.environ.get("BURSTS", "8"))):
    pipe.process_host(host, outs[i & 1])
pipe.host_sync()
torch.cuda.synchronize()
