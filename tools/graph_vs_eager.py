"""Whole-burst hipGraph replay vs eager launches at the headline workload (16 x 4K RGGB -> x2): ms per burst."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_frame_super_resolution_amd import synth
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config

dev = torch.device("cuda:0")
W, H, N = 3840, 2160, 16
cfg = default_config(W, H, N, scale=2)
frames, _, _ = synth.make_burst(W, H, N, seed=1236, device=dev)
pipe = BurstPipeline(cfg, dev)
for _ in range(3):
    pipe.process(frames)
torch.cuda.synchronize()

def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

e = timeit(lambda: pipe.process(frames))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    pipe.process(frames)
for _ in range(3): g.replay()
r = timeit(g.replay)
e2 = timeit(lambda: pipe.process(frames))
print(f"eager {e:.3f} ms  graph replay {r:.3f} ms  eager again {e2:.3f} ms")
