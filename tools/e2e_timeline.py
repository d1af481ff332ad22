#!/usr/bin/env python3
"""One-burst-in-flight end-to-end timeline: run under
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/<tag> -- python3 tools/e2e_timeline.py
and summarise with tools/e2e_timeline.py --summarise gpurun_out/<tag>: when, relative to the first H2D copy of the last
burst, the uploads end, the kernels of each kind start and end, and the downloads start and end."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N, s = 3840, 2160, 16, 2
    dev = torch.device("cuda:0")
    frames, _, _ = make_burst(W, H, N, scale=s, mono=False, seed=1236, device=dev)
    cfg = default_config(W, H, N, s, False)
    cfg.uploadRing = 16
    pipe = BurstPipeline(cfg, dev)
    host = [f.cpu().pin_memory() for f in frames]
    import time
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.process_host(host)
        pipe.host_sync()
        print(f"burst {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    pipe.close()


def summarise(d):
    kt = sorted(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
    mc = kt.replace("_kernel_trace.csv", "_memory_copy_trace.csv")     # the same process's copy records
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in csv.DictReader(open(mc))]
    big = [r for r in rows if r[1] - r[0] > 150_000]                      # frame uploads (0.3 ms) and image bands (0.45 ms)
    h = [r for r in big if "HOST_TO_DEVICE" in r[2]][-16:]                # the last burst's 16 uploads
    t0 = h[0][0]
    ms = lambda t: (t - t0) / 1e6
    print(f"H2D (last burst): 16 copies, first starts 0.000, last ends {ms(h[-1][1]):.3f} ms, {(h[0][1] - h[0][0]) / 1e6:.3f} ms each")
    down = [r for r in big if "HOST_TO_DEVICE" not in r[2] and r[0] > t0]  # 2-D device -> host copies (SDMA; listed as D2D / D2H)
    if down:
        print(f"D2H (SDMA, 2-D copies): {len(down)} copies, first starts {ms(down[0][0]):.3f}, last ends {ms(down[-1][1]):.3f} ms, "
              f"{sum(r[1] - r[0] for r in down) / 1e6:.3f} ms busy")
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt)) if int(r["Start_Timestamp"]) >= t0]
    groups = {}
    for a, b, n in ks:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
        g = groups.setdefault(n, [a, b, 0, 0])
        g[0], g[1], g[2], g[3] = min(g[0], a), max(g[1], b), g[2] + 1, g[3] + (b - a)
    for n, g in sorted(groups.items(), key=lambda kv: kv[1][0]):
        print(f"  {n:42s} n={g[2]:3d} first start {ms(g[0]):7.3f} last end {ms(g[1]):7.3f} busy {g[3] / 1e6:6.3f} ms")
    for key in ("k_accumulate2xTile", "k_finishFused", "copyBuffer"):
        sel = sorted([k for k in ks if key in k[2]])
        print(f"  {key} (start, end):", [(round(ms(a), 2), round(ms(b), 2)) for a, b, _ in sel])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
