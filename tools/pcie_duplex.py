#!/usr/bin/env python3
"""The copy pattern of round 3's host bursts: per 4K x 16 burst 16 x 16.6 MB up as 1-D hipMemcpyAsync calls and one 199 MB u16
image down as 2-D hipMemcpy2DAsync bands.  Both directions alone and together.  (This combination SERIALISES on this ROCm --
8.4 ms, the sum -- although the link is full duplex: tools/pcie_duplex2.py compares the mechanisms; the library's uploads are
2-D copies since.)"""
import ctypes
import json
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                 ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
H2D, D2H = 1, 2
dev = torch.device("cuda:0")
W, H, N = 3840, 2160, 16
frames_h = [torch.empty(H, W, dtype=torch.int16).pin_memory() for _ in range(N)]
frames_d = [torch.empty(H, W, dtype=torch.int16, device=dev) for _ in range(N)]
out_d = torch.empty(2 * H, 2 * W, 3, dtype=torch.int16, device=dev)
out_h = [torch.empty(2 * H, 2 * W, 3, dtype=torch.int16).pin_memory() for _ in range(2)]
up, down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
row = 2 * W * 6


def uploads():
    for k in range(N):
        assert hip.hipMemcpyAsync(frames_d[k].data_ptr(), frames_h[k].data_ptr(), W * H * 2, H2D, up.cuda_stream) == 0


def download(i):
    bands = 8
    rows = 2 * H // bands
    for b in range(bands):
        assert hip.hipMemcpy2DAsync(out_h[i & 1].data_ptr() + b * rows * row, row, out_d.data_ptr() + b * rows * row, row, row, rows, D2H,
                                    down.cuda_stream) == 0


def run(do_up, do_down, reps=20):
    for i in range(3):
        if do_up:
            uploads()
        if do_down:
            download(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        if do_up:
            uploads()
        if do_down:
            download(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


up_mb, down_mb = N * W * H * 2 / 1e6, 4 * W * H * 6 / 1e6
a, b, c = run(True, False), run(False, True), run(True, True)
print(json.dumps({"upload_only_ms_per_burst": round(a, 3), "upload_GBps": round(up_mb / a, 1), "download_only_ms_per_burst": round(b, 3),
                  "download_GBps": round(down_mb / b, 1), "both_ms_per_burst": round(c, 3), "both_total_GBps": round((up_mb + down_mb) / c, 1),
                  "upload_mb": up_mb, "download_mb": down_mb,
                  "note": "1-D uploads queued against 2-D downloads: serialised on this ROCm (see tools/pcie_duplex2.py)"}))
