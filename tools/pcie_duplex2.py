#!/usr/bin/env python3
"""Which copy mechanisms let the two directions of the host link overlap?  up / down = 1-D hipMemcpyAsync or 2-D hipMemcpy2DAsync
(the runtime may route them to different engines); per combination: each direction alone and both at once (ms per 4K x 16
burst's worth of bytes: 265 MB up, 199 MB down)."""
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


def upload(k, mode):
    if mode == "1d":
        assert hip.hipMemcpyAsync(frames_d[k].data_ptr(), frames_h[k].data_ptr(), W * H * 2, H2D, up.cuda_stream) == 0
    else:
        assert hip.hipMemcpy2DAsync(frames_d[k].data_ptr(), W * 2, frames_h[k].data_ptr(), W * 2, W * 2, H, H2D, up.cuda_stream) == 0


def download_band(i, b, bands, mode):
    rows = 2 * H // bands
    dst, src = out_h[i & 1].data_ptr() + b * rows * row, out_d.data_ptr() + b * rows * row
    if mode == "1d":
        assert hip.hipMemcpyAsync(dst, src, rows * row, D2H, down.cuda_stream) == 0
    else:
        assert hip.hipMemcpy2DAsync(dst, row, src, row, row, rows, D2H, down.cuda_stream) == 0


def run(um, dm, do_up, do_down, interleave, reps=12, bands=8):
    def burst(i):
        if interleave and do_up and do_down:
            for k in range(N):
                upload(k, um)
                if k % 2 == 1:
                    download_band(i, k // 2, bands, dm)
        else:
            if do_up:
                for k in range(N):
                    upload(k, um)
            if do_down:
                for b in range(bands):
                    download_band(i, b, bands, dm)
    for i in range(2):
        burst(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        burst(i)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


res = {}
for um in ("1d", "2d"):
    for dm in ("2d", "1d"):
        key = f"up_{um}_down_{dm}"
        res[key] = {"up_alone": run(um, dm, True, False, False), "down_alone": run(um, dm, False, True, False),
                    "both_queued": run(um, dm, True, True, False), "both_interleaved": run(um, dm, True, True, True)}
        print(key, res[key], flush=True)
print(json.dumps(res))
