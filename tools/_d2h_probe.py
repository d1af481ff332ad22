import ctypes, time, torch, os, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
W = 7680 * 6
Hn = 4320
n = W * Hn
x = torch.zeros(n + 4096, dtype=torch.uint8, device="cuda:0")
y = torch.zeros(n + 4096, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
torch.cuda.synchronize()
for kind in ("1D", "2D pitch==width", "2D width<pitch"):
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind == "1D":
            rc = hip.hipMemcpyAsync(y.data_ptr(), x.data_ptr(), n, 2, s.cuda_stream)
        elif kind == "2D pitch==width":
            rc = hip.hipMemcpy2DAsync(y.data_ptr(), W, x.data_ptr(), W, W, Hn, 2, s.cuda_stream)
        else:
            rc = hip.hipMemcpy2DAsync(y.data_ptr(), W, x.data_ptr(), W, W - 64, Hn, 2, s.cuda_stream)
        assert rc == 0, rc
        s.synchronize()
        dt = time.perf_counter() - t0
        print(f"{kind}: {dt*1e3:.3f} ms = {n/dt/1e9:.1f} GB/s", flush=True)
