#!/usr/bin/env python3
"""Does an RCCL transport kernel get onto the CUs while a warp+fuse launch fills them?  

RCCL's send / receive are KERNELS.  The x2 warp+fuse launch keeps four 128-VGPR workgroups per CU resident -- the whole
register file -- and round 3's timeline showed a 43 us tracker launch on a default-priority stream waiting 0.7 ms behind it.
csrc/dist.cpp therefore creates the comm stream with the HIGHEST priority.  This probe measures, on ONE GPU with a one-rank
RCCL communicator, what that buys: an ncclSend / ncclRecv pair of the rank with ITSELF (a real RCCL kernel: the same
channels, workgroup size and registers a peer exchange launches) of the size of one packed exchange message is issued in the
middle of a 4-frame 4K warp+fuse launch,

    (a) alone on an idle GPU,
    (b) during the launch, on a stream of the highest priority (what dist.cpp does),
    (c) during the launch, on a default-priority stream (round 3's behaviour),

and reports the time from "eligible" (an event on the comm stream) to "complete".  With kernel tracing on
(rocprofv3 --kernel-trace -- python3 tools/rccl_under_fuse.py) the trace shows where the RCCL kernel sat against the
k_accumulate2xTile dispatch.
"""
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class UID(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def main():
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    from multi_frame_super_resolution_amd.synth import make_burst

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rccl = ctypes.CDLL("librccl.so")
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UID)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
    for f in (rccl.ncclSend, rccl.ncclRecv):
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    uid = UID()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0

    W, H, N = 3840, 2160, 5
    frames, _, _ = make_burst(W, H, N, scale=2, seed=1236, device=dev)
    cfg = default_config(W, H, N, 2, False)
    cfg.asyncFuse = 0                      # the multi-GPU layer fuses on the caller's stream (mfsr_burst_fuse_rows)
    pipe = BurstPipeline(cfg, dev)
    msg_bytes = int(os.environ.get("MSG_MB", "19")) * 1000 * 1000     # one packed exchange message at configs[2] on 8 GPUs
    src = torch.empty(msg_bytes, dtype=torch.uint8, device=dev).fill_(3)
    dst = torch.empty(msg_bytes, dtype=torch.uint8, device=dev)

    def self_exchange(stream):
        rccl.ncclGroupStart()
        assert rccl.ncclSend(src.data_ptr(), msg_bytes, 1, 0, comm, stream.cuda_stream) == 0      # ncclUint8 = 1
        assert rccl.ncclRecv(dst.data_ptr(), msg_bytes, 1, 0, comm, stream.cuda_stream) == 0
        rccl.ncclGroupEnd()

    hi = torch.cuda.Stream(device=dev, priority=-1)
    lo = torch.cuda.Stream(device=dev, priority=0)
    main_s = torch.cuda.current_stream()
    for st in (hi, lo):                    # bring RCCL's kernels / channels up
        self_exchange(st)
    torch.cuda.synchronize()
    assert bool((dst == 3).all())

    # torch.cuda._sleep counts device clock ticks of an unspecified rate: calibrate ticks per microsecond
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    c0.record()
    torch.cuda._sleep(4000000)
    c1.record()
    torch.cuda.synchronize()
    ticks_per_us = 4000000 / (c0.elapsed_time(c1) * 1e3)

    def one(stream, under_fuse, delay_us):
        """returns (ms from eligible to complete, ms of the fuse launch)"""
        e_go, e0, e1 = torch.cuda.Event(), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pipe.begin_burst()
        pipe.set_reference(frames[0])
        pipe.add_frame(frames[0], True)
        for k in (1, 2, 3):
            pipe.add_frame(frames[k], False)       # the fourth frame completes the group: alignment, then the fuse launch
        pipe.flush()
        torch.cuda.synchronize()
        # second group of the same burst, this time observed: 4 more frames -> align + one 4-frame fuse launch
        for k in (1, 2, 3):
            pipe.add_frame(frames[k], False)
        if under_fuse:
            pipe.L.burst_timing(pipe._h, 1)
        e_go.record(main_s)                        # everything before the observed launch is enqueued
        pipe.add_frame(frames[4], False)           # -> alignment of the group (~0.7 ms), then the fuse launch (~0.9 ms)
        if under_fuse:
            stream.wait_event(e_go)
            with torch.cuda.stream(stream):
                torch.cuda._sleep(int(delay_us * ticks_per_us))   # lands the op in the middle of the fuse launch
                e0.record(stream)
                self_exchange(stream)
                e1.record(stream)
        else:
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                e0.record(stream)
                self_exchange(stream)
                e1.record(stream)
        pipe.flush()
        torch.cuda.synchronize()
        fuse_ms = None
        if under_fuse:
            t, n, fr = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
            pipe.L.burst_timing_read(pipe._h, ctypes.byref(t), ctypes.byref(n), ctypes.byref(fr))
            pipe.L.burst_timing(pipe._h, 0)
            fuse_ms = t.value / max(n.value, 1)
        return e0.elapsed_time(e1), fuse_ms

    res = {}
    delay = float(os.environ.get("DELAY_US", "1100"))      # alignment of the group ~0.7 ms + into the fuse launch
    for name, st, under in (("idle_high", hi, False), ("idle_default", lo, False), ("under_fuse_high_priority", hi, True),
                            ("under_fuse_default_priority", lo, True)):
        ts, fs = [], []
        for i in range(12):
            t, f = one(st, under, delay)
            if i >= 2:
                ts.append(t)
                if f:
                    fs.append(f)
        res[name] = {"ms_median": round(statistics.median(ts), 4), "ms_max": round(max(ts), 4), "ms_min": round(min(ts), 4),
                     **({"fuse_launch_ms_median": round(statistics.median(fs), 4)} if fs else {})}
    res["message_mb"] = msg_bytes / 1e6
    res["sleep_ticks_per_us"] = round(ticks_per_us, 2)
    res["note"] = ("ncclSend + ncclRecv of the rank with itself (one RCCL kernel) issued ~%d us after the group's launches were enqueued; "
                   "time from the op becoming eligible on its stream to its completion" % int(delay))
    print(json.dumps(res))
    pipe.close()


if __name__ == "__main__":
    main()
