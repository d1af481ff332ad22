#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MFSR hot path on MI355X.

Metric (BASELINE.json): Mpix/s end-to-end, N-frame burst -> x2 super-resolved
frame.  One "step" = one whole burst through the hot path (reference products,
per-frame align -> robustness -> warp+fuse, exchange, finish), inputs already
resident in HBM.  Workload at N=1 = BASELINE configs[2]: 16 frames of 3840x2160
RGGB u16, x2 (the configuration north_star quotes its roofline target on).

Multi-GPU: weak scaling over the burst dimension -- every rank aligns and fuses
FRAMES_PER_GPU frames of ONE burst of FRAMES_PER_GPU*N frames into its private
HR accumulators; the accumulators are summed with an RCCL reduce-scatter over
xGMI, each rank finishes its stripe, rank 0 gathers the u16 result
(multi_frame_super_resolution_amd/distributed.py).  `--strong` keeps the burst
at 16 frames and shards it instead.

Also reported on the same JSON line:
  roofline     : the warp+fuse launches (accumulateSuperResFullN: two frames per launch with frame
                 pairing), HIP-event timed around every launch inside the timed region, algorithmic
                 bytes = per-frame figure x frames per launch, vs the 8 TB/s HBM peak;
  cpu_baseline : the CPU oracle pipeline ("port" of the same algorithm; the
                 reference's own CPU path is third-party OpenCV BTVL1, absent here)
                 timed on the host cores on a bounded sample of the same workload, plus the parity of
                 the HIP path against it on that sample (PSNR, fractions off by more than 1 LSB).

Other modes (not the contract's `value`): --h2d (frames streamed from pinned host memory), --no-pair,
--async-fuse, --unfused, --workload {1080p5_gray_x2, 4k16_rggb_x4, 8k8_rggb_x2}, and
MFSR_DIST_BACKEND=gloo (functional rehearsal of the multi-rank schedule on fewer GPUs than ranks).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, frames_per_gpu, scale, mono)
    "4k16_rggb_x2": (3840, 2160, 16, 2, False),      # BASELINE configs[2]
    "1080p5_gray_x2": (1920, 1080, 5, 2, True),      # BASELINE configs[1]
    "4k16_rggb_x4": (3840, 2160, 16, 4, False),      # BASELINE configs[3] (per GPU)
    "8k8_rggb_x2": (7680, 4320, 8, 2, False),        # BASELINE configs[4]: 64-frame 8K burst = 8 frames per GPU at N = 8
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def fuse_bytes_per_launch(W, H, s, mono):
    """ALGORITHMIC bytes of one accumulate launch (one frame), reference structure
    (accumulators read-modify-written in HBM): HR*48 B for imgOut/totalWeights
    (2 x float3 read + write) + the per-frame inputs at their stored resolution:
    raw u16 (LR*2) + flow float2 + kernel-param float4 + certainty float4."""
    lr, hr = W * H, W * H * s * s
    trk = lr if mono else lr // 4          # flow / kernel-param field resolution
    return hr * 48 + lr * 2 + trk * (8 + 16) + (lr // 4) * 16


def cpu_baseline(W, H, s, mono, sample_frames, seed):
    """Oracle pipeline on the host cores, bounded sample: `sample_frames` frames
    (1 reference + the rest moved) of the same frame size."""
    import numpy as np
    from multi_frame_super_resolution_amd.pipeline import default_config
    from multi_frame_super_resolution_amd.synth import make_burst
    from oracle.bindings import oracle
    from oracle.pipeline import OraclePipeline

    frames, _, _ = make_burst(W, H, sample_frames, scale=s, mono=mono, seed=seed, device="cpu")
    nf = [f.numpy().view(np.uint16) for f in frames]
    cfg = default_config(W, H, sample_frames, s, mono)
    op = OraclePipeline(cfg)
    t0 = time.perf_counter()
    o_out, o_q = op.process(nf)
    dt = time.perf_counter() - t0
    res = {
        "value": round(sample_frames * W * H / dt / 1e6, 3),
        "unit": "Mpix/s",
        "cores": int(oracle().num_threads()),
        "kind": "port",
        "sample": f"{sample_frames} frames of {W}x{H} (1 reference + {sample_frames - 1} moved), x{s}, "
                  f"whole pipeline, OpenMP, {dt:.1f} s",
    }
    # the same sample through the HIP path: the "PSNR vs ref" half of BASELINE.json's metric (the oracle is
    # only the checker here)
    try:
        import torch
        from multi_frame_super_resolution_amd.pipeline import BurstPipeline
        dev = torch.device("cuda", torch.cuda.current_device())
        hp = BurstPipeline(cfg, dev)
        h_out, h_q = hp.process([f.to(dev) for f in frames])
        h_out = h_out.cpu().numpy()
        h_q = h_q.cpu().numpy().view(np.uint16)
        hp.close()
        mse = float(np.mean((h_out.astype(np.float64) - o_out.astype(np.float64)) ** 2))
        d8 = np.abs(np.round(h_out * 255.0) - np.round(o_out * 255.0))
        d16 = np.abs(h_q.astype(np.int64) - o_q.astype(np.int64))
        res["parity_on_sample"] = {
            "psnr_db_vs_oracle": round(200.0 if mse == 0 else 10 * np.log10(1.0 / mse), 2),
            "frac_gt_1lsb_8bit": float((d8 > 1).mean()),
            "frac_gt_1lsb_16bit": float((d16 > 1).mean()),
        }
    except Exception as e:  # the throughput line must not depend on the checker
        res["parity_on_sample"] = {"error": repr(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="4k16_rggb_x2", choices=list(WORKLOADS))
    ap.add_argument("--strong", action="store_true", help="fixed 16-frame burst sharded over the GPUs")
    ap.add_argument("--exchange", default="auto", choices=["auto", "reduce", "reduce_scatter"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=4)
    ap.add_argument("--unfused", action="store_true", help="one launch per reference kernel (A/B against the fused path)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: run the RCCL exchange of burst i on the compute stream instead of overlapping it with the "
                         "align+fuse of burst i+1")
    ap.add_argument("--force-pipelined", action="store_true", help="use the two-context pipelined step loop even at N=1 (test)")
    ap.add_argument("--no-pair", action="store_true", help="cfg.pairFrames = 0: one warp+fuse launch per frame (the reference's structure)")
    ap.add_argument("--async-fuse", action="store_true",
                    help="cfg.asyncFuse: warp+fuse on the burst's own stream, overlapping the alignment of the next frames "
                         "(+5 % burst throughput, the fuse launches themselves get 17 % slower; off by default)")
    ap.add_argument("--h2d", action="store_true",
                    help="N=1 only: frames start in pinned HOST memory and stream through a 4-deep device ring on a copy "
                         "stream (the PCIe-inclusive rate quoted in DESIGN.md; `value` of the contract is the HBM-resident run)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # MFSR_DIST_BACKEND=gloo: rehearsal of the multi-rank schedule on a box with fewer GPUs than ranks (the
    # ranks share GPUs, collectives are staged through the host); the measured runs use RCCL ("nccl")
    backend = os.environ.get("MFSR_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from multi_frame_super_resolution_amd import distributed as mdist
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    from multi_frame_super_resolution_amd.synth import make_burst

    W, H, fpg, s, mono = WORKLOADS[args.workload]
    n_frames = fpg if args.strong else fpg * world
    cfg = default_config(W, H, n_frames, s, mono)
    cfg.fused = 0 if args.unfused else 1
    if args.no_pair:
        cfg.pairFrames = 0
    if args.async_fuse and not args.h2d:   # (the upload ring of --h2d reuses a raw buffer right after the next add_frame)
        cfg.asyncFuse = 1
    pipe = BurstPipeline(cfg, dev)

    # synthetic burst: one scene (same seed on every rank), this rank's frames only
    seed = 1234 + 2
    mine = mdist.frames_of_rank(n_frames, rank, world)
    ref_frames, _, _ = make_burst(W, H, 1, scale=s, mono=mono, seed=seed, device=dev)
    shard, _, _ = make_burst(W, H, len(mine), scale=s, mono=mono, seed=seed, device=dev, shift_seed=seed + 100 + rank,
                             first_is_reference=False)
    frames = {k: shard[i] for i, k in enumerate(mine)}
    frames[cfg.reference] = ref_frames[0]
    del shard
    torch.cuda.synchronize()

    # Steps are independent bursts, so at N>1 the exchange (reduce-scatter, stripe finish, gather)
    # of burst i runs on a side stream while the compute stream already aligns and fuses burst
    # i+1 into a second burst context (own workspace + accumulators): collectives overlap compute.
    pipelined = (world > 1 and not args.no_overlap) or args.force_pipelined
    pipes = [pipe]
    step_no = [0]
    if pipelined:
        pipes.append(BurstPipeline(cfg, dev))
        side = torch.cuda.Stream(device=dev)
        ev_acc = [torch.cuda.Event() for _ in pipes]    # accumulate of the burst in context j done (compute stream)
        ev_done = [torch.cuda.Event() for _ in pipes]   # exchange + finish of context j done (side stream)
        used = [False for _ in pipes]

    h2d = args.h2d and world == 1
    if h2d:
        # double-buffered upload: frame k is copied into ring slot k % 4 on the copy stream while the
        # compute stream works on earlier frames; a slot is reused once the add_frame after its
        # frame's own has been issued (frame pairing keeps a raw buffer one call longer)
        order = sorted(frames.keys())
        host = {k: frames[k].cpu().pin_memory() for k in order}
        ring = [torch.empty_like(frames[order[0]]) for _ in range(4)]
        copy_s = torch.cuda.Stream(device=dev)
        prev_end = [None]

    def step_h2d():
        main = torch.cuda.current_stream()
        ev_up = [torch.cuda.Event() for _ in order]
        ev_done = [torch.cuda.Event() for _ in order]
        pipe.begin_burst()
        for i, k in enumerate(order):
            with torch.cuda.stream(copy_s):
                if i >= 4:
                    copy_s.wait_event(ev_done[i - 3])
                elif prev_end[0] is not None:
                    copy_s.wait_event(prev_end[0])
                ring[i % 4].copy_(host[k], non_blocking=True)
                ev_up[i].record(copy_s)
            main.wait_event(ev_up[i])
            if i == 0:
                assert k == cfg.reference
                pipe.set_reference(ring[0])
            pipe.add_frame(ring[i % 4], k == cfg.reference)
            ev_done[i].record(main)
        _, out = pipe.finish(want_float=False, want_u16=True)
        prev_end[0] = torch.cuda.Event()
        prev_end[0].record(main)
        return out

    def step():
        if h2d:
            return step_h2d()
        if not pipelined:
            if world > 1:
                mdist.accumulate_local(pipe, frames, rank, world, n_frames)
                return mdist.exchange_and_finish(pipe, args.exchange)
            return mdist.process_burst(pipe, frames, n_frames=n_frames)
        j = step_no[0] % 2
        step_no[0] += 1
        p = pipes[j]
        main = torch.cuda.current_stream()
        if used[j]:
            main.wait_event(ev_done[j])                 # context j is free again
        mdist.accumulate_local(p, frames, rank, world, n_frames)
        ev_acc[j].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_acc[j])
            if world > 1:
                out = mdist.exchange_and_finish(p, args.exchange)
            else:
                _, out = p.finish(want_float=False, want_u16=True)
            ev_done[j].record(side)
        used[j] = True
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # bring the RCCL communicator up outside the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, device=dev))
    for _ in range(args.warmup):
        step()
    barrier()
    for q in pipes:
        q.L.burst_timing(q._h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches, fused_frames = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
    for q in pipes:
        t_ms, n_l, n_f = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
        q.L.burst_timing_read(q._h, ctypes.byref(t_ms), ctypes.byref(n_l), ctypes.byref(n_f))
        q.L.burst_timing(q._h, 0)
        tot_ms.value += t_ms.value
        launches.value += n_l.value
        fused_frames.value += n_f.value

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_frames * W * H * args.steps / dt / 1e6
        # algorithmic bytes of one launch = per-frame figure x the frames that launch fuses (2 with
        # frame pairing, the default; the last frame of an odd shard goes alone)
        bytes_frame = fuse_bytes_per_launch(W, H, s, mono)
        frames_per_launch = fused_frames.value / max(launches.value, 1)
        bytes_launch = bytes_frame * frames_per_launch
        k_ms = tot_ms.value / max(launches.value, 1)
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "fuse_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        line = {
            "metric": "Mpix/s end-to-end (N-frame burst -> x2 SR)" if s == 2 else f"Mpix/s end-to-end (N-frame burst -> x{s} SR)",
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (", streamed from pinned host memory (4-deep device ring, copy stream)" if h2d else ""),
            "config": {
                "workload": f"{n_frames}-frame {W}x{H} {'gray' if mono else 'RGGB u16'} burst -> x{s} "
                            f"({args.workload}; BASELINE configs[{ {'4k16_rggb_x2': 2, '1080p5_gray_x2': 1, '4k16_rggb_x4': 3, '8k8_rggb_x2': 4}[args.workload] }])",
                "frames_per_gpu": len(mine),
                "burst_frames": n_frames,
                "output_mpix_per_s": round(s * s * W * H * args.steps / dt / 1e6, 2),
                "parallelism": "1 GPU" if world == 1 else f"frame-shard x{world} + RCCL {args.exchange} of HR accumulators"
                               + (" overlapped with the next burst's compute" if pipelined else ""),
                "kernels": "unfused (one launch per reference kernel)" if args.unfused else "fused",
                **({"rehearsal": "gloo backend, ranks share GPUs, collectives staged through the host: not a measurement"}
                   if (world > 1 and backend == "gloo") else {}),
            },
            "roofline": {
                "kernel": ("k_accumulate2xTile / k_accumulate2xStrip" if s == 2 else "k_accumulateSuperRes<GEOM_FULL,fast>")
                          + " (warp+fuse, accumulateSuperResFull[2]; launch = tile kernel + border kernel per frame)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "bytes_per_launch": int(bytes_launch),
                "bytes_per_frame": bytes_frame,
                "frames_per_launch": round(frames_per_launch, 3),
                "avg_launch_ms": round(k_ms, 4),
                "launches_timed": launches.value,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(W, H, s, mono, args.cpu_sample_frames, seed)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    for q in pipes:
        q.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
