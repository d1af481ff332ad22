#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MFSR hot path on MI355X.

Metric (BASELINE.json): Mpix/s end-to-end, N-frame burst -> x2 super-resolved
frame.  One "step" = one whole burst through the hot path (reference products,
per-frame align -> robustness -> warp+fuse, exchange, finish), inputs already
resident in HBM.  Workload at N=1 = BASELINE configs[2]: 16 frames of 3840x2160
RGGB u16, x2 (the configuration north_star quotes its roofline target on).

Multi-GPU (--gpus N, one rank per GPU): STRONG scaling of the workload's fixed burst by default.  Every rank aligns
frames {k : k mod N = rank}; the ranks exchange the rows of raw / flow / certainty their HR row stripes read
(point-to-point over xGMI), every rank fuses ALL frames onto its stripe and finishes it, rank 0 collects the u16
stripes -- bit-identical to the 1-GPU burst (include/mfsr_dist.h, csrc/dist.cpp: RCCL directly; --dist-impl torch runs
the torch.distributed mirror).  --exchange reduce | reduce_scatter sum private accumulators instead; --weak shards
frames-per-GPU x N frames.

Also reported on the same JSON line:
  end_to_end   : SURVEY.md 8(d)'s definition -- first H2D enqueue to D2H of the result complete, one burst in flight,
                 median of 20 (frames in pinned host memory through mfsr_burst_*_host);
  roofline     : the warp+fuse launches (a group of up to four frames per launch), HIP-event timed around every launch inside the timed
                 region; frac = bytes a launch MUST move / time / 8 TB/s; "bound": "valu" with the VALU-issue ceiling
                 from the PMC instruction count; traffic = PMC HBM bytes; reference_structure = SURVEY's per-frame RMW
                 accounting (what the reference's kernel would move), labelled as such;
  cpu_baseline : the CPU oracle pipeline ("port" of the same algorithm; the reference's own CPU path is third-party
                 OpenCV BTVL1, absent here) timed on the host cores on a bounded sample of the same workload, plus the
                 flip-set parity classification of the HIP path against it on that sample (tests/flipset.py).

Other modes (not the contract's `value`): --h2d (every burst's frames start in pinned host memory), --no-pair,
--no-async-fuse, --unfused, --workload {1080p5_gray_x2, 4k16_rggb_x4, 8k8_rggb_x2}, and
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
    "8k8_rggb_x2": (7680, 4320, 8, 2, False),        # BASELINE configs[4]'s per-GPU share at N = 8 (8 of the 64 frames)
    "8k64_rggb_x2": (7680, 4320, 64, 2, False),      # BASELINE configs[4] as stated: the 64-frame 8K burst (4.2 GB of raw frames)
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VALU_CLOCK_GHZ = 2.4    # MI355X peak engine clock


def fuse_input_bytes_per_frame(W, H, s, mono):
    """Per-frame inputs of the warp+fuse kernel at their stored resolution: raw u16 (LR*2) + flow float2 +
    certainty float4 (half res); the kernel-parameter float4 field is per launch, not per frame."""
    lr = W * H
    trk = lr if mono else lr // 4          # flow / kernel-param field resolution
    return lr * 2 + trk * 8 + (lr // 4) * 16


def fuse_bytes_reference_structure(W, H, s, mono):
    """SURVEY.md section 8(d)'s reference-structure figure for ONE frame: one launch per frame, both accumulator
    plane-sets read-modify-written in HBM (HR*48 B) + every input.  What the reference's kernel moves per frame; the
    frame-grouped kernel of this build moves the accumulators once per group of up to four frames, so this is NOT what it moves."""
    lr, hr = W * H, W * H * s * s
    trk = lr if mono else lr // 4
    return hr * 48 + fuse_input_bytes_per_frame(W, H, s, mono) + trk * 16


def fuse_bytes_must_move(W, H, s, mono, frames_in_launch, first_of_burst):
    """ALGORITHMIC bytes one warp+fuse launch has to move: the two accumulator plane-sets once (read + write, HR*48 B;
    the first launch of a burst overwrites them: HR*24 B), the kernel-parameter field once, and the per-frame inputs of
    the frames it fuses."""
    lr, hr = W * H, W * H * s * s
    trk = lr if mono else lr // 4
    return hr * (24 if first_of_burst else 48) + trk * 16 + frames_in_launch * fuse_input_bytes_per_frame(W, H, s, mono)


def burst_fuse_bytes(W, H, s, mono, frames, per):
    """(launches, must-move bytes) of the warp+fuse launches of one burst of `frames` frames on one rank,
    `per` frames per launch (mfsr_burst_group_size)."""
    per = max(int(per), 1)
    launches, total, left, first = 0, 0, frames, True
    while left > 0:
        n = min(per, left)
        total += fuse_bytes_must_move(W, H, s, mono, n, first)
        first = False
        left -= n
        launches += 1
    return launches, total


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
    # the same sample through the HIP path: the "PSNR vs ref" half of BASELINE.json's metric, with the flip-set
    # classification of tests/flipset.py (the oracle is only the checker here)
    try:
        from tests.burst_compare import classify, run_hip, run_oracle
        h = run_hip(cfg, frames, device=f"cuda:{__import__('torch').cuda.current_device()}")
        o = run_oracle(cfg, nf)   # per-frame flows and masks for the classification (not timed)
        rep = classify(cfg, h, o)
        res["parity_on_sample"] = {
            "psnr_db_vs_oracle": rep["psnr_db_vs_oracle"],
            "flip_fraction": rep["flip_fraction"],
            "n_gt_1lsb_8bit_outside_flip_set": rep["n_gt1_8bit_outside"],
            "max_8bit_outside_flip_set": rep["max8_outside"],
            "max_8bit_inside_flip_set": rep["max8_inside"],
            "frac_gt_1lsb_8bit": rep["frac_gt1_8bit"],
            "frac_gt_1lsb_16bit": rep["frac_gt1_16bit"],
            "max_flow_diff_px": rep["max_flow_diff_px"],
            # the u16 / accumulator half of the contract (tests/burst_compare.py::continuous_checks), outside the flip set:
            # u16 samples whose channel weight is >= 0.03 must be within 1 LSB16; lighter ones are "excused" from that
            # bound (counted here, bounded by 1 + 0.05 / weight); accumulators within 3e-5 rel + 1e-6
            "excused_fraction": rep["excused_fraction"],
            "max16_excused": rep["max16_excused"],
            "n_excused_over_bound": rep["n_excused_over_bound"],
            "max16_outside_well_weighted": rep["max16_outside_well_weighted"],
            "n_acc_violations_outside": rep["n_acc_violations_outside"],
            "note": "flip set = HR pixels where a rounding / threshold decision fed by the (non-bit-exact) Lucas-Kanade flow "
                    "differs between HIP and oracle (tests/flipset.py); outside it every 8-bit sample is within 1 LSB",
        }
    except Exception as e:  # the throughput line must not depend on the checker
        res["parity_on_sample"] = {"error": repr(e)}
    return res


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def spawn_ranks(n, argv, extra_env=None, timeout_s=None):
    """`python3 bench.py --gpus N` without a launcher: this parent makes NO GPU call (it never imports torch); it starts
    N fresh child processes of this same script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), relays rank 0's JSON line, and returns non-zero if any child fails (the others are then
    terminated by their exact PIDs).  Returns (rc, rank-0 stdout)."""
    import subprocess
    port = _free_port()
    timeout_s = timeout_s or float(os.environ.get("MFSR_BENCH_RANK_TIMEOUT_S", "900"))
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MFSR_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    t_end = time.time() + timeout_s
    rc, out0 = 0, ""
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        if live and (rc != 0 or time.time() > t_end):
            if rc == 0:
                rc = 124
                print(f"bench.py: ranks {sorted(live)} still running after {timeout_s:.0f} s", file=sys.stderr)
            time.sleep(2.0 if rc != 124 else 0.0)        # a failing rank usually takes its peers down with it
            for r in sorted(live):
                if procs[r].poll() is None:
                    procs[r].terminate()
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            live.clear()
            break
        if live:
            if 0 not in live:
                time.sleep(0.05)
            else:
                try:                                       # drain rank 0's pipe while waiting (one JSON line, but be safe)
                    o, _ = procs[0].communicate(timeout=0.2)
                    out0 += o or ""
                except subprocess.TimeoutExpired:
                    pass
    if procs[0].stdout and not procs[0].stdout.closed:
        try:
            out0 += procs[0].stdout.read() or ""
        except Exception:
            pass
    return rc, out0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # 20 bursts of 8 ms: the clocks settle after the first few
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="4k16_rggb_x2", choices=list(WORKLOADS))
    ap.add_argument("--strong", action="store_true", help="(default at N > 1) the workload's fixed burst sharded over the GPUs")
    ap.add_argument("--weak", action="store_true", help="N > 1: frames-per-GPU x N frames in one burst (per-GPU alignment work fixed)")
    ap.add_argument("--exchange", default="stripes", choices=["stripes", "auto", "reduce", "reduce_scatter"],
                    help="N > 1: stripes (default: p2p exchange of LR products, fuse sharded over HR row stripes, bit-identical "
                         "to 1 GPU), reduce (north_star's wording: accumulators onto rank 0), reduce_scatter")
    ap.add_argument("--dist-impl", default="auto", choices=["auto", "rccl", "torch", "local"],
                    help="N > 1: rccl = the C-ABI multi-GPU layer (libmfsr_dist.so), one process per GPU, RCCL directly; torch = its "
                         "torch.distributed mirror (distributed.py; also what MFSR_DIST_BACKEND=gloo rehearsals use); local = the "
                         "same C-ABI layer with all N ranks in THIS process (mfsr_dist_group_*: one thread per rank, peer copies "
                         "on the copy engines: no transport kernel competes with the warp+fuse launches for the CUs); auto (default) "
                         "= rccl under a launcher that started one process per GPU (WORLD_SIZE set: torch.distributed.run), local for "
                         "the plain command `bench.py --gpus N` when the node shows N devices, else rccl ranks started from here.  "
                         "The line's top-level \"transport\" says which ran")
    ap.add_argument("--virtual-ranks", action="store_true",
                    help="--dist-impl local: all N ranks on device 0 (functional rehearsal on a one-GPU box, not a measurement)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="(test hook) every rank prints its launch environment as JSON and exits without touching the GPU")
    ap.add_argument("--force-dist", action="store_true", help="N=1: run the step through mfsr_dist_* (one-rank communicator) -- rehearsal")
    ap.add_argument("--spawn-ranks", action="store_true",
                    help="start the rank processes from this parent even at N = 1 (with --force-dist: the whole child-process path of "
                         "`bench.py --gpus N` -- spawn_ranks, environment, RCCL context, relayed line -- on a one-GPU box)")
    ap.add_argument("--fixed-halo", action="store_true", help="N > 1, stripes: keep the default 64-row raw halo of the exchange instead "
                    "of sizing it from the probe burst's measured flow")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the H2D->D2H end-to-end leg (median of 20 bursts)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the stand-alone warp+fuse leg (with --no-e2e --no-cpu-baseline a profile of the run then holds only "
                         "the bursts of the warm-up and the timed region: profiles/*_kernel_stats_timed_region.csv)")
    ap.add_argument("--cpu-sample-frames", type=int, default=4)
    ap.add_argument("--unfused", action="store_true", help="one launch per reference kernel (A/B against the fused path)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: run the RCCL exchange of burst i on the compute stream instead of overlapping it with the "
                         "align+fuse of burst i+1")
    ap.add_argument("--force-pipelined", action="store_true", help="use the two-context pipelined step loop even at N=1 (test)")
    ap.add_argument("--no-pair", action="store_true", help="cfg.pairFrames = 0: one warp+fuse launch per frame (the reference's structure)")
    ap.add_argument("--group", type=int, default=None, help="cfg.pairFrames = N: N frames per warp+fuse launch (2..4; default: as many as "
                    "one launch takes, 4 at x2 Bayer, else 2)")
    ap.add_argument("--async-fuse", action="store_true",
                    help="cfg.asyncFuse = 1: warp+fuse launches on the burst's own stream beside the alignment of the next group (the "
                         "default of rounds 2-3; measured 1 %% slower than one stream since the fuse kernel fills the CUs: A/B)")
    ap.add_argument("--no-async-fuse", action="store_true", help="(default since round 4: cfg.asyncFuse = 0) accepted for compatibility")
    ap.add_argument("--h2d", action="store_true",
                    help="N=1 only: frames start in pinned HOST memory and stream through a 4-deep device ring on a copy "
                         "stream (the PCIe-inclusive rate quoted in DESIGN.md; `value` of the contract is the HBM-resident run)")
    args = ap.parse_args()

    if args.dist_impl == "auto":
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.dry_run_ranks:
            # plain command: one process drives all N GPUs if they are there (counting devices does not initialise the GPU)
            try:
                import torch as _t
                n_vis = _t.cuda.device_count()
            except Exception:
                n_vis = 0
            args.dist_impl = "local" if n_vis >= args.gpus else "rccl"
        else:
            args.dist_impl = "rccl"
    local_group = args.dist_impl == "local" and args.gpus > 1 and "WORLD_SIZE" not in os.environ
    if (args.gpus > 1 or args.spawn_ranks) and "WORLD_SIZE" not in os.environ and not local_group:
        # plain `python3 bench.py --gpus N`: start the N ranks ourselves (before anything here touches the GPU)
        rc, out0 = spawn_ranks(args.gpus, sys.argv[1:])
        if rc != 0 and args.dist_impl == "rccl" and not args.dry_run_ranks and os.environ.get("MFSR_BENCH_NO_FALLBACK") != "1":
            # the one-process-per-GPU RCCL run failed: measure the same sharded burst with all ranks in ONE fresh process
            # (mfsr_dist_group_*, peer copies) rather than report nothing; the line says which transport ran and why
            print(f"bench.py: the RCCL run failed (rc {rc}); falling back to --dist-impl local in a fresh process", file=sys.stderr)
            import subprocess
            env = dict(os.environ, MFSR_BENCH_FALLBACK_FROM=f"rccl run failed with rc {rc}")
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--dist-impl", "local"], env=env,
                               stdout=subprocess.PIPE, text=True,
                               timeout=float(os.environ.get("MFSR_BENCH_RANK_TIMEOUT_S", "900")))
            rc, out0 = p.returncode, p.stdout
        sys.stdout.write(out0)
        sys.stdout.flush()
        raise SystemExit(rc)

    world = 1 if local_group else int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")) if not local_group else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not local_group else 0
    if args.dry_run_ranks:
        fail_rank = os.environ.get("MFSR_BENCH_FAIL_RANK")
        if fail_rank is not None and int(fail_rank) == rank:
            raise SystemExit(3)
        print(json.dumps({"dry_run": True, "rank": rank, "local_rank": local_rank, "world": world, "gpus": args.gpus,
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "child": os.environ.get("MFSR_BENCH_CHILD") == "1"}), flush=True)
        return

    import torch
    import torch.distributed as dist

    if world != args.gpus and not local_group:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # MFSR_DIST_BACKEND=gloo: rehearsal of the multi-rank schedule on a box with fewer GPUs than ranks (the
    # ranks share GPUs, collectives are staged through the host); the measured runs use RCCL ("nccl")
    backend = os.environ.get("MFSR_DIST_BACKEND", "nccl")
    if backend == "gloo" or args.virtual_ranks:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The C-ABI multi-GPU path (libmfsr_dist.so) owns its RCCL communicator; torch.distributed is only the launcher's
    # control plane there (unique-id broadcast, status / timing reductions, barriers) and runs on GLOO with host tensors, so
    # that a failing RCCL (it has never run with more than one rank before the driver's own 8-GPU run) cannot take the
    # control plane with it: the ranks then AGREE on the fallback below instead of dying one by one.
    ctl_gloo = world > 1 and backend != "gloo" and args.dist_impl == "rccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo" or ctl_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl_dev = torch.device("cpu") if ctl_gloo else dev     # where the control-plane tensors live

    from multi_frame_super_resolution_amd import distributed as mdist
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    from multi_frame_super_resolution_amd.synth import make_burst

    W, H, fpg, s, mono = WORKLOADS[args.workload]
    strong = not args.weak           # N > 1: the workload's burst is fixed and sharded (north_star: ">= 6x at 8 GPUs vs 1")
    n_ranks = args.gpus if local_group else world      # ranks of the burst (local group: all of them in this process)
    n_frames = fpg if strong else fpg * n_ranks
    cfg = default_config(W, H, n_frames, s, mono)
    cfg.fused = 0 if args.unfused else 1
    if args.no_pair:
        cfg.pairFrames = 0
    if args.group is not None:
        cfg.pairFrames = args.group
    if args.no_async_fuse:
        cfg.asyncFuse = 0
    if args.async_fuse:
        cfg.asyncFuse = 1
    if world == 1 and not local_group:
        # device slots for --h2d and the end-to-end leg (mfsr_burst_*_host); unused by the resident run.  16 slots: a whole
        # 16-frame burst uploads without waiting for a slot (11.2 ms per 4K burst incl. the download of the result, against
        # 14.9 / 15.6 ms with 4 / 8 slots; 8K x 8 is PCIe-bound at 19.2 ms whatever the depth,
        # profiles/r02_h2d_ring_sweep.txt); 32 (the maximum): the next burst's frames upload under this burst's tail when
        # bursts come back to back (end_to_end.back_to_back)
        cfg.uploadRing = int(os.environ.get("MFSR_UPLOAD_RING", "32"))
    exchange = "stripes" if args.exchange == "auto" else args.exchange
    dist_impl = "torch" if backend == "gloo" else args.dist_impl
    # --force-dist: the multi-GPU code path with a world of one rank (one-rank RCCL communicator): rehearsal of what the
    # driver's N > 1 runs execute, on a single GPU
    use_cabi_dist = (world > 1 or args.force_dist) and dist_impl == "rccl"
    pipe = None if (use_cabi_dist or local_group) else BurstPipeline(cfg, dev)
    if local_group:
        args.no_e2e = True
    if use_cabi_dist and world == 1:
        args.no_e2e = True

    # synthetic burst: one scene (same seed on every rank), this rank's frames only
    seed = 1234 + 2

    def rank_frames(r, device):
        # THE burst of the workload, whatever N: frame k (scene, shift, noise) depends on k only -- a rank renders the
        # frames it owns plus the reference out of the one random stream (make_burst(keep=...)), so the N = 1 and the
        # N = 8 lines process the same 16 frames and their out16 checksums can be compared for the bit-identity the
        # stripes mode claims
        own = mdist.frames_of_rank(n_frames, r, n_ranks)
        burst, _, _ = make_burst(W, H, n_frames, scale=s, mono=mono, seed=seed, device=device,
                                 keep=sorted(set(own) | {cfg.reference}))
        fr = {k: burst[k] for k in sorted(set(own) | {cfg.reference})}
        return own, fr

    mine, frames = rank_frames(rank, dev)
    torch.cuda.synchronize()

    # --dist-impl local: all ranks of the burst in this process (mfsr_dist_group_*: one worker thread per rank inside the
    # library, peer copies over xGMI in place of RCCL calls; the same per-rank code as the RCCL contexts)
    grp = None
    fallback_local = False     # set when the RCCL contexts could not be created and rank 0 took over with the in-process group
    if local_group:
        n_dev = torch.cuda.device_count()
        if not args.virtual_ranks and n_dev < n_ranks:
            raise SystemExit(f"--dist-impl local --gpus {n_ranks}: only {n_dev} device(s) visible (--virtual-ranks puts every rank "
                             "on device 0: a functional rehearsal)")
        g_devices = [0] * n_ranks if args.virtual_ranks else list(range(n_ranks))
        per_rank = [frames] + [rank_frames(r, torch.device("cuda", g_devices[r]))[1] for r in range(1, n_ranks)]
        grp = mdist.LocalGroup(cfg, g_devices)
        g_table = grp.frame_table(per_rank)
        g_mode = "stripes" if args.exchange == "auto" else args.exchange
        for d_i in sorted(set(g_devices)):
            torch.cuda.synchronize(d_i)

    # N > 1, C-ABI path: one mfsr_dist context per rank (RCCL communicator + burst context + per-frame product buffers);
    # a step is one mfsr_dist_process_burst on the current stream
    dctx = None
    if use_cabi_dist:
        from multi_frame_super_resolution_amd import capi
        D = capi.dist_lib()
        uid = torch.zeros(capi.DIST_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_uint8 * capi.DIST_ID_BYTES)()
            D.dist_get_unique_id(buf)
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        uid = uid.to(ctl_dev)
        if world > 1:
            dist.broadcast(uid, src=0)
        uid_c = (ctypes.c_uint8 * capi.DIST_ID_BYTES)(*uid.cpu().tolist())
        nbytes = D.dist_workspace_bytes(ctypes.byref(cfg), world)
        d_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        d_h = ctypes.c_void_p()
        created, why = 1, ""
        try:
            if os.environ.get("MFSR_BENCH_FAIL_RCCL") == "1":      # (test hook)
                raise capi.MfsrError("mfsr_dist_create", -5, "MFSR_BENCH_FAIL_RCCL")
            D.dist_create(ctypes.byref(d_h), ctypes.byref(cfg), rank, world, uid_c, (d_ws.data_ptr() + 255) // 256 * 256, nbytes)
        except capi.MfsrError as e:
            created, why = 0, str(e)
        if world > 1:
            ok = torch.tensor([created], dtype=torch.int32, device=ctl_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            all_created = int(ok.item()) == 1
        else:
            all_created = created == 1
        if not all_created:
            # RCCL did not come up on every rank.  Agreed fallback (unless MFSR_BENCH_NO_FALLBACK=1): the ranks release their
            # GPUs, rank 0 runs the same sharded burst with all ranks inside its own process (mfsr_dist_group_*, peer copies);
            # the line says which transport ran and why.
            msg = f"rank {rank}: mfsr_dist_create {'failed: ' + why if not created else 'ok, but a peer failed'}"
            print("bench.py: " + msg, file=sys.stderr)
            if created:
                D.dist_destroy(d_h)
            del d_ws
            if world == 1 or os.environ.get("MFSR_BENCH_NO_FALLBACK") == "1" or not ctl_gloo:
                raise SystemExit("bench.py: the RCCL contexts could not be created (" + msg + ")")
            if rank != 0:
                del frames
                torch.cuda.empty_cache()
                dist.barrier()           # rank 0's final barrier
                dist.destroy_process_group()
                return
            os.environ["MFSR_BENCH_FALLBACK_FROM"] = "mfsr_dist_create failed on a rank (RCCL): " + (why or "see stderr")
            use_cabi_dist, fallback_local = False, True
            n_dev = torch.cuda.device_count()
            if n_dev < n_ranks and not args.virtual_ranks:
                raise SystemExit(f"fallback to the in-process group needs {n_ranks} visible devices, found {n_dev}")
            g_devices = [0] * n_ranks if (args.virtual_ranks or n_dev < n_ranks) else list(range(n_ranks))
            per_rank = [frames] + [rank_frames(r, torch.device("cuda", g_devices[r]))[1] for r in range(1, n_ranks)]
            grp = mdist.LocalGroup(cfg, g_devices)
            g_table = grp.frame_table(per_rank)
            g_mode = exchange
            args.no_e2e = True
            for d_i in sorted(set(g_devices)):
                torch.cuda.synchronize(d_i)
        else:
            d_out16 = torch.empty(H * s, W * s, 3, dtype=torch.int16, device=dev) if rank == 0 else None
            d_status = torch.zeros(1, dtype=torch.int32, device=dev)
            d_ptrs = (ctypes.c_void_p * n_frames)(*[frames[k].data_ptr() if k in frames else None for k in range(n_frames)])
            d_mode = {"stripes": capi.DIST_STRIPES, "reduce": capi.DIST_REDUCE, "reduce_scatter": capi.DIST_REDUCE_SCATTER}[exchange]
            dctx = dict(D=D, h=d_h, burst=D.dist_burst(d_h))

    # torch.distributed mirror, accumulator-summing modes: steps are independent bursts, so the exchange (reduce-scatter,
    # stripe finish, gather) of burst i runs on a side stream while the compute stream already aligns and fuses burst i+1
    # into a second burst context (own workspace + accumulators): collectives overlap compute.
    pipelined = ((world > 1 and not args.no_overlap and not use_cabi_dist and exchange != "stripes" and grp is None) or args.force_pipelined)
    pipes = [pipe] if pipe is not None else []
    step_no = [0]
    stripe_bufs = None
    if world > 1 and not use_cabi_dist and exchange == "stripes" and grp is None:
        stripe_bufs = mdist.StripeBuffers(pipe, n_frames, rank, world)
    if pipelined:
        pipes.append(BurstPipeline(cfg, dev))
        side = torch.cuda.Stream(device=dev)
        ev_acc = [torch.cuda.Event() for _ in pipes]    # accumulate of the burst in context j done (compute stream)
        ev_done = [torch.cuda.Event() for _ in pipes]   # exchange + finish of context j done (side stream)
        used = [False for _ in pipes]

    h2d = args.h2d and world == 1 and pipe is not None
    host = None
    if world == 1 and pipe is not None:
        order = sorted(frames.keys())
        host = [frames[k].cpu().pin_memory() for k in order]   # pinned host copies for --h2d / the end-to-end leg

    def step_h2d():
        # frames start in pinned HOST memory; the library uploads them on its copy stream into a ring of device slots while
        # the compute stream works on earlier frames (mfsr_burst_*_host), and copies the u16 result back
        return pipe.process_host(host)

    def step():
        if h2d:
            return step_h2d()
        if grp is not None:
            grp.process(g_table, g_mode)
            return grp.out16
        if use_cabi_dist:
            D.dist_process_burst(d_h, d_ptrs, d_mode, d_out16.data_ptr() if d_out16 is not None else None, d_status.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            return d_out16
        if stripe_bufs is not None:
            return mdist.process_burst_stripes(pipe, frames, stripe_bufs, n_frames=n_frames)[0]
        if not pipelined:
            if world > 1:
                mdist.accumulate_local(pipe, frames, rank, world, n_frames)
                return mdist.exchange_and_finish(pipe, exchange)
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
                out = mdist.exchange_and_finish(p, exchange)
            else:
                _, out = p.finish(want_float=False, want_u16=True)
            ev_done[j].record(side)
        used[j] = True
        return out

    def barrier():
        if grp is not None:
            grp.synchronize()   # every rank's streams idle (the last burst's stripes are on rank 0)
            return
        if use_cabi_dist:   # the last burst's stripes are collected on the dist context's own stream
            D.dist_wait_output(d_h, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if h2d:
            pipe.host_sync()      # the last image's download runs on the burst's own stream

    # N > 1: a transport that hangs (it has never run with more than one rank before the driver's own run) must not hang the
    # launcher: if the bursts before the timed region do not complete, this rank exits non-zero -- torch.distributed.run then
    # ends the other ranks, and the plain command's parent (spawn_ranks) falls back to the in-process group
    watchdog = None
    if world > 1 or grp is not None:
        import threading
        hang_s = float(os.environ.get("MFSR_BENCH_HANG_S", "300"))

        def _hung():
            print(f"bench.py: rank {rank}: the bursts before the timed region did not complete within {hang_s:.0f} s "
                  f"(transport {'local group' if grp is not None else dist_impl}): giving up", file=sys.stderr, flush=True)
            os._exit(5)

        watchdog = threading.Timer(hang_s, _hung)
        watchdog.daemon = True
        watchdog.start()
    if os.environ.get("MFSR_BENCH_TEST_HANG") == "1" and rank == world - 1:      # (test hook: this rank never arrives)
        time.sleep(3600)
    if world > 1 and not ctl_gloo:  # (torch mirror) bring torch's RCCL communicator up outside the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, device=dev))
    for _ in range(args.warmup):
        step()
    barrier()
    halo_note = None
    if grp is not None and g_mode == "stripes":
        if args.warmup == 0:
            step()
            barrier()
        if any(int(t.item()) != 0 for t in grp.status):
            grp.set_raw_halo(H)
            halo_note = "whole raw frames exchanged (a flow exceeded the default 64-row halo)"
            step()
            barrier()
        elif not args.fixed_halo:
            # raw-row halo from the flow the probe burst measured (+ 3 rows of tap / rounding reach + 4 of margin) instead of
            # the default 64 rows; status 1 still guards every burst of the timed region
            v = grp.measured_flow()
            halo = max(8, (int(v + 0.999) + 3 + 4 + 3) // 4 * 4)
            if halo < 64:
                grp.set_raw_halo(halo)
                halo_note = f"raw halo {halo} rows from the measured vertical flow {v:.2f} px (default 64)"
                step()
                barrier()
    if use_cabi_dist and exchange == "stripes":
        # a vertical flow beyond the raw halo of the stripes exchange (status 1) invalidates the result: exchange whole raw
        # frames instead (always valid, ~2.5x the traffic); decided on a probe burst outside the timed region
        if args.warmup == 0:
            step()
            barrier()
        st = d_status.clone().to(ctl_dev)
        if world > 1:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        if int(st.item()) != 0:
            D.dist_set_raw_halo(d_h, H)
            halo_note = "whole raw frames exchanged (a flow exceeded the default 64-row halo)"
            step()
            barrier()
        elif not args.fixed_halo:
            v = ctypes.c_float(0.0)
            D.dist_measured_flow(d_h, ctypes.byref(v), torch.cuda.current_stream().cuda_stream)   # (already max-reduced over the ranks)
            halo = max(8, (int(v.value + 0.999) + 3 + 4 + 3) // 4 * 4)
            if halo < 64:
                D.dist_set_raw_halo(d_h, halo)
                halo_note = f"raw halo {halo} rows from the measured vertical flow {v.value:.2f} px (default 64)"
                step()
                barrier()
    if watchdog is not None:
        watchdog.cancel()
    from multi_frame_super_resolution_amd import capi as _capi
    LIB = _capi.lib()
    # (local group: rank 0's burst context is the one whose warp+fuse launches are event-timed)
    timed_bursts = [q._h for q in pipes]
    for hb in timed_bursts:
        LIB.burst_timing(hb, 1)
    if dctx:
        dctx["D"].dist_timing(dctx["h"], 1)   # both burst contexts of the rank (pipelined bursts alternate)
    if grp is not None:
        grp.timing(0, True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches, fused_frames = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
    for hb in timed_bursts:
        t_ms, n_l, n_f = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
        LIB.burst_timing_read(hb, ctypes.byref(t_ms), ctypes.byref(n_l), ctypes.byref(n_f))
        LIB.burst_timing(hb, 0)
        tot_ms.value += t_ms.value
        launches.value += n_l.value
        fused_frames.value += n_f.value
    if dctx:
        t_ms, n_l, n_f = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
        dctx["D"].dist_timing_read(dctx["h"], ctypes.byref(t_ms), ctypes.byref(n_l), ctypes.byref(n_f))
        dctx["D"].dist_timing(dctx["h"], 0)
        tot_ms.value, launches.value, fused_frames.value = t_ms.value, n_l.value, n_f.value
    if grp is not None:
        tot_ms.value, launches.value, fused_frames.value = grp.timing_read(0)
        grp.timing(0, False)
    # checksum of the last burst's u16 image (rank 0): N = 1 and N > 1 (stripes) lines of the same workload must agree
    out16_sha = None
    if rank == 0:
        import hashlib
        last = grp.out16 if grp is not None else (d_out16 if use_cabi_dist else (pipes[(step_no[0] - 1) % len(pipes)].out16 if pipes else None))
        if h2d:
            last = pipe._out16_host
        if last is not None:
            out16_sha = hashlib.sha256(last.cpu().numpy().tobytes()).hexdigest()[:16]

    if world > 1 and not fallback_local:
        t = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if grp is not None and any(int(t.item()) != 0 for t in grp.status):
        raise SystemExit("mfsr_dist: a frame's vertical flow exceeded the raw halo of the stripes exchange (status 1)")
    if use_cabi_dist and int(d_status.item()) != 0:
        raise SystemExit("mfsr_dist: a frame's vertical flow exceeded the raw halo of the stripes exchange (status 1): the result is "
                         "invalid; use --exchange reduce_scatter or a larger halo")

    # With cfg.asyncFuse (--async-fuse; not the default any more) the warp+fuse launches share the GPU with the alignment of the
    # following frames, which stretches them.  A short extra leg times the same launches WITHOUT that overlap (launches back to
    # back on one stream) so that the line also carries the kernel's stand-alone figure.
    isolated = None
    if world == 1 and rank == 0 and pipe is not None and cfg.asyncFuse and not h2d and not args.no_isolated:
        cfg_iso = default_config(W, H, n_frames, s, mono)
        cfg_iso.fused, cfg_iso.pairFrames, cfg_iso.asyncFuse = cfg.fused, cfg.pairFrames, 0
        p_iso = BurstPipeline(cfg_iso, dev)
        mdist.process_burst(p_iso, frames, n_frames=n_frames)
        torch.cuda.synchronize()
        LIB.burst_timing(p_iso._h, 1)
        for _ in range(3):
            mdist.process_burst(p_iso, frames, n_frames=n_frames)
        torch.cuda.synchronize()
        t_ms, n_l, n_f = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
        LIB.burst_timing_read(p_iso._h, ctypes.byref(t_ms), ctypes.byref(n_l), ctypes.byref(n_f))
        LIB.burst_timing(p_iso._h, 0)
        if n_l.value:
            isolated = {"avg_launch_ms": t_ms.value / n_l.value, "launches_timed": n_l.value}
        p_iso.close()

    # SURVEY.md section 8(d)'s end-to-end figure, beside the HBM-resident `value`: wall time from the first H2D enqueue
    # to the final D2H complete, one burst at a time, median of 20 (after 5 warm-up bursts)
    e2e = None
    if world == 1 and rank == 0 and not args.no_e2e:
        import statistics
        ts = []
        for i in range(5 + 20):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.process_host(host)
            pipe.host_sync()
            if i >= 5:
                ts.append(time.perf_counter() - t1)
        med = statistics.median(ts)
        # the same bursts back to back (a camera delivering burst after burst): burst i + 1 is enqueued while burst i's tail
        # is still being fused and downloaded -- its uploads take the ring slots as burst i's fuse frees them, the results
        # land alternately in two pinned host images; host -> host rate of the stream of bursts
        outs = [torch.empty(H * s, W * s, 3, dtype=torch.int16).pin_memory() for _ in range(2)]
        n_stream = 20
        for i in range(3):
            pipe.process_host(host, outs[i & 1])
        pipe.host_sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_stream):
            pipe.process_host(host, outs[i & 1])
        pipe.host_sync()
        torch.cuda.synchronize()
        stream_ms = (time.perf_counter() - t1) / n_stream * 1e3
        same = bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], pipe._out16_host))
        e2e = {
            "value": round(n_frames * W * H / med / 1e6, 2), "unit": "Mpix/s", "ms_median": round(med * 1e3, 3),
            "ms_min": round(min(ts) * 1e3, 3), "bursts": len(ts),
            "includes": f"H2D of {n_frames} raw frames ({n_frames * W * H * 2 / 1e6:.0f} MB, pinned host memory, library copy "
                        f"stream + {cfg.uploadRing}-slot device ring) and D2H of the u16 HR image ({s * s * W * H * 6 / 1e6:.0f} MB); "
                        "one burst in flight",
            "back_to_back": {"ms_per_burst": round(stream_ms, 3), "value": round(n_frames * W * H / (stream_ms * 1e-3) / 1e6, 2),
                             "unit": "Mpix/s", "bursts": n_stream, "images_identical_to_single_burst": same,
                             "note": "host -> host, bursts enqueued back to back (no host synchronisation between them): "
                                     "uploads of burst i+1 overlap the fuse tail and the download of burst i"},
        }

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_frames * W * H * args.steps / dt / 1e6
        # bytes a launch MUST move (accumulators once per launch, not once per frame) vs the reference-structure figure
        # frames fused per rank and the fraction of the HR rows a launch covers: all frames on 1/world of the rows in the
        # stripes mode, the rank's own frames on the whole grid otherwise
        stripes_mode = (n_ranks > 1 or use_cabi_dist) and exchange == "stripes"
        fused_per_rank = n_frames if stripes_mode else len(mine)
        row_frac = 1.0
        if stripes_mode:
            pl = _capi.StripePlan()
            LIB.dist_stripe_plan(ctypes.byref(cfg), n_ranks, 0, 64, ctypes.byref(pl))
            row_frac = (pl.rowEnd - pl.rowBegin) / float(H * s)
        n_launch_burst, bytes_burst = burst_fuse_bytes(W, H, s, mono, fused_per_rank, int(LIB.raw["mfsr_burst_group_size"](ctypes.byref(cfg))))
        bytes_launch = bytes_burst / n_launch_burst * row_frac
        frames_per_launch = fused_frames.value / max(launches.value, 1)
        bytes_ref_launch = fuse_bytes_reference_structure(W, H, s, mono) * frames_per_launch * row_frac
        k_ms = tot_ms.value / max(launches.value, 1)
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        achieved_ref = bytes_ref_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, valu, traffic_note = None, None, None
        tpath = os.path.join(ROOT, "profiles", "fuse_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # (the 64-frame 8K burst launches the very kernel instance on the very grid of its 8-frame share)
                ent = tj.get(args.workload) or (tj.get("8k8_rggb_x2") if args.workload == "8k64_rggb_x2" else None)
                # the counters describe one state of the kernel sources: an entry taken from another state is not reported
                import hashlib
                hsh = hashlib.sha256()
                for f in ("accumulate_fast.hip", "accumulate_common.hpp"):
                    hsh.update(open(os.path.join(ROOT, "multi_frame_super_resolution_amd", "csrc", f), "rb").read())
                if isinstance(ent, dict) and ent.get("kernel_source_sha16") != hsh.hexdigest()[:16]:
                    traffic_note = ("profiles/fuse_traffic.json was measured on another state of accumulate_fast.hip / accumulate_common.hpp "
                                    f"(stamp {ent.get('kernel_source_sha16')}, sources {hsh.hexdigest()[:16]}): traffic and the VALU floor are not "
                                    "reported; refresh with tools/gpu_pmc_workloads.sh")
                    ent = None
                # the counters were taken on whole-frame launches of one grouping: they say nothing about a stripe-sharded
                # run (row_frac < 1) or another --group
                if isinstance(ent, dict) and (row_frac != 1.0 or (ent.get("frames_per_launch") is not None and
                                                                  abs(ent["frames_per_launch"] - frames_per_launch) > 0.01)):
                    ent = None
                if isinstance(ent, dict):
                    traffic = ent.get("hbm_bytes_per_launch")
                    vi = ent.get("valu_wave_insts_per_launch")
                    if vi:
                        # VALU-issue ceiling: every wave-level VALU instruction occupies its SIMD for 4 cycles (wave64 on a
                        # 16-lane SIMD); 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock
                        # nominal: 4 cycles per wave-instruction.  Measured (tools/ubench/valu_ops.hip): 2.8 cycles for the
                        # fast class, 4.4 for the slow one, 8.3 for transcendentals -- the static mix of this kernel
                        # (tools/valu_mix.py) prices its stream at cpw cycles per instruction: the floor a launch can reach
                        cpw = ent.get("cycles_per_inst_weighted")
                        nominal_ms = vi * 4 / (1024 * VALU_CLOCK_GHZ * 1e9) * 1e3
                        floor_ms = vi * (cpw or 4) / (1024 * VALU_CLOCK_GHZ * 1e9) * 1e3
                        valu = {"wave_insts_per_launch": vi, "cycles_per_inst": cpw or 4, "simds": 1024, "clock_ghz": VALU_CLOCK_GHZ,
                                "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / k_ms, 4) if k_ms > 0 else None,
                                "cycles_per_inst_note": ("class-weighted issue cost of the kernel's static instruction mix "
                                                         f"({ent.get('valu_mix_static')}; fast 2.8 / slow 4.4 / transcendental 8.3 cycles, "
                                                         "tools/ubench/valu_ops.hip)") if cpw else "nominal 4 cycles (no instruction mix recorded)",
                                "nominal_4_cycles_floor_ms": round(nominal_ms, 4),   # (not a floor: the fast class issues at 2.8 cycles)
                                # every instruction at the fastest class's cost: a bound no mix can beat (the weighted figure above
                                # prices a STATIC mix and is good to a few percent)
                                "all_fast_class_floor_ms": round(vi * 2.8 / (1024 * VALU_CLOCK_GHZ * 1e9) * 1e3, 4),
                                "source": ent.get("source")}
                        if k_ms > 0 and floor_ms > k_ms:
                            # a floor above the measurement is refuted by it: the launch executes a lighter mix than the static
                            # one (x4: the saturated / branchy paths).  The fraction is then taken against the bound no mix can
                            # beat, every instruction at the fast class's cost, and the estimate is kept beside it.
                            valu["static_mix_estimate_ms"] = valu["floor_ms"]
                            valu["floor_ms"] = valu["all_fast_class_floor_ms"]
                            valu["cycles_per_inst"] = 2.8
                            valu["frac"] = round(valu["floor_ms"] / k_ms, 4)
                            valu["floor_note"] = ("the class-weighted estimate of the STATIC mix lies above the measured launch, i.e. the "
                                                  "executed mix is lighter: floor_ms / frac are the all-fast-class bound (2.8 cycles per "
                                                  "instruction), the estimate is static_mix_estimate_ms")
                elif ent is not None:
                    traffic = ent
            except Exception:
                traffic = None
        line = {
            "metric": "Mpix/s end-to-end (N-frame burst -> x2 SR)" if s == 2 else f"Mpix/s end-to-end (N-frame burst -> x{s} SR)",
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "transport": (None if n_ranks == 1 and not use_cabi_dist else
                          (("local" if grp is not None else ("rccl" if use_cabi_dist else f"torch.distributed ({backend})"))
                           + (f" (fallback: {os.environ['MFSR_BENCH_FALLBACK_FROM']})" if os.environ.get("MFSR_BENCH_FALLBACK_FROM") else ""))),
            "out16_sha256_16": out16_sha,
            "data": "synthetic" + (", streamed from pinned host memory (library copy stream, device ring)" if h2d else
                                   ", frames resident in HBM"),
            "config": {
                "workload": f"{n_frames}-frame {W}x{H} {'gray' if mono else 'RGGB u16'} burst -> x{s} "
                            f"({args.workload}; BASELINE configs[{ {'4k16_rggb_x2': 2, '1080p5_gray_x2': 1, '4k16_rggb_x4': 3, '8k8_rggb_x2': 4, '8k64_rggb_x2': 4}[args.workload] }])",
                "frames_per_gpu": len(mine),
                "burst_frames": n_frames,
                "output_mpix_per_s": round(s * s * W * H * args.steps / dt / 1e6, 2),
                "parallelism": "1 GPU" if n_ranks == 1 else (
                    (f"align frame-sharded x{n_ranks}, p2p exchange of the LR products (one packed message of raw/flow/certainty rows "
                     f"per peer), fuse sharded over {n_ranks} HR row stripes, u16 stripes gathered on rank 0 (bit-identical to 1 GPU)"
                     if exchange == "stripes" else
                     f"frame-shard x{n_ranks} + {exchange} of the HR accumulators"
                     + (" overlapped with the next burst's compute" if pipelined else ""))
                    + (", libmfsr_dist.so, one process per GPU (RCCL directly)" if use_cabi_dist else
                       (", libmfsr_dist.so, all ranks in one process (mfsr_dist_group: one thread per rank, peer copies)"
                        + (", ALL RANKS ON DEVICE 0: rehearsal, not a measurement" if (args.virtual_ranks or (grp is not None and len(set(grp.devices)) < n_ranks)) else "")
                        if grp is not None else ", torch.distributed mirror"))
                    + (f"; {halo_note}" if halo_note else "")
                    + (f"; fallback: {os.environ['MFSR_BENCH_FALLBACK_FROM']}" if os.environ.get("MFSR_BENCH_FALLBACK_FROM") else "")),
                "kernels": "unfused (one launch per reference kernel)" if args.unfused else "fused",
                **({"rehearsal": "gloo backend, ranks share GPUs, collectives staged through the host: not a measurement"}
                   if (world > 1 and backend == "gloo") else {}),
                **({"exchange_per_rank": [dict(zip(("messages_sent", "bytes_sent"), grp.exchange_stats(r))) for r in range(n_ranks)]}
                   if grp is not None else {}),
            },
            "end_to_end": e2e,
            "roofline": {
                "kernel": ("k_accumulate2xTile / k_accumulate2xStrip" if s == 2 else "k_accumulate4xTile")
                          + " (warp+fuse, accumulateSuperResFullN; a launch = the tile kernel + the margin kernel of each frame it fuses)",
                # the kernel sits at its VALU-issue ceiling, not at the HBM one (profiles/: SQ_INSTS_VALU x 4 cycles fills
                # the launch time); the HBM fraction below is on the bytes a launch has to move
                "bound": "valu",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                **({"traffic_note": traffic_note} if traffic_note else {}),
                "bytes_per_launch": int(bytes_launch),
                "bytes_note": "algorithmic bytes a launch must move: both accumulator plane-sets once (HR*48 B; HR*24 B on the "
                              "first launch of a burst, which overwrites them), the kernel-parameter field once, raw + flow + "
                              "certainty of each fused frame; mean over the launches of a burst",
                "frames_per_launch": round(frames_per_launch, 3),
                "avg_launch_ms": round(k_ms, 4),
                "launches_timed": launches.value,
                "valu": valu,
                "timed_region_note": ("launches of the timed region run on the burst's own stream concurrently with the alignment of the "
                                      "following frames (cfg.asyncFuse, +5 % burst throughput), which stretches them; `isolated` = the "
                                      "same launches back to back on one stream (3 extra bursts, cfg.asyncFuse = 0)") if isolated else None,
                "isolated": ({"avg_launch_ms": round(isolated["avg_launch_ms"], 4), "launches_timed": isolated["launches_timed"],
                              "achieved": round(bytes_launch / (isolated["avg_launch_ms"] * 1e-3) / 1e9, 1),
                              "frac": round(bytes_launch / (isolated["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              # (against the same floor as valu.frac; the all-fast-class bound if the isolated launch beats the
                              # static-mix estimate)
                              "valu_frac": (round((valu["floor_ms"] if valu["floor_ms"] <= isolated["avg_launch_ms"]
                                                   else valu["all_fast_class_floor_ms"]) / isolated["avg_launch_ms"], 4) if valu else None)}
                             if isolated else None),
                "reference_structure": {
                    "bytes_per_launch": int(bytes_ref_launch), "achieved": round(achieved_ref, 1),
                    "frac": round(achieved_ref / HBM_PEAK_GBPS, 4),
                    "note": "SURVEY.md 8(d) reference-structure bytes (one launch per frame, accumulators RMW per frame) x frames "
                            "per launch: what the reference's kernel would move for the same frames, NOT what this kernel moves",
                },
            },
        }
        if not args.no_cpu_baseline and n_ranks == 1:
            line["cpu_baseline"] = cpu_baseline(W, H, s, mono, args.cpu_sample_frames, seed)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    for q in pipes:
        q.close()
    if grp is not None:
        grp.close()
    if dctx:
        dctx["D"].dist_destroy(dctx["h"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
