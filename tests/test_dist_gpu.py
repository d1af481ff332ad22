"""Multi-GPU paths on the one-GPU box (-m gpu):

  * the stripe building blocks (mfsr_burst_align_frame / mfsr_burst_fuse_rows / mfsr_burst_finish_rows +
    mfsr_dist_stripe_plan): every "virtual rank" of a world of 2, 3 or 5 fuses its HR stripe from buffers that hold ONLY
    the rows its plan says it receives (everything else poisoned) -- the assembled image is bit-identical to the
    single-GPU burst;
  * the torch.distributed mirror (distributed.py) with two real processes sharing the GPU over gloo (collectives staged
    through the host: a functional test, not a measurement) driving the real BurstPipeline;
  * the RCCL layer itself (libmfsr_dist.so, include/mfsr_dist.h) with a communicator of one rank: ncclCommInitRank, the
    three modes, status flag.  Two RCCL ranks cannot share one GPU, so N > 1 over RCCL is only exercised by bench.py on
    the driver's multi-GPU node.
"""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _burst(W, H, N, scale, mono, seed=41, max_shift=4.0):
    from multi_frame_super_resolution_amd.synth import make_burst
    frames, _, _ = make_burst(W, H, N, scale=scale, mono=mono, seed=seed, max_shift=max_shift)
    return frames


@pytest.mark.parametrize("W,H,N,scale,mono,worlds", [(384, 256, 5, 2, False, (2, 3)), (328, 200, 4, 4, False, (3,)),
                                                      (256, 192, 3, 2, True, (2, 5)), (392, 264, 4, 3, False, (2,))])
def test_virtual_rank_stripes_equal_single_gpu(W, H, N, scale, mono, worlds):
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, mono)]
    cfg = default_config(W, H, N, scale, mono)
    cfg.reference = 1
    single = BurstPipeline(cfg, dev)
    _, want = single.process(frames)
    want = want.clone()
    single.close()

    pipe = BurstPipeline(cfg, dev)
    pipe.set_reference(frames[cfg.reference])
    prods = []
    for k in range(N):
        f, m = pipe.new_frame_products()
        pipe.align_frame(frames[k], k == cfg.reference, f, m)
        prods.append((f, m))
    # raw halo from the flows themselves (rows a stripe reads reach |v| + 3 raw rows past it); wild border flows included
    vmax = max(float(f[..., 1].abs().max()) for f, _ in prods)
    halo = max(8, int(np.ceil(vmax)) + 4)
    print(f"max |flow_y| {vmax:.2f} raw px -> halo {halo}")
    for world in worlds:
        got = torch.zeros_like(want)
        for r in range(world):
            pl = pipe.stripe_plan(world, r, halo)
            if pl.rowEnd <= pl.rowBegin:
                continue
            # what rank r holds: only the plan's rows of every frame's products, the rest poisoned
            raws, flows, masks = [], [], []
            for k in range(N):
                raw = torch.full_like(frames[k], -1)                                 # 0xFFFF
                raw[pl.rawRow0:pl.rawRow0 + pl.rawRows] = frames[k][pl.rawRow0:pl.rawRow0 + pl.rawRows]
                f = torch.full_like(prods[k][0], float("nan"))
                f[pl.flowRow0:pl.flowRow0 + pl.flowRows] = prods[k][0][pl.flowRow0:pl.flowRow0 + pl.flowRows]
                m = torch.full_like(prods[k][1], float("nan"))
                m[pl.maskRow0:pl.maskRow0 + pl.maskRows] = prods[k][1][pl.maskRow0:pl.maskRow0 + pl.maskRows]
                raws.append(raw)
                flows.append(f)
                masks.append(m)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            for k in range(N):
                pipe.check_flow_bound(flows[k][pl.flowRow0:pl.flowRow0 + pl.flowRows], float(pl.maxFlowY), flag)
            assert int(flag.item()) == 0
            pipe._img_out.fill_(float("nan"))       # fresh mode must not read them
            pipe._total_weights.fill_(float("nan"))
            per = pipe.group_size()   # the single-GPU burst's grouping: same sums in the same order
            for k in range(0, N, per):
                ks = list(range(k, min(k + per, N)))
                pipe.fuse_rows([raws[j] for j in ks], [flows[j] for j in ks], [masks[j] for j in ks], pl.rowBegin, pl.rowEnd, k == 0)
            out = pipe.finish_rows(pl.rowBegin, pl.rowEnd - pl.rowBegin)
            got[pl.rowBegin:pl.rowEnd] = out[pl.rowBegin:pl.rowEnd]
        assert torch.equal(got, want), f"world {world}"
    # the flow bound check fires when a flow exceeds what the halo covers
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = prods[0][0].clone()
    bad[3, 5, 1] = 1000.0
    pipe.check_flow_bound(bad, 61.0, flag)
    assert int(flag.item()) == 1
    pipe.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, mode, W, H, N, scale, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multi_frame_super_resolution_amd import distributed as mdist
        from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        frames = _burst(W, H, N, scale, False)
        cfg = default_config(W, H, N, scale, False)
        pipe = BurstPipeline(cfg, dev)
        mine = mdist.frames_of_rank(N, rank, world)
        local = {k: frames[k].to(dev) for k in mine}
        local[cfg.reference] = frames[cfg.reference].to(dev)
        for rep in range(2):     # second burst on the same context and buffers
            if mode == "stripes":
                out, flag = mdist.process_burst_stripes(pipe, local, n_frames=N)
                assert int(flag.item()) == 0
            else:
                out = mdist.process_burst(pipe, local, mode=mode, n_frames=N)
            torch.cuda.synchronize()
        if rank == 0:
            q.put(out.cpu().numpy().view(np.uint16).copy())
        else:
            assert out is None
        pipe.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,scale", [("stripes", 2), ("stripes", 4), ("reduce_scatter", 2), ("reduce", 2)])
def test_two_processes_on_one_gpu_equal_single_gpu(mode, scale):
    import torch.multiprocessing as mp
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    W, H, N = 320, 256, 5
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False)]
    single = BurstPipeline(default_config(W, H, N, scale, False), dev)
    _, want = single.process(frames)
    want = want.cpu().numpy().view(np.uint16).copy()
    single.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, mode, W, H, N, scale, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if mode == "stripes":
        assert np.array_equal(got, want)          # same summation order as one GPU: bit-identical
    else:
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 2 and np.mean(d > 0) < 0.05


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_rccl_layer_with_one_rank(mode):
    """libmfsr_dist.so end to end with a one-rank RCCL communicator: get_unique_id, create (ncclCommInitRank), the three
    modes, status, destroy.  Equal to the plain burst bit for bit (one rank: the same launches in the same order)."""
    from multi_frame_super_resolution_amd import capi
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    W, H, N, scale = 320, 256, 5, 2
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False)]
    cfg = default_config(W, H, N, scale, False)
    single = BurstPipeline(cfg, dev)
    _, want = single.process(frames)
    want = want.clone()
    single.close()
    D = capi.dist_lib()
    uid = (ctypes.c_uint8 * capi.DIST_ID_BYTES)()
    D.dist_get_unique_id(uid)
    nbytes = D.dist_workspace_bytes(ctypes.byref(cfg), 1)
    assert nbytes > 0
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) // 256 * 256
    h = ctypes.c_void_p()
    D.dist_create(ctypes.byref(h), ctypes.byref(cfg), 0, 1, uid, base, nbytes)
    r0, r1 = ctypes.c_int(), ctypes.c_int()
    D.dist_stripe(h, 0, ctypes.byref(r0), ctypes.byref(r1))
    assert (r0.value, r1.value) == (0, H * scale)
    out16 = torch.zeros(H * scale, W * scale, 3, dtype=torch.int16, device=dev)
    status = torch.full((1,), 7, dtype=torch.int32, device=dev)
    ptrs = (ctypes.c_void_p * N)(*[f.data_ptr() for f in frames])
    for rep in range(2):
        D.dist_process_burst(h, ptrs, mode, out16.data_ptr(), status.data_ptr(), torch.cuda.current_stream().cuda_stream)
        D.dist_wait_output(h, torch.cuda.current_stream().cuda_stream)   # stripes: collection runs on the context's comm stream
        torch.cuda.current_stream().synchronize()
        assert int(status.item()) == 0
        assert torch.equal(out16, want), (mode, rep)
        out16.zero_()
    D.dist_destroy(h)


def test_bench_child_process_path_with_a_one_rank_rccl_context():
    """`bench.py --gpus N` as a plain command = spawn_ranks -> child processes with the launcher environment -> RCCL unique id
    broadcast -> mfsr_dist_create -> process_burst / wait_output -> rank 0's line relayed by the parent.  The whole path with
    N = 1 (`--spawn-ranks --force-dist`): rc 0, one JSON line, transport rccl, and the image checksum of the plain one-GPU
    run of the same workload (the stripes mode is bit-identical)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "1080p5_gray_x2", "--no-cpu-baseline", "--no-e2e",
              "--no-isolated"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + ["--force-dist", "--spawn-ranks"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["transport"] == "rccl" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, env=env, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    e = json.loads([l for l in q.stdout.splitlines() if l.strip().startswith("{")][0])
    assert e["transport"] is None
    assert d["out16_sha256_16"] == e["out16_sha256_16"] and d["out16_sha256_16"]


def test_bench_ranks_agree_on_the_in_process_fallback_when_rccl_does_not_come_up():
    """The driver's N > 1 command is the first run of the RCCL back end with more than one rank.  If mfsr_dist_create fails
    on any rank (here: forced on every rank, MFSR_BENCH_FAIL_RCCL=1), the ranks agree over the gloo control plane, the
    others release their GPUs and park at the final barrier, and rank 0 measures the same sharded burst with all ranks
    inside its own process (mfsr_dist_group_*): rc 0, one line, transport 'local (fallback: ...)', and -- the stripes mode
    being bit-identical -- the checksum of the one-GPU run.  (Two ranks on the one device: --virtual-ranks, a rehearsal.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--workload", "1080p5_gray_x2", "--no-cpu-baseline", "--no-e2e", "--no-isolated"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-impl", "rccl", "--virtual-ranks"] + common,
                       env=dict(env, MFSR_BENCH_FAIL_RCCL="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    # the IN-RUN agreement, not the parent's second attempt in a fresh process (which would say "rccl run failed with rc ...")
    assert d["transport"].startswith("local (fallback: mfsr_dist_create failed") and d["n_gpus"] == 2 and d["value"] > 0, d["transport"]
    assert "rehearsal" in d["config"]["parallelism"]
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env, capture_output=True, text=True,
                       timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    e = json.loads([l for l in q.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["out16_sha256_16"] == e["out16_sha256_16"] and e["out16_sha256_16"]


def test_bench_rank_that_never_arrives_ends_the_run_instead_of_hanging_it():
    """A transport that hangs must not hang the launcher: the bursts before the timed region run under a watchdog
    (MFSR_BENCH_HANG_S, default 300 s).  Here rank 1 of a two-rank gloo rehearsal never arrives (MFSR_BENCH_TEST_HANG=1): both
    ranks give up after the limit, the parent returns non-zero with the reason on stderr, and no line is printed."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(MFSR_DIST_BACKEND="gloo", MFSR_BENCH_TEST_HANG="1", MFSR_BENCH_HANG_S="8", MFSR_BENCH_NO_FALLBACK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-impl", "rccl", "--steps", "1", "--warmup", "1",
                        "--workload", "1080p5_gray_x2", "--no-cpu-baseline", "--no-e2e", "--no-isolated"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode != 0
    assert "did not complete within 8 s" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert time.time() - t0 < 300
