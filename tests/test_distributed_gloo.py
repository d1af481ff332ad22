"""world_size-2 gloo test (CPU) of the frame-sharded multi-GPU path
(multi_frame_super_resolution_amd/distributed.py): frame sharding, the two
exchange modes and the rank-0 gather, driven with a CPU stand-in for the HIP
pipeline (the oracle, test infrastructure) so that it runs without a GPU.

Checks: every frame is accumulated exactly once across ranks; the reduced result
equals the single-process result to fp32 rounding (the sum order differs);
non-root ranks return None.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H, N, S = 128, 96, 5, 2


class OraclePipe:
    """Duck-typed stand-in for BurstPipeline backed by the CPU oracle."""

    def __init__(self, cfg):
        from oracle.pipeline import OraclePipeline
        self.cfg = cfg
        self.op = OraclePipeline(cfg)
        self.img_out = torch.zeros(H * S, W * S, 3)
        self.total_weights = torch.zeros(H * S, W * S, 3)
        self.out16 = torch.zeros(H * S, W * S, 3, dtype=torch.int16)
        self.added = []

    def reset_accumulators(self):
        self.img_out.zero_()
        self.total_weights.zero_()
        self.added = []

    def set_reference(self, raw):
        self.op.set_reference(raw.numpy().view(np.uint16))

    def add_frame(self, raw, is_reference=False):
        self.added.append(bool(is_reference))
        self.op.add_frame(raw.numpy().view(np.uint16), is_reference, self.img_out.numpy(), self.total_weights.numpy())

    def finish(self, want_float=True, want_u16=True):
        out, q = self.op.finish(self.img_out.numpy(), self.total_weights.numpy())
        self.out16.copy_(torch.from_numpy(q.view(np.int16)))
        return torch.from_numpy(out), self.out16

    def finish_rows(self, row0, rows):
        _, q = self.op.finish(self.img_out.numpy(), self.total_weights.numpy())
        self.out16[row0:row0 + rows].copy_(torch.from_numpy(q.view(np.int16))[row0:row0 + rows])
        return self.out16


class OracleStripePipe(OraclePipe):
    """Stand-in with the stripe building blocks (align_frame / fuse_rows / finish_rows / stripe_plan) on the oracle."""

    device = torch.device("cpu")

    def new_frame_products(self):
        o = self.op
        return torch.full((o.th, o.tw, 2), float("nan")), torch.full((o.hh, o.hw, 4), float("nan"))

    def align_frame(self, raw, is_reference, flow, mask):
        scratch_a = np.zeros((H * S, W * S, 3), np.float32)
        scratch_w = np.zeros_like(scratch_a)
        self.op.add_frame(raw.numpy().view(np.uint16), is_reference, scratch_a, scratch_w)   # alignment (+ an unused fuse)
        flow.copy_(torch.from_numpy(self.op.flow))
        mask.copy_(torch.from_numpy(self.op.mask))
        self.added.append(bool(is_reference))

    def fuse_rows(self, raws, flows, masks, row_begin, row_end, fresh):
        o = self.op
        if fresh:
            self.img_out[row_begin:row_end] = 0
            self.total_weights[row_begin:row_end] = 0
        for raw, flow, mask in zip(raws, flows, masks):
            # the oracle fuses the whole frame; rows outside the stripe see the garbage of rows never received and are dropped
            a = self.img_out.numpy().copy()
            w = self.total_weights.numpy().copy()
            f, m = flow.numpy(), mask.numpy()
            with np.errstate(all="ignore"):
                o.o.set_cfa_pattern(o.cfa)
                o.o.accumulateSuperResFull(raw.numpy().view(np.uint16), a, w, m, o.kparam4, int(o.kparam4.strides[0]), o.tw, o.th,
                                           f, int(f.strides[0]), o.tw, o.th, o.white, o.black, o.W, o.H, S, int(a.strides[0]),
                                           int(m.strides[0]))
            self.img_out[row_begin:row_end] = torch.from_numpy(a[row_begin:row_end])
            self.total_weights[row_begin:row_end] = torch.from_numpy(w[row_begin:row_end])

    def group_size(self):
        import ctypes
        from multi_frame_super_resolution_amd import capi
        return int(capi.lib().raw["mfsr_burst_group_size"](ctypes.byref(self.cfg)))

    def check_flow_bound(self, flow_rows, bound, flag):
        if not bool((flow_rows[..., 1].abs() <= bound).all()):
            flag |= 1

    def stripe_plan(self, world, rank, raw_halo=64):
        import ctypes
        from multi_frame_super_resolution_amd import capi
        plan = capi.StripePlan()
        assert capi.lib().raw["mfsr_dist_stripe_plan"](ctypes.byref(self.cfg), world, rank, raw_halo, ctypes.byref(plan)) == 0
        return plan


def _cfg():
    import ctypes
    from multi_frame_super_resolution_amd import capi
    cfg = capi.Config()
    assert capi.lib().raw["mfsr_config_default"](ctypes.byref(cfg), W, H, N, S, 0) == 0
    cfg.levels = 1
    cfg.levelFactor[0] = 1
    cfg.tileSize[0] = 16
    cfg.maxShift[0] = 4
    cfg.lkIterations = 1
    return cfg


def _frames():
    from multi_frame_super_resolution_amd.synth import make_burst
    frames, _, _ = make_burst(W, H, N, scale=S, mono=False, seed=11, max_shift=2.0)
    return frames


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multi_frame_super_resolution_amd import distributed as mdist
        pipe = OracleStripePipe(_cfg()) if mode == "stripes" else OraclePipe(_cfg())
        frames = _frames()
        # a rank keeps only its shard + the reference resident
        mine = mdist.frames_of_rank(N, rank, world)
        local = {k: frames[k] for k in mine}
        local[0] = frames[0]
        if mode == "stripes":
            out, flag = mdist.process_burst_stripes(pipe, local, n_frames=N, raw_halo=8)
            assert int(flag.item()) == 0
        else:
            out = mdist.process_burst(pipe, local, mode=mode, n_frames=N)
        added = torch.tensor([len(pipe.added), sum(pipe.added)], dtype=torch.int64)
        dist.all_reduce(added)
        if rank == 0:
            q.put((out.numpy().view(np.uint16).copy(), added.tolist()))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["reduce", "reduce_scatter", "auto", "stripes"])
def test_two_ranks_equal_single_process(mode):
    from multi_frame_super_resolution_amd import distributed as mdist
    # single process reference
    pipe = OraclePipe(_cfg())
    single = mdist.process_burst(pipe, _frames()).numpy().view(np.uint16).copy()
    assert len(pipe.added) == N and sum(pipe.added) == 1

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, added = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert added == [N, 1]          # every frame exactly once, the reference exactly once
    if mode == "stripes":
        # stripes fuse every frame in frame order on each rank's rows: the single-process summation order, bit for bit
        # (the rows a rank never received stay NaN in its buffers, so a read outside the plan's ranges would show)
        assert np.array_equal(got, single)
        return
    d = np.abs(got.astype(np.int64) - single.astype(np.int64))
    # fp32 sum order differs between the sharded and the sequential accumulation
    assert d.max() <= 2 and np.mean(d > 0) < 0.05


def test_frame_shards_partition_the_burst():
    from multi_frame_super_resolution_amd.distributed import frames_of_rank
    for n in (1, 5, 16, 64):
        for world in (1, 2, 3, 8):
            all_frames = sorted(k for r in range(world) for k in frames_of_rank(n, r, world))
            assert all_frames == list(range(n))
            sizes = [len(frames_of_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
