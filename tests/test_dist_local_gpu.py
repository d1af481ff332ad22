"""csrc/dist.cpp's multi-rank code on the one-GPU box (-m gpu): `mfsr_dist_group_*` runs G "virtual ranks" on the same
device -- one worker thread and one stream per rank, peer copies in place of RCCL calls -- through the SAME
process_stripes / exchange_rows / gather_stripes / process_reduce code the RCCL contexts execute (only the transport
table differs).  What this covers that a one-rank communicator cannot: the packed send / receive ranges built from the
PEERS' stripe plans vs the receive ranges built from the own plan, non-root staging images, the two-slot status flag and
its all-reduce, gatherPending across bursts and mode switches, empty stripes, the halo-exceeded status path.

Every per-rank workspace is poisoned before the first burst (raw 0xFFFF, floats NaN): a row that was not exchanged but is
read would show in the result.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _burst(W, H, N, scale, mono, seed=41, max_shift=4.0):
    from multi_frame_super_resolution_amd.synth import make_burst
    frames, _, _ = make_burst(W, H, N, scale=scale, mono=mono, seed=seed, max_shift=max_shift)
    return frames


def _single(cfg, frames, dev):
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    p = BurstPipeline(cfg, dev)
    _, want = p.process(frames)
    want = want.clone()
    p.close()
    return want


def _group(cfg, G, frames, dev):
    from multi_frame_super_resolution_amd.distributed import LocalGroup, frames_of_rank
    grp = LocalGroup(cfg, [dev.index or 0] * G)
    for w in grp._ws:                      # poison: NaN floats / 0xFFFF raw everywhere
        w.fill_(0xFF)
    torch.cuda.synchronize()
    per_rank = []
    for r in range(G):
        own = {k: frames[k] for k in frames_of_rank(cfg.frames, r, G)}
        own[cfg.reference] = frames[cfg.reference]
        per_rank.append(own)
    return grp, grp.frame_table(per_rank)


@pytest.mark.parametrize("W,H,N,scale,mono,G", [
    (384, 256, 5, 2, False, 2), (384, 256, 5, 2, False, 3), (384, 256, 7, 2, False, 8),
    (328, 200, 4, 4, False, 3), (256, 192, 3, 2, True, 5), (392, 264, 4, 3, False, 2),
    (256, 96, 3, 2, False, 8),     # 192 HR rows = 12 bands of 16 on 8 ranks: ragged stripes
])
def test_group_stripes_bit_identical_to_single_gpu(W, H, N, scale, mono, G):
    from multi_frame_super_resolution_amd.pipeline import default_config
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, mono)]
    cfg = default_config(W, H, N, scale, mono)
    cfg.reference = 1
    want = _single(cfg, frames, dev)
    grp, table = _group(cfg, G, frames, dev)
    assert grp.D.raw["mfsr_dist_transport"](grp.rank_handle(0)) == b"local"
    plans = []
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    probe = BurstPipeline(cfg, dev)
    plans = [probe.stripe_plan(G, r, 64) for r in range(G)]
    probe.close()
    fusing = [r for r in range(G) if plans[r].rowEnd > plans[r].rowBegin]
    for rep in range(3):                   # back to back: flag slots alternate, the gather of burst i overlaps burst i+1
        grp.out16.zero_()
        grp.process(table, "stripes")
        if rep == 1:
            grp.process(table, "stripes")  # two bursts in flight without a host synchronisation in between
        grp.synchronize()
        for r in range(G):
            assert int(grp.status[r].item()) == 0
        assert torch.equal(grp.out16, want), f"G={G} rep={rep}"
    # one packed message per peer whose stripe is not empty (if this rank owns a frame at all), + its stripe to rank 0
    from multi_frame_super_resolution_amd.distributed import frames_of_rank
    for r in range(G):
        msgs, nbytes = grp.exchange_stats(r)
        owns = len(frames_of_rank(N, r, G)) > 0
        expect = (len([p for p in fusing if p != r]) if owns else 0) + (1 if (r != 0 and r in fusing) else 0)
        assert msgs == expect, (r, msgs, expect)
        assert nbytes > 0 or expect == 0
    grp.close()


@pytest.mark.parametrize("mode", ["reduce", "reduce_scatter"])
@pytest.mark.parametrize("G,N", [(2, 5), (4, 5), (8, 3)])
def test_group_reduce_modes_match_single_gpu(mode, G, N):
    """Private accumulators summed over the ranks (another order than one GPU: equal to fp32 rounding).  G = 8 with 3
    frames: five ranks own no frame and contribute zeros."""
    from multi_frame_super_resolution_amd.pipeline import default_config
    W, H, scale = 320, 256, 2
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False)]
    cfg = default_config(W, H, N, scale, False)
    want = _single(cfg, frames, dev).cpu().numpy().view(np.uint16).astype(np.int64)
    grp, table = _group(cfg, G, frames, dev)
    for rep in range(2):
        grp.out16.zero_()
        grp.process(table, mode)
        grp.synchronize()
        got = grp.out16.cpu().numpy().view(np.uint16).astype(np.int64)
        d = np.abs(got - want)
        assert d.max() <= 2 and np.mean(d > 0) < 0.05, (mode, G, rep, int(d.max()), float(np.mean(d > 0)))
    grp.close()


def test_group_mode_switches_and_halo_status():
    """stripes -> reduce_scatter -> stripes on one group (gatherPending across the switch), then the status path: a raw
    halo smaller than the burst's vertical flow sets status 1 on EVERY rank; whole-frame halo repairs it."""
    from multi_frame_super_resolution_amd.pipeline import default_config
    W, H, N, scale, G = 384, 256, 5, 2, 3
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False, seed=7, max_shift=6.0)]
    cfg = default_config(W, H, N, scale, False)
    want = _single(cfg, frames, dev)
    grp, table = _group(cfg, G, frames, dev)
    grp.process(table, "stripes")
    grp.process(table, "reduce_scatter")
    grp.process(table, "stripes")
    grp.synchronize()
    assert torch.equal(grp.out16, want)
    assert all(int(s.item()) == 0 for s in grp.status)
    # a 4-row halo covers |flow.y| <= 1 px: the 6 px shifts of this burst exceed it
    grp.set_raw_halo(4)
    grp.process(table, "stripes")
    grp.synchronize()
    assert all(int(s.item()) == 1 for s in grp.status), [int(s.item()) for s in grp.status]
    grp.set_raw_halo(H)                    # whole raw frames: always valid
    grp.out16.zero_()
    grp.process(table, "stripes")
    grp.synchronize()
    assert all(int(s.item()) == 0 for s in grp.status)
    assert torch.equal(grp.out16, want)
    grp.close()


@pytest.mark.parametrize("G", [2, 3])
def test_pipelined_bursts_release_the_callers_frames_at_return(G):
    """Pipelined STRIPES bursts (world > 1): the fuse of burst i is enqueued by call i + 1.  include/mfsr_dist.h promises that
    the frame buffers of call i may be refilled once the caller's stream has passed call i (the rank keeps its own copy):
    burst A is enqueued, the ranks' streams are drained (NOT the group: that would end the pipeline), the SAME device
    buffers are overwritten with burst B's frames, burst B is enqueued with another output image -- image A must be burst
    A's and image B burst B's."""
    from multi_frame_super_resolution_amd.pipeline import default_config
    W, H, N, scale = 384, 256, 6, 2
    dev = torch.device("cuda:0")
    fa = [f.to(dev) for f in _burst(W, H, N, scale, False, seed=41)]
    fb = [f.to(dev) for f in _burst(W, H, N, scale, False, seed=43)]
    cfg = default_config(W, H, N, scale, False)
    cfg.reference = 2
    want_a, want_b = _single(cfg, fa, dev), _single(cfg, fb, dev)
    assert not torch.equal(want_a, want_b)
    live = [f.clone() for f in fa]                     # the buffers the group is given, refilled in place
    grp, table = _group(cfg, G, live, dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    out_a, out_b = torch.zeros_like(want_a), torch.zeros_like(want_a)
    torch.cuda.synchronize()
    grp.process(table, "stripes", out16=out_a, streams=streams)
    for st in streams:
        st.synchronize()                               # call A's work on the callers' streams is done; its fuse is not enqueued yet
    for dst, src in zip(live, fb):
        dst.copy_(src)
    torch.cuda.synchronize()
    grp.process(table, "stripes", out16=out_b, streams=streams)
    grp.synchronize(streams)
    assert all(int(s.item()) == 0 for s in grp.status)
    assert torch.equal(out_a, want_a), "image A was fused from refilled frame buffers"
    assert torch.equal(out_b, want_b)
    grp.close()


def test_measured_flow_sizes_the_raw_halo():
    """mfsr_dist_measured_flow = the largest |vertical flow| over all frames of the burst (max-reduced over the ranks): equal
    to the maximum over the single-GPU burst's per-frame flows; a raw halo of ceil(v) + 3 rows passes the status check and
    gives the same image, one of ceil(v) - 1 rows does not."""
    import math

    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config, view_as_tensor
    W, H, N, scale, G = 384, 256, 5, 2, 3
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False, seed=9, max_shift=5.0)]
    cfg = default_config(W, H, N, scale, False)
    p = BurstPipeline(cfg, dev)
    p.begin_burst()
    p.set_reference(frames[0])
    vmax = 0.0
    for k in range(N):
        p.add_frame(frames[k], k == 0)
        p.flush()
        flow_t, _, _, _ = p.debug_views()
        vmax = max(vmax, float(view_as_tensor(flow_t, 2, dev)[..., 1].abs().max()))
    p.finish()
    p.close()
    want = _single(cfg, frames, dev)       # (the burst above fused frame by frame: another summation order than a group's)
    grp, table = _group(cfg, G, frames, dev)
    grp.process(table, "stripes")
    grp.synchronize()
    v = grp.measured_flow()
    print(f"measured max |vertical flow| {v:.4f} px (single-GPU flows: {vmax:.4f})")
    assert abs(v - vmax) <= 1e-6 * max(1.0, vmax)
    assert torch.equal(grp.out16, want)
    halo = max(4, int(math.ceil(v)) + 3)
    grp.set_raw_halo(halo)
    grp.out16.zero_()
    grp.process(table, "stripes")
    grp.synchronize()
    assert all(int(s.item()) == 0 for s in grp.status)
    assert torch.equal(grp.out16, want)
    if int(math.ceil(v)) - 1 >= 4:
        grp.set_raw_halo(int(math.ceil(v)) - 1)        # covers |v| <= ceil(v) - 4: too small
        grp.process(table, "stripes")
        grp.synchronize()
        assert all(int(s.item()) == 1 for s in grp.status)
    grp.close()


def test_group_rejects_bad_arguments_without_hanging():
    """A rank that lacks one of its frames is refused before anything is enqueued -- on every rank's worker, so nobody
    waits for a peer that gave up (the call returns an error instead of timing out)."""
    import time

    from multi_frame_super_resolution_amd import capi
    from multi_frame_super_resolution_amd.pipeline import default_config
    W, H, N, scale, G = 256, 192, 4, 2, 2
    dev = torch.device("cuda:0")
    frames = [f.to(dev) for f in _burst(W, H, N, scale, False)]
    cfg = default_config(W, H, N, scale, False)
    grp, table = _group(cfg, G, frames, dev)
    table[1 * N + 3] = None                # rank 1 owns frames 1 and 3
    t0 = time.time()
    with pytest.raises(capi.MfsrError):
        grp.process(table, "stripes")
    assert time.time() - t0 < 30.0
    grp.close()
