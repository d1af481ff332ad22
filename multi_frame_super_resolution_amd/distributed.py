"""Frame-sharded multi-GPU burst processing: one process per GPU, RCCL over xGMI.

SURVEY.md section 8e: the burst (frame) dimension shards naturally.  Stages D
(flow), F (robustness) and G (accumulate) are independent per moved frame given
the reference-frame products, which every rank computes redundantly (they are
LR-sized); ``imgOut`` / ``totalWeights`` are plain sums over frames
(reference DeBayerKernels.cu:440-441,374-375), so the only exchange step is a
sum of the HR accumulators.  The reference itself is single-GPU
(``cudaSetDevice(0)``, kernel.cu:45): this module is new work.

THE PRODUCT IS ``csrc/dist.cpp`` (``include/mfsr_dist.h``: RCCL or in-process peer copies behind one transport table,
pipelined bursts, packed messages with 3-float certainties, measured raw halo, early stripe gather).  This module is its
torch.distributed MIRROR over the same C-ABI building blocks: the plain schedule without those refinements, kept for two
purposes only -- the world-2 gloo tests on the CPU (``tests/test_distributed_gloo.py``, driven through a stand-in pipe) and
``bench.py --dist-impl torch`` / ``MFSR_DIST_BACKEND=gloo`` rehearsals -- plus ``LocalGroup``, the ctypes wrapper of
``mfsr_dist_group_*``.  Three exchange modes:

``stripes``         (default) the FUSE stage is sharded over HR row stripes instead of over frames: every rank aligns
                    its own frames, sends each peer the rows of raw / flow / certainty that the peer's stripe reads
                    (point-to-point, a few MB per message, all 7 xGMI links of a GPU busy at once), fuses ALL frames in
                    frame order onto its stripe, finishes it, and rank 0 collects the u16 stripes.  ~20x less traffic
                    than summing accumulators, and bit-identical to the single-GPU burst (same summation order).

``reduce``          ``reduce(sum)`` of both accumulators onto rank 0, which then
                    runs the finish stage on the whole HR grid (what north_star
                    describes).  Root-bound: every peer sends the full 2 x HR x 12 B
                    over its one xGMI link to rank 0.
``reduce_scatter``  (default when the HR rows divide evenly) every rank receives
                    1/world of the HR rows summed over all ranks, finishes its own
                    stripe, and rank 0 gathers the 6 B/px u16 result.  All 7 xGMI
                    links of every GPU carry traffic at once instead of funnelling
                    through the root's.

Summation order differs from the sequential single-GPU loop, so results agree
with it to fp32 rounding (inside the +-1 LSB output budget), not bit for bit.

The functions are written against a small duck-typed "pipe" interface
(``cfg``, ``img_out``, ``total_weights``, ``reset_accumulators``,
``set_reference``, ``add_frame``, ``finish``, ``finish_rows``) so that the
world_size-2 gloo tests can drive them with a CPU stand-in.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def _staged(group) -> bool:
    """gloo moves host memory only (reduce / gather have no GPU path): a rehearsal of the multi-rank
    schedule on fewer GPUs than ranks (``MFSR_DIST_BACKEND=gloo`` in bench.py) stages through the host."""
    return dist.get_backend(group) == "gloo"


def _reduce_to(t: torch.Tensor, dst: int, group):
    if _staged(group) and t.is_cuda:
        h = t.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) == dst:
            t.copy_(h)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Round-robin frame shard: rank g owns frames {k : k mod world == g}."""
    return [k for k in range(n_frames) if k % world == rank]


def accumulate_local(pipe, frames, rank: int, world: int, n_frames: Optional[int] = None) -> List[int]:
    """Reference products (replicated) + this rank's frame shard into the local accumulators.

    ``frames`` is indexable by global frame number: a list of all frames, or a
    mapping that holds only this rank's shard plus the reference frame (what a
    rank of a large burst keeps resident)."""
    ref = pipe.cfg.reference
    n = len(frames) if n_frames is None else n_frames
    getattr(pipe, "begin_burst", pipe.reset_accumulators)()   # accumulators := 0 (lazily, where the pipe can)
    pipe.set_reference(frames[ref])
    mine = frames_of_rank(n, rank, world)
    for k in mine:
        pipe.add_frame(frames[k], k == ref)
    return mine


def exchange_and_finish(pipe, mode: str = "auto", group=None):
    """Sum the HR accumulators over ranks and finish.  Returns the u16 HR image on
    rank 0 (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    hr_h = pipe.img_out.shape[0]
    if mode == "auto":
        mode = "reduce_scatter" if (world > 1 and hr_h % world == 0) else "reduce"
    if world == 1:
        _, out16 = pipe.finish(want_float=False, want_u16=True)
        return out16
    if mode == "reduce":
        _reduce_to(pipe.img_out, 0, group)
        _reduce_to(pipe.total_weights, 0, group)
        if rank == 0:
            _, out16 = pipe.finish(want_float=False, want_u16=True)
            return out16
        return None
    if mode != "reduce_scatter":
        raise ValueError(f"unknown exchange mode {mode!r}")
    rows = hr_h // world
    row0 = rank * rows
    # in-place reduce-scatter: rank r's output is rows [r*rows, (r+1)*rows) of its own accumulator
    _reduce_scatter_rows(pipe.img_out, rows, rank, world, group)
    _reduce_scatter_rows(pipe.total_weights, rows, rank, world, group)
    out16 = pipe.finish_rows(row0, rows)  # full-size u16 buffer, stripe filled
    # gather as bytes: neither RCCL nor gloo has a 16-bit integer type
    out8 = out16.view(torch.uint8)
    stripe = out8[row0:row0 + rows]
    if _staged(group) and stripe.is_cuda:
        hs = stripe.cpu()
        if rank == 0:
            hparts = [torch.empty_like(hs) for _ in range(world)]
            dist.gather(hs, gather_list=hparts, dst=0, group=group)
            for r in range(world):
                out8[r * rows:(r + 1) * rows].copy_(hparts[r])
            return out16
        dist.gather(hs, gather_list=None, dst=0, group=group)
        return None
    if rank == 0:
        parts = [out8[r * rows:(r + 1) * rows] for r in range(world)]
        dist.gather(stripe, gather_list=parts, dst=0, group=group)
        return out16
    dist.gather(stripe, gather_list=None, dst=0, group=group)
    return None


def _reduce_scatter_rows(acc: torch.Tensor, rows: int, rank: int, world: int, group):
    flat = acc.view(-1)
    chunk = flat.numel() // world
    out = flat[rank * chunk:(rank + 1) * chunk]
    if dist.get_backend(group) == "gloo":
        # gloo has no reduce_scatter: emulate with per-chunk reduces (CPU tests / rehearsals only)
        for r in range(world):
            _reduce_to(flat[r * chunk:(r + 1) * chunk], r, group)
        return
    tmp = torch.empty_like(out)
    dist.reduce_scatter_tensor(tmp, flat, op=dist.ReduceOp.SUM, group=group)
    out.copy_(tmp)


class StripeBuffers:
    """Per-frame products of a whole burst on one rank (flow, certainty, raw of frames owned elsewhere), allocated once
    per context."""

    def __init__(self, pipe, n_frames: int, rank: int, world: int):
        self.flow, self.mask, self.raw = [], [], {}
        for k in range(n_frames):
            f, m = pipe.new_frame_products()
            self.flow.append(f)
            self.mask.append(m)
            if k % world != rank:
                self.raw[k] = torch.empty(pipe.cfg.height, pipe.cfg.width, dtype=torch.int16, device=pipe.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=pipe.device)


def _p2p(ops, group):
    """Run a list of (kind, tensor, peer) with kind in {"send", "recv"}: batched isend/irecv on RCCL; host-staged,
    pairwise ordered blocking calls on gloo (CPU tests / rehearsals with fewer GPUs than ranks)."""
    if not ops:
        return
    if not _staged(group):
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, t, p, group) for k, t, p in ops])
        for r in reqs:
            r.wait()
        return
    me = dist.get_rank(group)
    peers = sorted({p for _, _, p in ops})
    for p in peers:                       # one peer at a time; the lower rank sends first
        mine = [(k, t) for k, t, q in ops if q == p]
        order = ("send", "recv") if me < p else ("recv", "send")
        for phase in order:
            for k, t in mine:
                if k != phase:
                    continue
                if k == "send":
                    dist.send(t.cpu().contiguous(), dst=p, group=group)
                else:
                    h = torch.empty(t.shape, dtype=t.dtype)
                    dist.recv(h, src=p, group=group)
                    t.copy_(h)


def process_burst_stripes(pipe, frames, bufs: Optional[StripeBuffers] = None, group=None, n_frames: Optional[int] = None,
                          raw_halo: int = 64):
    """STRIPES mode (see the module docstring; mirrors csrc/dist.cpp::process_stripes).  ``frames`` maps global frame
    number -> raw tensor for this rank's frames and the reference.  Returns (u16 HR image on rank 0 else None, status):
    status 1 = a vertical flow exceeded the raw halo (result invalid)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(frames) if n_frames is None else n_frames
    ref = pipe.cfg.reference
    if bufs is None:
        bufs = StripeBuffers(pipe, n, rank, world)
    plans = [pipe.stripe_plan(world, p, raw_halo) for p in range(world)]
    mine = plans[rank]
    bufs.flag.zero_()
    pipe.set_reference(frames[ref])
    raws = {}
    for k in range(rank, n, world):
        raws[k] = frames[k]
        pipe.align_frame(frames[k], k == ref, bufs.flow[k], bufs.mask[k])
    if world > 1:
        ops = []
        for p in range(world):
            pl = plans[p]
            if p == rank or pl.rowEnd <= pl.rowBegin:
                continue
            for k in range(rank, n, world):
                ops.append(("send", raws[k][pl.rawRow0:pl.rawRow0 + pl.rawRows], p))
                ops.append(("send", bufs.flow[k][pl.flowRow0:pl.flowRow0 + pl.flowRows], p))
                ops.append(("send", bufs.mask[k][pl.maskRow0:pl.maskRow0 + pl.maskRows], p))
        if mine.rowEnd > mine.rowBegin:
            for k in range(n):
                owner = k % world
                if owner == rank:
                    continue
                raws[k] = bufs.raw[k]
                ops.append(("recv", bufs.raw[k][mine.rawRow0:mine.rawRow0 + mine.rawRows], owner))
                ops.append(("recv", bufs.flow[k][mine.flowRow0:mine.flowRow0 + mine.flowRows], owner))
                ops.append(("recv", bufs.mask[k][mine.maskRow0:mine.maskRow0 + mine.maskRows], owner))
        _p2p(ops, group)
    out16 = None
    if mine.rowEnd > mine.rowBegin:
        for k in range(n if mine.rawRows < pipe.cfg.height else 0):   # a whole-frame halo needs no check
            pipe.check_flow_bound(bufs.flow[k][mine.flowRow0:mine.flowRow0 + mine.flowRows], float(mine.maxFlowY), bufs.flag)
        per = pipe.group_size()   # the grouping of the single-GPU burst: same sums in the same order
        for k in range(0, n, per):
            ks = list(range(k, min(k + per, n)))
            pipe.fuse_rows([raws[j] for j in ks], [bufs.flow[j] for j in ks], [bufs.mask[j] for j in ks], mine.rowBegin,
                           mine.rowEnd, fresh=(k == 0))
        out16 = pipe.finish_rows(mine.rowBegin, mine.rowEnd - mine.rowBegin)
    else:
        out16 = pipe.out16
    if world > 1:
        out8 = out16.view(torch.uint8)
        ops = []
        if rank == 0:
            for p in range(1, world):
                pl = plans[p]
                if pl.rowEnd > pl.rowBegin:
                    ops.append(("recv", out8[pl.rowBegin:pl.rowEnd], p))
        elif mine.rowEnd > mine.rowBegin:
            ops.append(("send", out8[mine.rowBegin:mine.rowEnd], 0))
        _p2p(ops, group)
        if _staged(group):
            h = bufs.flag.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            bufs.flag.copy_(h)
        else:
            dist.all_reduce(bufs.flag, op=dist.ReduceOp.MAX, group=group)
    return (out16 if rank == 0 else None), bufs.flag


def process_burst(pipe, frames, mode: str = "auto", group=None, n_frames: Optional[int] = None):
    """Whole burst, frame-sharded over the ranks of ``group``; u16 HR image on rank 0."""
    if not dist.is_initialized():
        accumulate_local(pipe, frames, 0, 1, n_frames)
        _, out16 = pipe.finish(want_float=False, want_u16=True)
        return out16
    if mode == "stripes":
        return process_burst_stripes(pipe, frames, None, group, n_frames)[0]
    accumulate_local(pipe, frames, dist.get_rank(group), dist.get_world_size(group), n_frames)
    return exchange_and_finish(pipe, mode, group)


class LocalGroup:
    """``mfsr_dist_group_*`` (include/mfsr_dist.h): G ranks of a burst inside ONE process, one worker thread per rank in
    the library, peer copies in place of RCCL calls -- the same per-rank code as the RCCL contexts (csrc/dist.cpp).
    ``devices[r]`` = HIP device index of rank r; several ranks may share a device ("virtual ranks": how the multi-rank
    code is exercised on a one-GPU box)."""

    MODES = {"stripes": 0, "reduce": 1, "reduce_scatter": 2}

    def __init__(self, cfg, devices: Sequence[int]):
        import ctypes

        from . import capi
        self._ct = ctypes
        self.D = capi.dist_lib()
        self.cfg = cfg
        self.world = len(devices)
        self.devices = [int(d) for d in devices]
        nbytes = self.D.dist_workspace_bytes(ctypes.byref(cfg), self.world)
        if nbytes == 0:
            raise ValueError("mfsr_dist_workspace_bytes: invalid configuration")
        self._ws = [torch.empty(nbytes + 256, dtype=torch.uint8, device=f"cuda:{d}") for d in self.devices]
        bases = (ctypes.c_void_p * self.world)(*[(w.data_ptr() + 255) // 256 * 256 for w in self._ws])
        devs = (ctypes.c_int * self.world)(*self.devices)
        self._h = ctypes.c_void_p()
        self.D.dist_group_create(ctypes.byref(self._h), ctypes.byref(cfg), self.world, devs, bases, nbytes)
        s, W, H = cfg.scale, cfg.width, cfg.height
        self.out16 = torch.zeros(H * s, W * s, 3, dtype=torch.int16, device=f"cuda:{self.devices[0]}")
        self.status = [torch.zeros(1, dtype=torch.int32, device=f"cuda:{d}") for d in self.devices]
        self._status_ptrs = (ctypes.c_void_p * self.world)(*[t.data_ptr() for t in self.status])

    def rank_handle(self, r: int):
        return self.D.raw["mfsr_dist_group_rank"](self._h, r)

    def burst_handle(self, r: int):
        return self.D.raw["mfsr_dist_burst"](self.rank_handle(r))

    def set_raw_halo(self, halo: int):
        self.D.dist_group_set_raw_halo(self._h, int(halo))

    def frame_table(self, per_rank_frames):
        """per_rank_frames[r] = mapping frame number -> raw tensor ON devices[r] (the rank's own frames + the reference)."""
        ct, N = self._ct, self.cfg.frames
        tab = (ct.c_void_p * (self.world * N))()
        for r in range(self.world):
            for k in range(N):
                t = per_rank_frames[r].get(k)
                tab[r * N + k] = t.data_ptr() if t is not None else None
        return tab

    def _stream_table(self, streams):
        if streams is None:
            return None
        return (self._ct.c_void_p * self.world)(*[st.cuda_stream for st in streams])

    def process(self, table, mode: str = "stripes", out16: Optional[torch.Tensor] = None, streams=None):
        """One burst; returns after every rank has enqueued it.  ``streams``: one torch stream per rank (default: streams the
        group owns); ``out16``: where rank 0 collects this burst's image (default ``self.out16``) -- pipelined stripes
        bursts write it during the NEXT call (or ``synchronize``), see include/mfsr_dist.h."""
        dst = self.out16 if out16 is None else out16
        self.D.dist_group_process_burst(self._h, table, self.MODES[mode], dst.data_ptr(), self._status_ptrs,
                                        self._stream_table(streams))

    def synchronize(self, streams=None):
        self.D.dist_group_synchronize(self._h, self._stream_table(streams))

    def measured_flow(self) -> float:
        """Largest |vertical flow| (raw px) of the last burst, max over the ranks (mfsr_dist_measured_flow on rank 0)."""
        v = self._ct.c_float(0.0)
        with torch.cuda.device(self.devices[0]):
            self.D.dist_measured_flow(self.rank_handle(0), self._ct.byref(v), torch.cuda.current_stream().cuda_stream)
        return float(v.value)

    def timing(self, r: int, enable: bool):
        self.D.dist_timing(self.rank_handle(r), 1 if enable else 0)

    def timing_read(self, r: int):
        ct = self._ct
        ms, nl, nf = ct.c_double(0), ct.c_int(0), ct.c_int(0)
        self.D.dist_timing_read(self.rank_handle(r), ct.byref(ms), ct.byref(nl), ct.byref(nf))
        return ms.value, nl.value, nf.value

    def exchange_stats(self, r: int):
        ct = self._ct
        m, b = ct.c_longlong(0), ct.c_longlong(0)
        self.D.dist_exchange_stats(self.rank_handle(r), ct.byref(m), ct.byref(b))
        return m.value, b.value

    def close(self):
        if self._h:
            self.D.dist_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
