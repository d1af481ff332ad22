"""Host-side mirror of the burst driver in ``csrc/pipeline.cpp`` (C-ABI
``mfsr_burst_*``): N raw frames in -> one x-s frame out, the contract of the
reference CLI ``finalProject/Project/multi_frame_sr.cpp:146-209``.

torch only owns device memory and streams here; all arithmetic is in the HIP
library.  No CPU fallback: constructing a pipeline without a HIP device raises.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, Optional, Sequence

import torch

from . import capi


def default_config(width: int, height: int, frames: int, scale: int = 2, mono: bool = False) -> capi.Config:
    cfg = capi.Config()
    capi.lib().config_default(ctypes.byref(cfg), width, height, frames, scale, 1 if mono else 0)
    return cfg


class BurstPipeline:
    """One burst context on one device (ctx-per-device, not thread-safe; the
    reference is single-device/single-stream, kernel.cu:45)."""

    def __init__(self, cfg: capi.Config, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("multi_frame_super_resolution_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.L = capi.lib()
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        nbytes = self.L.burst_workspace_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            raise ValueError("invalid mfsr_config")
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = self.workspace.data_ptr()
            self._ws_ptr = (base + 255) // 256 * 256
            self.hr_w, self.hr_h = cfg.width * cfg.scale, cfg.height * cfg.scale
            # accumulators: float3 HR, pitch 12*hrW (caller-owned, RMW across frames,
            # reference DeBayerKernels.cu:306-307,374-375)
            self._img_out = torch.zeros(self.hr_h, self.hr_w, 3, dtype=torch.float32, device=self.device)
            self._total_weights = torch.zeros_like(self._img_out)
            self.out_img = torch.empty_like(self._img_out)
            self.out16 = torch.empty(self.hr_h, self.hr_w, 3, dtype=torch.int16, device=self.device)
            handle = ctypes.c_void_p()
            self.L.burst_create(ctypes.byref(handle), ctypes.byref(cfg), self._ws_ptr, nbytes)
            self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            self.L.burst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    # With cfg.pairFrames (the default) add_frame defers the warp+fuse of every other frame until its
    # partner is aligned (mfsr.h, mfsr_burst_add_frame): readers of the accumulators flush first.
    def flush(self):
        self.L.burst_flush(self._h, self._stream())

    @property
    def img_out(self) -> torch.Tensor:
        self.flush()
        return self._img_out

    @property
    def total_weights(self) -> torch.Tensor:
        self.flush()
        return self._total_weights

    def reset_accumulators(self):
        self.flush()
        self._img_out.zero_()
        self._total_weights.zero_()

    def begin_burst(self):
        """Start a burst without zeroing: the first warp+fuse launch overwrites the accumulators
        (mfsr_burst_begin).  Equivalent to reset_accumulators() for every reader (flush zeroes them if no
        frame was added)."""
        self.L.burst_begin(self._h, self._img_out.data_ptr(), self._total_weights.data_ptr(), self._stream())

    def set_reference(self, raw: torch.Tensor):
        self._check_raw(raw)
        self.L.burst_set_reference(self._h, raw.data_ptr(), self._stream())

    def add_frame(self, raw: torch.Tensor, is_reference: bool = False):
        self._check_raw(raw)
        self.L.burst_add_frame(self._h, raw.data_ptr(), 1 if is_reference else 0, self._img_out.data_ptr(),
                               self._total_weights.data_ptr(), self._stream())

    def finish(self, want_float: bool = True, want_u16: bool = True):
        self.L.burst_finish(self._h, self._img_out.data_ptr(), self._total_weights.data_ptr(),
                            self.out_img.data_ptr() if want_float else None,
                            self.out16.data_ptr() if want_u16 else None, self._stream())
        return (self.out_img if want_float else None), (self.out16 if want_u16 else None)

    def finish_rows(self, row0: int, rows: int) -> torch.Tensor:
        """Finish only HR rows [row0, row0+rows) (reduce-scatter mode); returns the
        full-size u16 buffer with that stripe filled."""
        self.L.burst_finish_rows(self._h, self._img_out.data_ptr(), self._total_weights.data_ptr(), None,
                                 self.out16.data_ptr(), row0, rows, self._stream())
        return self.out16

    def process(self, frames: Sequence[torch.Tensor], frame_ids: Optional[Iterable[int]] = None):
        """Whole burst on this device: reference products, every frame, finish."""
        self.begin_burst()
        ref = self.cfg.reference
        self.set_reference(frames[ref])
        ids = range(len(frames)) if frame_ids is None else frame_ids
        for k in ids:
            self.add_frame(frames[k], k == ref)
        return self.finish()

    def host_sync(self):
        """Block the host until the image of the last process_host has landed in host memory."""
        self.L.burst_host_sync(self._h)

    def process_joint(self, frames: Sequence[torch.Tensor]):
        """Whole burst with the joint shift minimiser in the loop (mfsr_burst_process_joint: every neighbouring pair is
        measured besides the (reference, k) pairs; per-tile least squares with outlier rejection gives the tile shifts)."""
        if len(frames) != self.cfg.frames:
            raise ValueError("process_joint needs exactly cfg.frames frames")
        for f in frames:
            self._check_raw(f)
        nbytes = self.L.burst_joint_workspace_bytes(ctypes.byref(self.cfg))
        if nbytes == 0:
            raise ValueError("the joint mode needs 2 <= frames <= 64")
        if getattr(self, "_joint_ws", None) is None or self._joint_ws.numel() < nbytes + 256:
            self._joint_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self._joint_ws.data_ptr() + 255) // 256 * 256
        ptrs = (ctypes.c_void_p * len(frames))(*[f.data_ptr() for f in frames])
        self.L.burst_process_joint(self._h, ptrs, base, nbytes, self._img_out.data_ptr(), self._total_weights.data_ptr(),
                                   self._stream())
        return self.finish()

    # ---- building blocks of stripe-sharded (multi-GPU) bursts: mfsr_burst_align_frame / mfsr_burst_fuse_rows ----
    def field_dims(self):
        """((flow_h, flow_w), (mask_h, mask_w)) of the per-frame products."""
        v = [ctypes.c_int() for _ in range(4)]
        self.L.burst_field_dims(self._h, *[ctypes.byref(x) for x in v])
        return (v[1].value, v[0].value), (v[3].value, v[2].value)

    def new_frame_products(self):
        """(flow [fh, fw, 2] f32, mask [mh, mw, 4] f32) device tensors with dense rows (a row range = one message)."""
        (fh, fw), (mh, mw) = self.field_dims()
        return (torch.empty(fh, fw, 2, dtype=torch.float32, device=self.device),
                torch.empty(mh, mw, 4, dtype=torch.float32, device=self.device))

    def align_frame(self, raw: torch.Tensor, is_reference: bool, flow: torch.Tensor, mask: torch.Tensor):
        self._check_raw(raw)
        self.L.burst_align_frame(self._h, raw.data_ptr(), 1 if is_reference else 0, flow.data_ptr(), flow.stride(0) * 4,
                                 mask.data_ptr(), mask.stride(0) * 4, self._stream())

    def group_size(self) -> int:
        """Frames per warp+fuse launch cfg.pairFrames stands for (mfsr_burst_group_size)."""
        return int(self.L.raw["mfsr_burst_group_size"](ctypes.byref(self.cfg)))

    def fuse_rows(self, raws, flows, masks, row_begin: int, row_end: int, fresh: bool):
        n = len(raws)
        P = ctypes.c_void_p * n
        self.L.burst_fuse_rows(self._h, n, P(*[r.data_ptr() for r in raws]), P(*[f.data_ptr() for f in flows]),
                               flows[0].stride(0) * 4, P(*[m.data_ptr() for m in masks]), masks[0].stride(0) * 4,
                               self._img_out.data_ptr(), self._total_weights.data_ptr(), 1 if fresh else 0, row_begin, row_end,
                               self._stream())

    def check_flow_bound(self, flow_rows: torch.Tensor, bound: float, flag: torch.Tensor):
        """flag |= 1 if a vertical flow of these rows exceeds ``bound`` (mfsr_checkFlowBound)."""
        self.L.checkFlowBound(flow_rows.data_ptr(), flow_rows.stride(0) * 4, flow_rows.shape[1], flow_rows.shape[0], bound,
                              flag.data_ptr(), self._stream())

    def stripe_plan(self, world: int, rank: int, raw_halo: int = 64) -> "capi.StripePlan":
        plan = capi.StripePlan()
        self.L.dist_stripe_plan(ctypes.byref(self.cfg), world, rank, raw_halo, ctypes.byref(plan))
        return plan

    # ---- frames in (pinned) host memory: the library uploads them on its own copy stream (cfg.uploadRing > 0) ----
    def process_host(self, host_frames: Sequence[torch.Tensor], out16_host: Optional[torch.Tensor] = None):
        """Whole burst from HOST frames (pin them: ``t.pin_memory()``) to the u16 HR image in host memory:
        mfsr_burst_set_reference_host / add_frame_host / finish_host.  Returns the (pinned) host image; it is complete
        after ``host_sync()`` (the download runs on a stream of its own so that the next burst overlaps it)."""
        if self.cfg.uploadRing <= 0:
            raise ValueError("cfg.uploadRing must be > 0 for host-frame bursts")
        for f in host_frames:
            if f.is_cuda or f.dtype not in (torch.int16, torch.uint16) or not f.is_contiguous() or \
                    tuple(f.shape) != (self.cfg.height, self.cfg.width):
                raise ValueError("host frames must be contiguous 16-bit CPU tensors of the configured size")
        if out16_host is None:
            if getattr(self, "_out16_host", None) is None:
                self._out16_host = torch.empty(self.hr_h, self.hr_w, 3, dtype=torch.int16).pin_memory()
            out16_host = self._out16_host
        st = self._stream()
        self.begin_burst()
        ref = self.cfg.reference
        self.L.burst_set_reference_host(self._h, host_frames[ref].data_ptr(), st)
        # every copy queued before the first kernel: the copy engine then runs them back to back (mfsr_burst_prefetch_host)
        n = len(host_frames)
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in host_frames])
        self.L.burst_prefetch_host(self._h, ptrs, n, st)
        for k, f in enumerate(host_frames):
            self.L.burst_add_frame_host(self._h, f.data_ptr(), 1 if k == ref else 0, self._img_out.data_ptr(),
                                        self._total_weights.data_ptr(), st)
        self.L.burst_finish_host(self._h, self._img_out.data_ptr(), self._total_weights.data_ptr(), self.out16.data_ptr(),
                                 out16_host.data_ptr(), st)
        return out16_host

    def debug_views(self):
        """(flow, mask, kernel_param, tracking) descriptors of the last add_frame."""
        t = [capi.Tex2D() for _ in range(4)]
        self.L.burst_debug_views(self._h, *[ctypes.byref(x) for x in t])
        return t

    def frame_views(self, frames_back: int):
        """(flow, mask) descriptors of the frame aligned ``frames_back`` frames before the last one (mfsr_burst_debug_frame_views).
        With frame-batched alignment a frame is aligned when its group is complete (or on flush / finish), not by add_frame."""
        f, m = capi.Tex2D(), capi.Tex2D()
        self.L.burst_debug_frame_views(self._h, int(frames_back), ctypes.byref(f), ctypes.byref(m))
        return f, m

    def _check_raw(self, raw: torch.Tensor):
        if raw.device != self.device or raw.dtype not in (torch.int16, torch.uint16) or not raw.is_contiguous():
            raise ValueError("raw frame must be a contiguous 16-bit tensor on the pipeline's device")
        if tuple(raw.shape) != (self.cfg.height, self.cfg.width):
            raise ValueError(f"raw frame must be {self.cfg.height}x{self.cfg.width}, got {tuple(raw.shape)}")


class FrameStream:
    """Sliding-window stream (C-ABI ``mfsr_stream_*``; the reference's ``setTemporalAreaRadius``,
    finalProject/Project/multi_frame_sr.cpp:182): output t fuses frames [t-radius, t+radius] around reference t.  Every
    frame is uploaded and prepared once.  ``host_frames``: frames are pinned CPU tensors, uploaded by the library's
    copy stream."""

    def __init__(self, cfg: capi.Config, radius: int = 1, device: Optional[torch.device] = None, host_frames: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("multi_frame_super_resolution_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.L = capi.lib()
        self.cfg = cfg
        self.radius = radius
        self.host_frames = host_frames
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        nbytes = self.L.stream_workspace_bytes(ctypes.byref(cfg), radius)
        if nbytes == 0:
            raise ValueError("invalid mfsr_config / radius")
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = (self.workspace.data_ptr() + 255) // 256 * 256
            self.hr_w, self.hr_h = cfg.width * cfg.scale, cfg.height * cfg.scale
            self.out16 = torch.empty(self.hr_h, self.hr_w, 3, dtype=torch.int16, device=self.device)
            h = ctypes.c_void_p()
            self.L.stream_create(ctypes.byref(h), ctypes.byref(cfg), radius, 1 if host_frames else 0, base, nbytes)
            self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, frame: torch.Tensor):
        """Hand over the next frame; returns (t, u16 HR image) once output t is produced (the image is this object's
        buffer, valid until the next push / drain), else None."""
        if frame.is_cuda == self.host_frames or not frame.is_contiguous() or tuple(frame.shape) != (self.cfg.height, self.cfg.width):
            raise ValueError("frame must be a contiguous 16-bit tensor of the configured size, on the "
                             + ("host" if self.host_frames else "device"))
        produced = ctypes.c_longlong(-1)
        self.L.stream_push(self._h, frame.data_ptr(), None, self.out16.data_ptr(), ctypes.byref(produced),
                           torch.cuda.current_stream().cuda_stream)
        return (produced.value, self.out16) if produced.value >= 0 else None

    def drain(self):
        """End of the stream: yields the outstanding (t, u16 HR image) outputs."""
        while True:
            produced = ctypes.c_longlong(-1)
            self.L.stream_drain(self._h, None, self.out16.data_ptr(), ctypes.byref(produced), torch.cuda.current_stream().cuda_stream)
            if produced.value < 0:
                return
            yield produced.value, self.out16


def view_as_tensor(t: capi.Tex2D, channels: int, device) -> torch.Tensor:
    """Copy a device image described by a Tex2D into a fresh [H, W, C] float tensor (tests)."""
    L = capi.lib()
    out = torch.empty(t.height, t.width, channels, dtype=torch.float32, device=device)
    # hipMemcpy2D through torch: build a strided view over the raw pointer is not possible
    # without owning it, so go through the C-ABI's resample-free path: a 1:1 float copy
    # kernel is not exported; use ctypes + hipMemcpy2DAsync from libamdhip64 instead.
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy2D.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                ctypes.c_size_t, ctypes.c_int]
    rc = hip.hipMemcpy2D(out.data_ptr(), t.width * channels * 4, t.ptr, t.pitch, t.width * channels * 4, t.height, 3)
    if rc != 0:
        raise RuntimeError(f"hipMemcpy2D failed: {rc}")
    torch.cuda.synchronize()
    return out
