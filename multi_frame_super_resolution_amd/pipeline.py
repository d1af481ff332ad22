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

    # ---- frames in (pinned) host memory: the library uploads them on its own copy stream (cfg.uploadRing > 0) ----
    def process_host(self, host_frames: Sequence[torch.Tensor], out16_host: Optional[torch.Tensor] = None):
        """Whole burst from HOST frames (pin them: ``t.pin_memory()``) to the u16 HR image in host memory:
        mfsr_burst_set_reference_host / add_frame_host / finish_host.  Returns the (pinned) host image; it is complete
        once the current stream has been synchronised."""
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
        for k, f in enumerate(host_frames):
            self.L.burst_add_frame_host(self._h, f.data_ptr(), 1 if k == ref else 0, self._img_out.data_ptr(),
                                        self._total_weights.data_ptr(), st)
        self.L.burst_finish_host(self._h, self._img_out.data_ptr(), self._total_weights.data_ptr(), self.out16.data_ptr(),
                                 out16_host.data_ptr(), st)
        return out16_host

    def process_stream(self, frames: Sequence[torch.Tensor], radius: int = 1):
        """Sliding-window ("temporal area radius", reference multi_frame_sr.cpp:182) use of the burst path:
        output t fuses frames [t-radius, t+radius] (clipped to the stream) with frame t as the reference.
        Yields (t, u16 HR image) -- the image is the pipeline's output buffer, valid until the next step.
        cfg.frames only sizes nothing here: any window length works with one context."""
        n = len(frames)
        for t in range(n):
            lo, hi = max(0, t - radius), min(n - 1, t + radius)
            self.begin_burst()
            self.set_reference(frames[t])
            for k in range(lo, hi + 1):
                self.add_frame(frames[k], k == t)
            _, out16 = self.finish(want_float=False, want_u16=True)
            yield t, out16

    def debug_views(self):
        """(flow, mask, kernel_param, tracking) descriptors of the last add_frame."""
        t = [capi.Tex2D() for _ in range(4)]
        self.L.burst_debug_views(self._h, *[ctypes.byref(x) for x in t])
        return t

    def _check_raw(self, raw: torch.Tensor):
        if raw.device != self.device or raw.dtype not in (torch.int16, torch.uint16) or not raw.is_contiguous():
            raise ValueError("raw frame must be a contiguous 16-bit tensor on the pipeline's device")
        if tuple(raw.shape) != (self.cfg.height, self.cfg.width):
            raise ValueError(f"raw frame must be {self.cfg.height}x{self.cfg.width}, got {tuple(raw.shape)}")


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
