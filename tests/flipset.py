"""Classification of HIP-vs-oracle differences of the burst pipeline (test helper).

The per-pixel flow is the only non-bit-exact intermediate of the path (the Lucas-Kanade solve uses
atan2/cos/sin/sqrt: ocml and glibc agree to 1-2 ulp, CUDA's own differ from both), and it reaches the fused image
only through three discontinuities (oracle/diagnostics.c):

  (1) round(s * flow)   per HR pixel     accumulateImagesSuperRes, DeBayerKernels.cu:403-406
  (2) round(0.5 * flow) per half-res px  ComputeRobustnessMask, RobustnessModell.cu:76-77
  (3) M > thresholdM    per half-res px  ComputeRobustnessMask, RobustnessModell.cu:147-148
plus, on the accumulated weights,
  (4) weight < threshold per channel     ApplyWeighting, kernel.cu:444-462

``FlipSet`` collects, frame by frame, the HR pixels whose value can be affected by a flip of (1)-(4) between the two
implementations (the "flip set"); everywhere else the two outputs must agree to +-1 LSB with NO exception.
A pixel is also put in the set when either side is within ``tie_eps`` of a rounding tie / of the threshold: the HIP
kernels evaluate the same bilinear blend with the weights in another order (1 ulp), so a value that close to a tie
may flip inside the kernel without showing in the recomputed roundings.  ``tie_eps`` = 2.5e-7 relative = two ulp: what that
reorder can move (round 2 used 4e-6, which made the set ~4x the real flips; the sweep 4e-6 / 1e-6 / 2.5e-7 / 0 of
tools/parity_audit.py leaves the outside-the-set statistics unchanged -- profiles/r03_parity_audit_*.log).  The report
carries the set's size with and without the guard.
"""
from __future__ import annotations

import numpy as np


def _pitch(a):
    return int(a.strides[0])


BAND_ROWS = 256


def map_row_bands(fn, rows, band_rows=None, threads=None):
    """[fn(y0, y1) for consecutive row bands] on a thread pool (the HR arrays of a 4K x4 / 8K burst have 133 M pixels: whole-array
    numpy expressions on them are single-threaded and allocate gigabytes of temporaries)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    band_rows = band_rows or BAND_ROWS
    bands = [(y, min(y + band_rows, rows)) for y in range(0, rows, band_rows)]
    threads = threads or min(16, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(lambda b: fn(*b), bands))


def _dilate(m: np.ndarray, r: int) -> np.ndarray:
    """Binary dilation by a (2r+1)^2 box (numpy only)."""
    out = m.copy()
    H, W = m.shape
    for dy in range(-r, r + 1):
        ys, yd = (slice(max(dy, 0), H + min(dy, 0)), slice(max(-dy, 0), H + min(-dy, 0)))
        for dx in range(-r, r + 1):
            xs, xd = (slice(max(dx, 0), W + min(dx, 0)), slice(max(-dx, 0), W + min(-dx, 0)))
            out[yd, xd] |= m[ys, xs]
    return out


def _near_tie(v: np.ndarray, eps: float) -> np.ndarray:
    """|v - (n + 0.5)| < eps*max(1,|v|) for some integer n, any component."""
    fr = np.abs(v - np.floor(v) - 0.5)
    return (fr < eps * np.maximum(1.0, np.abs(v))).any(-1)


class FlipSet:
    def __init__(self, cfg, tie_eps: float = 2.5e-7):
        from oracle.bindings import oracle
        self.o = oracle()
        self.c = cfg
        self.s = cfg.scale
        self.W, self.H = cfg.width, cfg.height
        self.hw, self.hh = self.W // 2, self.H // 2
        self.hrW, self.hrH = self.W * self.s, self.H * self.s
        self.tie_eps = tie_eps
        self.flips = np.zeros((self.hrH, self.hrW), bool)
        self.flips_actual = np.zeros((self.hrH, self.hrW), bool)     # decisions that really differ (no tie guard)
        self.n = dict(fuse_round=0, fuse_round_actual=0, robust_round=0, robust_M=0, weight_threshold=0)
        self.frames = 0
        self.max_flow_diff = 0.0

    def _fuse_shifts(self, flow):
        val = np.zeros((self.hrH, self.hrW, 2), np.float32)
        sh = np.zeros((self.hrH, self.hrW, 2), np.int32)
        self.o.dbgFuseShifts(flow, _pitch(flow), flow.shape[1], flow.shape[0], self.W, self.H, self.s, val, sh)
        return val, sh

    def _robust_shifts(self, flow):
        val = np.zeros((self.hh, self.hw, 2), np.float32)
        sh = np.zeros((self.hh, self.hw, 2), np.int32)
        self.o.dbgRobustnessShifts(flow, _pitch(flow), flow.shape[1], flow.shape[0], self.hw, self.hh, val, sh)
        return val, sh

    def add_frame(self, flow_h, flow_o, mask_h, mask_o, numpy_reference=False):
        """flow_*: [th, tw, 2] float32 (raw-pixel units, what the fuse and robustness kernels read);
        mask_*: [hh, hw, 4] float32 (.w = M).  numpy_reference: decision (1) with the four HR-sized numpy arrays per frame
        this class started with (kept as the check of the C form, tests/test_flipset_cpu.py)."""
        flow_h = np.ascontiguousarray(flow_h, np.float32)
        flow_o = np.ascontiguousarray(flow_o, np.float32)
        self.frames += 1
        self.max_flow_diff = max(self.max_flow_diff, float(np.abs(flow_h - flow_o).max()))
        # (1)
        if numpy_reference:
            vh, sh = self._fuse_shifts(flow_h)
            vo, so = self._fuse_shifts(flow_o)
            a1 = (sh != so).any(-1)                                    # roundings that really differ
            f1 = a1 | ((vh != vo).any(-1) & (_near_tie(vh, self.tie_eps) | _near_tie(vo, self.tie_eps)))   # + the tie guard
            self.n["fuse_round"] += int(f1.sum())
            self.n["fuse_round_actual"] += int(a1.sum())
            self.flips |= f1
            self.flips_actual |= a1
        else:
            # the same decision per HR pixel in one pass of oracle/diagnostics.c (OpenMP), written into the two sets in place
            assert flow_h.shape == flow_o.shape and _pitch(flow_h) == _pitch(flow_o)
            counts = np.zeros(2, np.int64)
            self.o.dbgFuseFlips(flow_h, flow_o, _pitch(flow_h), flow_h.shape[1], flow_h.shape[0], self.W, self.H, self.s,
                                float(self.tie_eps), self.flips.view(np.uint8), self.flips_actual.view(np.uint8), counts)
            self.n["fuse_round"] += int(counts[0])
            self.n["fuse_round_actual"] += int(counts[1])
        # (2) + (3), half resolution
        rvh, rsh = self._robust_shifts(flow_h)
        rvo, rso = self._robust_shifts(flow_o)
        f2 = (rsh != rso).any(-1) | ((rvh != rvo).any(-1) & (_near_tie(rvh, self.tie_eps) | _near_tie(rvo, self.tie_eps)))
        thr = float(self.c.thresholdM)
        Mh, Mo = mask_h[..., 3], mask_o[..., 3]
        f3 = ((Mh > thr) != (Mo > thr)) | ((Mh != Mo) & ((np.abs(Mh - thr) < 1e-5 * thr) | (np.abs(Mo - thr) < 1e-5 * thr)))
        self.n["robust_round"] += int(f2.sum())
        self.n["robust_M"] += int(f3.sum())
        fm = f2 | f3
        fm_actual = (rsh != rso).any(-1) | ((Mh > thr) != (Mo > thr))
        for m, dst in ((fm, self.flips), (fm_actual, self.flips_actual)):
            if m.any():
                # certainty site of HR pixel X, tap px: floor((X+px)/s)/2, px in [-2,2]  ->  half-res pixel x is read by
                # X in [2s*x - 2, 2s*x + 2s + 1]
                ys, xs = np.nonzero(m)
                if numpy_reference or ys.size > 50000:
                    up = np.repeat(np.repeat(m, 2 * self.s, 0), 2 * self.s, 1)[:self.hrH, :self.hrW]
                    dst[:up.shape[0], :up.shape[1]] |= _dilate(up, 2)
                else:   # the usual case, a few hundred cells: their HR blocks directly
                    b = 2 * self.s
                    ly, lx = min(b * m.shape[0], self.hrH), min(b * m.shape[1], self.hrW)   # (the up-sampled mask's extent)
                    for y, x in zip(ys.tolist(), xs.tolist()):
                        dst[max(b * y - 2, 0):min(b * y + b + 2, ly), max(b * x - 2, 0):min(b * x + b + 2, lx)] = True

    def add_weights(self, tw_h, tw_o):
        """(4): the accumulated weights on either side of ApplyWeighting's threshold."""
        thr = float(self.c.weightThreshold)
        f4 = ((tw_h < thr) != (tw_o < thr)).any(-1)
        self.n["weight_threshold"] += int(f4.sum())
        self.flips |= f4
        self.flips_actual |= f4

    def report(self, h_out, o_out, h16, o16):
        """Error statistics inside / outside the flip set (8 bit: the CLI's output depth; 16 bit as the finer diagnostic).
        Row bands on a thread pool (numpy releases the GIL): the same per-sample expressions, partial maxima / counts merged."""
        E = self.flips
        tot = float(E.size)

        def band(y0, y1):
            e = E[y0:y1]
            d8 = np.abs(np.round(np.clip(h_out[y0:y1], 0, 1) * 255.0) - np.round(np.clip(o_out[y0:y1], 0, 1) * 255.0)).max(-1)
            d16 = np.abs(h16[y0:y1].astype(np.int64) - o16[y0:y1].astype(np.int64)).max(-1)
            i8, o8, i16, o16_ = d8[e], d8[~e], d16[e], d16[~e]
            return dict(max8_inside=int(i8.max()) if i8.size else 0, max8_outside=int(o8.max()) if o8.size else 0,
                        n_gt1_8bit_inside=int((i8 > 1).sum()), n_gt1_8bit_outside=int((o8 > 1).sum()),
                        max16_inside=int(i16.max()) if i16.size else 0, max16_outside=int(o16_.max()) if o16_.size else 0,
                        n16_out=int((o16_ > 1).sum()), n8=int((d8 > 1).sum()), n16=int((d16 > 1).sum()))

        parts = map_row_bands(band, E.shape[0])
        mx = lambda k: max(q[k] for q in parts)
        sm = lambda k: sum(q[k] for q in parts)
        r = {
            "flip_fraction": float(E.sum()) / tot,
            "flip_fraction_no_guard": float(self.flips_actual.sum()) / tot,
            "tie_eps": self.tie_eps,
            "flips_by_cause_per_frame": {k: v / max(self.frames, 1) / tot for k, v in self.n.items()},
            "max_flow_diff_px": self.max_flow_diff,
            "max8_inside": mx("max8_inside"),
            "max8_outside": mx("max8_outside"),
            "n_gt1_8bit_inside": sm("n_gt1_8bit_inside"),
            "n_gt1_8bit_outside": sm("n_gt1_8bit_outside"),
            "max16_inside": mx("max16_inside"),
            "max16_outside": mx("max16_outside"),
            "frac_gt1_16bit_outside": float(sm("n16_out")) / tot,
            "frac_gt1_8bit": float(sm("n8")) / tot,
            "frac_gt1_16bit": float(sm("n16")) / tot,
        }
        return r
