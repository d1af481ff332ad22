"""CPU oracle of the burst pipeline -- TEST INFRASTRUCTURE ONLY.

Composes the oracle kernels (oracle/*.c) in the stage order of
multi_frame_super_resolution_amd/csrc/pipeline.cpp (its un-fused path, one call
per reference kernel; SURVEY.md section 3.3 letters A..H).  PARITY UNPINNED:
the reference has no host for this path, so the stage order is the build's own
reconstruction (DESIGN.md "Pipeline glue").

``cfg`` is any object with the fields of ``mfsr_config`` (include/mfsr.h).
"""
from __future__ import annotations

import numpy as np

from .bindings import oracle


def _ilog2(v: int) -> int:
    l = 0
    while (1 << l) < v:
        l += 1
    return l


def _pitch(a: np.ndarray) -> int:
    return int(a.strides[0])


class OraclePipeline:
    def __init__(self, cfg):
        self.o = oracle()
        self.c = cfg
        c = cfg
        self.W, self.H = c.width, c.height
        self.hw, self.hh = c.width // 2, c.height // 2
        self.tw, self.th = (self.W, self.H) if c.mono else (self.hw, self.hh)
        self.flow_scale = 1 if c.mono else 2
        self.hrW, self.hrH = c.width * c.scale, c.height * c.scale
        self.taps = np.zeros(99, np.float32)
        self.ntaps = self.o.gaussin_filter_1D(float(c.sigmaTracking), self.taps)
        self.ttaps = np.zeros(99, np.float32)
        self.nttaps = self.o.gaussin_filter_1D(float(c.sigmaTensor), self.ttaps)
        self.nl = _ilog2(c.levelFactor[0]) + 1
        self.tc = []
        for l in range(c.levels):
            f = c.levelFactor[l]
            lw, lh = self.tw // f, self.th // f
            self.tc.append((max(lw // c.tileSize[l], 1), max(lh // c.tileSize[l], 1)))
        self.white = np.array(list(c.white), np.float32)
        self.black = np.array(list(c.black), np.float32)
        self.cfa = np.array(list(c.cfa), np.int32)
        self.flow = None
        self.mask = None

    # A1 + tracking pyramid
    def _prepare(self, raw):
        o, c = self.o, self.c
        o.set_cfa_pattern(self.cfa)
        half = np.zeros((self.hh, self.hw, 3), np.float32)
        max_val = 2.0 * c.maxVal if c.mono else c.maxVal
        o.deBayersSubSample3(raw, half, float(max_val), self.hw, self.hh, _pitch(half))
        tmp = np.zeros((self.th, self.tw), np.float32)
        if c.mono:
            o.u16ToFloat(raw, tmp, _pitch(tmp), self.W, self.H, float(np.float32(1.0) / np.float32(c.maxVal)))
        else:
            o.rgbToGray(half, _pitch(half), tmp, _pitch(tmp), self.tw, self.th)
        pyr = [np.zeros((self.th, self.tw), np.float32)]
        tmp2 = np.zeros_like(tmp)
        o.separableFilter(tmp, _pitch(tmp), tmp2, pyr[0], _pitch(pyr[0]), self.tw, self.th, 1, self.taps, self.ntaps)
        for i in range(1, self.nl):
            lw, lh = self.tw >> i, self.th >> i
            nxt = np.zeros((lh, lw), np.float32)
            o.downsample2x(pyr[i - 1], _pitch(pyr[i - 1]), nxt, _pitch(nxt), lw, lh)
            pyr.append(nxt)
        return half, pyr

    def set_reference(self, raw):
        o, c = self.o, self.c
        self.ref_half, self.ref_pyr = self._prepare(raw)
        t0 = self.ref_pyr[0]
        Ix, Iy = np.zeros_like(t0), np.zeros_like(t0)
        o.ComputeDerivatives2Kernel(self.tw, self.th, _pitch(Ix), Ix, Iy, t0, _pitch(t0), self.tw, self.th)
        tensor = np.zeros((self.th, self.tw, 3), np.float32)
        o.ComputeStructureTensor(Ix, Iy, tensor, self.tw, self.th, _pitch(Ix), _pitch(tensor))
        tmp, sm = np.zeros_like(tensor), np.zeros_like(tensor)
        o.separableFilter(tensor, _pitch(tensor), tmp, sm, _pitch(sm), self.tw, self.th, 3, self.ttaps, self.nttaps)
        o.ComputeKernelParam(sm, self.tw, self.th, _pitch(sm), float(c.Dth), float(c.Dtr), float(c.kDetail),
                             float(c.kDenoise), float(c.kStretch), float(c.kShrink))
        self.kparam4 = np.zeros((self.th, self.tw, 4), np.float32)
        o.float3ToFloat4(sm, _pitch(sm), self.kparam4, _pitch(self.kparam4), self.tw, self.th)
        rawf = np.zeros((self.H, self.W), np.float32)
        o.u16ToFloat(raw, rawf, _pitch(rawf), self.W, self.H, 1.0)
        self.fallback = np.zeros((self.H, self.W, 3), np.float32)
        scale = (np.float32(1.0) / self.white).astype(np.float32)
        o.deBayerGreenKernel(self.W, self.H, rawf, _pitch(rawf), self.fallback, _pitch(self.fallback), self.black, scale)
        o.deBayerRedBlueKernel(self.W, self.H, rawf, _pitch(rawf), self.fallback, _pitch(self.fallback), self.black, scale)

    def _track(self, mov_pyr, base=(0.0, 0.0, 0.0)):
        o, c = self.o, self.c
        shifts = None
        for l in range(c.levels):
            pi = _ilog2(c.levelFactor[l])
            ref, mov = self.ref_pyr[pi], mov_pyr[pi]
            lh, lw = ref.shape
            T, S = c.tileSize[l], c.maxShift[l]
            tcx, tcy = self.tc[l]
            n, L, R = tcx * tcy, T + 2 * S, 2 * S + 1
            pre = np.zeros((tcy, tcx, 2), np.float32)
            if l > 0:
                ptcx, ptcy = self.tc[l - 1]
                o.UpSampleShifts(shifts, pre, _pitch(shifts), _pitch(pre), c.levelFactor[l - 1], c.levelFactor[l], ptcx,
                                 ptcy, tcx, tcy, c.tileSize[l - 1], T)
            rt = np.zeros((n, L, L), np.float32)
            mt = np.zeros((n, L, L), np.float32)
            cc = np.zeros((n, L, L), np.float32)
            bx = np.zeros((n, L, L), np.float32)
            by = np.zeros((n, L, L), np.float32)
            sq = np.zeros(n, np.float32)
            dist = np.zeros((n, R, R), np.float32)
            found = np.zeros((tcy, tcx, 2), np.float32)
            o.convertToTilesOverlapBorder(ref, rt, lw, lh, _pitch(ref), S, T, tcx, tcy, 0.0, 0.0, 0.0)
            inv = np.float32(1.0) / np.float32(c.levelFactor[l])   # base shift in pixels of this pyramid level
            o.convertToTilesOverlapPreShift(mov, mt, pre, _pitch(pre), lw, lh, _pitch(mov), S, T, tcx, tcy,
                                            float(np.float32(base[0]) * inv), float(np.float32(base[1]) * inv), float(base[2]))
            o.crossCorrelateTiles(rt, mt, cc, S, T, n)
            o.squaredSum(rt, sq, S, T, n)
            o.boxFilterWithBorderX(mt, bx, S, T, n)
            o.boxFilterWithBorderY(bx, by, S, T, n)
            o.normalizedCC(cc, sq, by, dist, S, T, n)
            o.findMinimum(dist, found, _pitch(found), S, n, tcx, float(c.minimumThreshold))
            o.addRoundedPreShift(pre, _pitch(pre), found, _pitch(found), tcx, tcy)
            shifts = found
        return shifts

    def add_frame(self, raw, is_reference, img_out, total_weights, given_shifts=None):
        """given_shifts: tile shifts [tcy, tcx, 2] from the joint minimiser (process_joint) instead of the tracker's."""
        o, c = self.o, self.c
        o.set_cfa_pattern(self.cfa)
        if is_reference:
            flow = np.zeros((self.th, self.tw, 2), np.float32)
            mask = np.ones((self.hh, self.hw, 4), np.float32)
        else:
            mov_half, mov_pyr = self._prepare(raw)
            base = (0.0, 0.0, 0.0)
            if getattr(c, "preAlign", 0):
                # I: global pre-alignment (oracle/prealign.c) -> baseShift / baseRotation of B2 and D1
                res = np.zeros(5, np.float32)
                st = np.zeros(4, np.int32)
                o.preAlign(self.ref_pyr[0], mov_pyr[0], self.tw, self.th, _pitch(self.ref_pyr[0]), float(c.preAlignMaxAngle),
                           res, st)
                base = (float(res[0]), float(res[1]), float(res[2]))
                self.prealign = dict(shift=(float(res[0]), float(res[1])), rotation=float(res[2]), angle_index=int(st[0]),
                                     t=(int(st[1]), int(st[2])), level=int(st[3]))
            shifts = self._track(mov_pyr, base) if given_shifts is None else given_shifts
            tcx, tcy = self.tc[-1]
            flow = np.zeros((self.th, self.tw, 2), np.float32)
            o.CreateFlowFieldFromTiles(flow, shifts, _pitch(shifts), tcx, tcy, c.tileSize[c.levels - 1], tcx, tcy,
                                       self.tw, self.th, _pitch(flow), base[0], base[1], base[2])
            ref0, mov0 = self.ref_pyr[0], mov_pyr[0]
            warped = np.zeros_like(ref0)
            Ix, Iy, It = np.zeros_like(ref0), np.zeros_like(ref0), np.zeros_like(ref0)
            for _ in range(c.lkIterations):
                o.WarpingKernel(self.tw, self.th, _pitch(warped), flow, _pitch(flow), self.tw, self.th, warped, mov0,
                                _pitch(mov0), self.tw, self.th)
                # texSource = warped moved frame, texTarget = reference: with the reference's
                # derivative sign (opticalFlow.cu:116-119 is MINUS the usual 5-point stencil) and
                # Iz = source - target (:131) this is the argument order for which
                # `shift += UV` (:322-323) descends; the other order diverges.
                o.ComputeDerivativesKernel(self.tw, self.th, _pitch(Ix), Ix, Iy, It, warped, _pitch(warped), self.tw,
                                           self.th, ref0, _pitch(ref0), self.tw, self.th)
                o.lucasKanadeOptim(flow, Ix, Iy, It, _pitch(flow), _pitch(Ix), self.tw, self.th, c.lkHalfWindow,
                                   float(c.lkMinDet))
            if self.flow_scale != 1:
                o.scaleFlow(flow, _pitch(flow), self.tw, self.th, float(self.flow_scale))
            mask = np.zeros((self.hh, self.hw, 4), np.float32)
            o.ComputeRobustnessMask(self.ref_half, mov_half, mask, flow, _pitch(flow), self.tw, self.th, self.hw, self.hh,
                                    _pitch(self.ref_half), _pitch(mask), float(c.alpha), float(c.beta),
                                    float(c.thresholdM))
        self.flow, self.mask = flow, mask
        o.accumulateSuperResFull(raw, img_out, total_weights, mask, self.kparam4, _pitch(self.kparam4), self.tw, self.th,
                                 flow, _pitch(flow), self.tw, self.th, self.white, self.black, self.W, self.H, c.scale,
                                 _pitch(img_out), _pitch(mask))

    def finish(self, img_out, total_weights, want16=True):
        o, c = self.o, self.c
        out = np.zeros((self.hrH, self.hrW, 3), np.float32)
        o.resampleFloat3(self.fallback, _pitch(self.fallback), self.W, self.H, out, _pitch(out), self.hrW, self.hrH, 0.0,
                         1.0, 0.0, 1.0)
        o.ApplyWeighting(out, img_out, total_weights, self.hrW, self.hrH, _pitch(out), float(c.weightThreshold))
        if c.applyGamma:
            o.GammasRGB(out, self.hrW, self.hrH, _pitch(out))
        q = None
        if want16:
            q = np.zeros((self.hrH, self.hrW, 3), np.uint16)
            o.quantize(out, _pitch(out), q, None, self.hrW, self.hrH, 65535.0)
        return out, q

    @staticmethod
    def joint_pairs(n, ref):
        """Pairs (a < b) the joint mode measures: neighbours first, then the reference against every non-neighbour
        (same order as csrc/pipeline.cpp::joint_pairs)."""
        pairs = [(k, k + 1) for k in range(n - 1)]
        for k in range(n):
            a, b = (k, ref) if k < ref else (ref, k)
            if b - a >= 2:
                pairs.append((a, b))
        return pairs

    def process_joint(self, frames):
        """Whole burst with the joint shift minimiser (stage C, ShiftMinimizerKernels.cu:81-258) between the tile tracker
        and the flow field: mirrors mfsr_burst_process_joint.  Returns (float HR image, u16 HR image)."""
        o, c = self.o, self.c
        n, ref = len(frames), c.reference
        n1 = n - 1
        prods = [self._prepare(f) for f in frames]
        pairs = self.joint_pairs(n, ref)
        m = len(pairs)
        tcx, tcy = self.tc[-1]
        tiles = tcx * tcy
        measured = np.zeros((tiles, m, 2), np.float32)
        for p, (a, b) in enumerate(pairs):
            self.ref_pyr = prods[a][1]
            sh = self._track(prods[b][1])                       # [tcy, tcx, 2]
            measured[:, p, :] = sh.reshape(tiles, 2)            # concatenateShifts (:223): pure data movement
        A = np.zeros((tiles, n1 * m), np.float32)              # column-major m x n1 per tile (:137)
        for p, (a, b) in enumerate(pairs):
            for col in range(a, b):
                A[0, p + col * m] = 1.0
        o.copyShiftMatrix(A, tiles, n, m)
        one = np.zeros((tiles, n1, 2), np.float32)
        opt = np.zeros((tiles, 2, m), np.float32)
        info = np.zeros(tiles, np.int32)
        status = np.zeros(tiles, np.int32)
        for _ in range(m + 1):
            o.solveShiftsBatched(A, measured, one, opt, info, tiles, n, m)
            o.checkForOutliers(measured, opt, A, status, info, tiles, n, m)
            if (status < 0).all():
                break
        self.joint = dict(pairs=pairs, one_to_one=one.copy(), status=status.copy(), dropped=int((A.reshape(tiles, n1, m).sum(1) == 0).sum()))
        img_out = np.zeros((self.hrH, self.hrW, 3), np.float32)
        tw = np.zeros_like(img_out)
        self.set_reference(frames[ref])
        self.flows, self.masks = [], []
        for k, f in enumerate(frames):
            given = None
            if k != ref:
                given = np.zeros((tcy, tcx, 2), np.float32)
                o.getOptimalShifts(given, one, n, tcx, tcy, _pitch(given), ref, k)
            self.add_frame(f, k == ref, img_out, tw, given_shifts=given)
            self.flows.append(self.flow)
            self.masks.append(self.mask)
        self.img_out, self.tw = img_out, tw
        return self.finish(img_out, tw)

    def process(self, frames):
        """frames: list of HxW uint16 arrays; returns (float HR image, u16 HR image)."""
        c = self.c
        img_out = np.zeros((self.hrH, self.hrW, 3), np.float32)
        tw = np.zeros_like(img_out)
        self.set_reference(frames[c.reference])
        for k, f in enumerate(frames):
            self.add_frame(f, k == c.reference, img_out, tw)
        return self.finish(img_out, tw)
