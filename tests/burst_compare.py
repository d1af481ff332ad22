"""Run one burst through the HIP pipeline (C-ABI mfsr_burst_*) and through the CPU oracle pipeline, keeping every
frame's flow field and robustness mask, and compare the two with the flip-set classification of tests/flipset.py.

Used by the -m gpu pipeline tests and by bench.py's cpu_baseline leg (the oracle is the checker there)."""
from __future__ import annotations

import numpy as np


def run_hip(cfg, frames, device="cuda:0", per_frame=True):
    """frames: list of [H, W] int16/uint16 torch tensors (any device).  Returns numpy results."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, view_as_tensor
    dev = torch.device(device)
    pipe = BurstPipeline(cfg, dev)
    dframes = [f.to(dev) for f in frames]
    n = len(dframes)
    flows, masks = [None] * n, [None] * n
    pipe.begin_burst()
    ref = cfg.reference
    pipe.set_reference(dframes[ref])
    group = max(pipe.group_size(), 1)
    unread = []
    for k in range(n):
        pipe.add_frame(dframes[k], k == ref)
        unread.append(k)
        # a frame is aligned when its fuse group is complete (frame-batched alignment) or on flush -- which finish() does
        # for the last, partial group anyway, so flushing there changes nothing of the burst
        if (k + 1) % group == 0 or k == n - 1:
            if k == n - 1:
                pipe.flush()
            for j in (unread if per_frame else unread[-1:]):
                flow_t, mask_t = pipe.frame_views(k - j)
                flows[j] = view_as_tensor(flow_t, 2, dev).cpu().numpy()
                masks[j] = view_as_tensor(mask_t, 4, dev).cpu().numpy()
            unread = []
    if not per_frame:
        flows, masks = [flows[-1]], [masks[-1]]
    out, out16 = pipe.finish()
    torch.cuda.synchronize()
    _, _, kp_t, trk_t = pipe.debug_views()
    res = dict(out=out.cpu().numpy(), out16=out16.cpu().numpy().view(np.uint16), img_out=pipe.img_out.cpu().numpy(),
               tw=pipe.total_weights.cpu().numpy(), flows=flows, masks=masks, flow=flows[-1], mask=masks[-1],
               kparam=view_as_tensor(kp_t, 4, dev).cpu().numpy(), tracking=view_as_tensor(trk_t, 1, dev).cpu().numpy()[..., 0])
    pipe.close()
    return res


def run_oracle(cfg, frames):
    """frames: list of [H, W] 16-bit torch tensors or numpy arrays."""
    from oracle.pipeline import OraclePipeline
    op = OraclePipeline(cfg)
    nf = [(f.cpu().numpy() if hasattr(f, "cpu") else f).view(np.uint16) for f in frames]
    img_out = np.zeros((op.hrH, op.hrW, 3), np.float32)
    tw = np.zeros_like(img_out)
    op.set_reference(nf[cfg.reference])
    flows, masks = [], []
    for k, f in enumerate(nf):
        op.add_frame(f, k == cfg.reference, img_out, tw)
        flows.append(op.flow)
        masks.append(op.mask)
    out, q = op.finish(img_out, tw)
    return dict(out=out, out16=q, img_out=img_out, tw=tw, flows=flows, masks=masks, flow=flows[-1], mask=masks[-1],
                kparam=op.kparam4, tracking=op.ref_pyr[0])


def psnr(a, b, peak=1.0):
    if a.ndim >= 2 and a.shape[0] >= 1024:      # HR images: squared error by row bands on the thread pool (float64 sums)
        from tests.flipset import map_row_bands
        sq = map_row_bands(lambda y0, y1: float(((a[y0:y1].astype(np.float64) - b[y0:y1].astype(np.float64)) ** 2).sum()), a.shape[0])
        mse = sum(sq) / a.size
    else:
        mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 200.0 if mse == 0 else float(10 * np.log10(peak * peak / mse))


def classify(cfg, h, o):
    """Flip-set report of a HIP result against an oracle result (both from run_* with per-frame products)."""
    from tests.flipset import FlipSet
    fs = FlipSet(cfg)
    assert len(h["flows"]) == len(o["flows"])
    for k in range(len(h["flows"])):
        if k == cfg.reference:
            continue   # identity flow, certainty 1 on both sides
        fs.add_frame(h["flows"][k], o["flows"][k], h["masks"][k], o["masks"][k])
    fs.add_weights(h["tw"], o["tw"])
    rep = fs.report(h["out"], o["out"], h["out16"], o["out16"])
    rep["psnr_db_vs_oracle"] = round(psnr(h["out"], o["out"]), 2)
    rep["_flips"] = fs.flips          # the set itself (HR mask)
    rep.update(continuous_checks(fs.flips, h, o))
    return rep


ACC_RTOL, ACC_ATOL = 3e-5, 1e-6       # the kernel-level tolerance of the accumulate kernels (tests/test_parity_kernels.py)
TAU_WEIGHT = 0.03                     # total weight from which a u16 sample must be within 1 LSB16
EXCUSED_SLOPE = 0.05                  # below it: |d16| <= 1 + EXCUSED_SLOPE / weight


def continuous_checks(flips, h, o):
    """What must hold OUTSIDE the flip set for the continuous quantities (no decision differs there, so nothing may differ
    by more than rounding):
      * accumulators: |d imgOut| <= ACC_RTOL |imgOut| + ACC_ATOL, same for totalWeights;
      * u16 image (the library's own output depth): every sample <= 1 LSB16 where its channel's total weight >= TAU_WEIGHT.
        Below that weight the output is a ratio of two small sums whose absolute errors (~1e-6: 1 ulp of a v_exp_f32
        weight, 3e-7 of a certainty) no longer vanish against them: those pixels are excused from the 1-LSB16 bound,
        counted, and bounded by 1 + EXCUSED_SLOPE / weight instead.
    Evaluated in row bands on a thread pool (tests/flipset.py::map_row_bands): the same per-sample expressions; the printed
    p99.99 of the relative accumulator difference is taken over every 8th sample outside the set (a diagnostic, not asserted)."""
    from tests.flipset import map_row_bands
    PCT_STRIDE = 8

    def band(y0, y1):
        out = ~flips[y0:y1]
        q = {}
        worst, n_viol = 0.0, 0
        for name in ("img_out", "tw"):
            a, b = h[name][y0:y1].astype(np.float64), o[name][y0:y1].astype(np.float64)
            diff = np.abs(a - b)
            excess = diff - (ACC_RTOL * np.abs(b) + ACC_ATOL)
            excess = np.where(np.isnan(excess), np.inf, excess)[out]
            n_viol += int((excess > 0).sum())
            worst = max(worst, float(excess.max()) if excess.size else 0.0)
            q["rel_" + name] = (diff / (np.abs(b) + 1e-30))[out].ravel()[::PCT_STRIDE]
        q["n_viol"], q["worst"] = n_viol, worst
        # per SAMPLE: a channel is judged by its own total weight
        th, to = h["tw"][y0:y1], o["tw"][y0:y1]
        d16 = np.abs(h["out16"][y0:y1].astype(np.int64) - o["out16"][y0:y1].astype(np.int64))
        w = np.minimum(th, to).astype(np.float64)
        w[(th == 0) & (to == 0)] = np.inf      # a channel nothing was fused into (mono: G, B) is exact, not "light"
        out3 = np.broadcast_to(out[..., None], d16.shape)
        well = out3 & (w >= TAU_WEIGHT)
        exc = out3 & ~(w >= TAU_WEIGHT)
        dw, de = d16[well], d16[exc]
        q["max_well"] = int(dw.max()) if dw.size else 0
        q["n_gt1_well"] = int((dw > 1).sum())
        q["n_exc"] = int(exc.sum())
        q["max_exc"] = int(de.max()) if de.size else 0
        bound = 1.0 + EXCUSED_SLOPE / np.maximum(w, 1e-30)
        q["n_exc_over"] = int((de > bound[exc]).sum())
        return q

    parts = map_row_bands(band, flips.shape[0])
    r = {}
    for name in ("img_out", "tw"):
        rel = np.concatenate([q["rel_" + name] for q in parts])
        r[f"acc_rel_p9999_{name}"] = float(np.percentile(rel, 99.99)) if rel.size else 0.0
    r["n_acc_violations_outside"] = sum(q["n_viol"] for q in parts)
    r["acc_worst_excess"] = max(q["worst"] for q in parts)
    r["max16_outside_well_weighted"] = max(q["max_well"] for q in parts)
    r["n_gt1_16bit_outside_well_weighted"] = sum(q["n_gt1_well"] for q in parts)
    r["excused_fraction"] = float(sum(q["n_exc"] for q in parts)) / float(h["out16"].size)
    r["max16_excused"] = max(q["max_exc"] for q in parts)
    r["n_excused_over_bound"] = sum(q["n_exc_over"] for q in parts)
    return r


def assert_parity(rep, what, max_flip_fraction=2e-2):
    """The +-1 LSB contract.  Outside the flip set: NO 8-bit sample is off by more than 1 LSB (the CLI's output depth,
    multi_frame_sr.cpp:207); the accumulators agree to the kernel-level tolerance; NO u16 sample (the library's output
    depth) is off by more than 1 LSB16 where its channel's total weight is at least TAU_WEIGHT (the rest is counted and
    bounded, see continuous_checks).  The flip set is small; inside it the error is what one mis-rounded tap can do."""
    print(f"[{what}] PSNR vs oracle {rep['psnr_db_vs_oracle']:.1f} dB; flip set {rep['flip_fraction']:.2e} of the pixels "
          f"({rep['flip_fraction_no_guard']:.2e} without the tie guard) "
          f"(max flow diff {rep['max_flow_diff_px']:.1e} px); >1 LSB 8-bit: outside {rep['n_gt1_8bit_outside']} "
          f"(max {rep['max8_outside']}), inside {rep['n_gt1_8bit_inside']} (max {rep['max8_inside']}); "
          f"16-bit outside: max {rep['max16_outside_well_weighted']} where weight >= {TAU_WEIGHT} "
          f"(excused {rep['excused_fraction']:.2e} of the samples, max {rep['max16_excused']}, over their bound {rep['n_excused_over_bound']}), "
          f"max inside {rep['max16_inside']}; accumulators outside: {rep['n_acc_violations_outside']} beyond {ACC_RTOL:g} rel + {ACC_ATOL:g} "
          f"(p99.99 rel {rep['acc_rel_p9999_img_out']:.1e} / {rep['acc_rel_p9999_tw']:.1e}); causes/frame {rep['flips_by_cause_per_frame']}")
    assert rep["psnr_db_vs_oracle"] >= 70.0
    assert rep["n_gt1_8bit_outside"] == 0, "a sample outside the flip set is off by more than 1 LSB"
    assert rep["max8_outside"] <= 1
    assert rep["flip_fraction"] <= max_flip_fraction
    assert rep["frac_gt1_8bit"] <= 1e-3
    assert rep["n_acc_violations_outside"] == 0, f"accumulators differ outside the flip set (worst excess {rep['acc_worst_excess']:.2e})"
    assert rep["max16_outside_well_weighted"] <= 1, "a well-weighted u16 sample outside the flip set is off by more than 1 LSB16"
    assert rep["n_excused_over_bound"] == 0
    assert rep["excused_fraction"] <= 0.25    # (x4 from few frames: many HR samples far from every raw sample of their colour)


def _box(a, r):
    """(2r+1)^2 box sums, zero beyond the border (float64)."""
    H, W = a.shape
    c = np.zeros((H + 1, W + 1), np.float64)
    c[1:, 1:] = np.cumsum(np.cumsum(a.astype(np.float64), 0), 1)
    y0, y1 = np.clip(np.arange(H) - r, 0, H), np.clip(np.arange(H) + r + 1, 0, H)
    x0, x1 = np.clip(np.arange(W) - r, 0, W), np.clip(np.arange(W) + r + 1, 0, W)
    return c[y1][:, x1] - c[y0][:, x1] - c[y1][:, x0] + c[y0][:, x0]


def flow_conditioning(img, half_window):
    """(sigma1, sigma2): larger and smaller singular value of the Lucas-Kanade normal matrix of every pixel's window,
    from the 5-point derivative of `img` (the reference tracking image; opticalFlow.cu:116-119,219-233,250-253).  The
    reference tests only sigma1 against lkMinDet (:255, `fminf(sigma1, sigma1)`): sigma2 may be arbitrarily small."""
    p = np.pad(img.astype(np.float64), 2, mode="reflect")
    ix = (p[2:-2, :-4] - 8 * p[2:-2, 1:-3] + 8 * p[2:-2, 3:-1] - p[2:-2, 4:]) / 12.0
    iy = (p[:-4, 2:-2] - 8 * p[1:-3, 2:-2] + 8 * p[3:-1, 2:-2] - p[4:, 2:-2]) / 12.0
    a, b, d = _box(ix * ix, half_window), _box(ix * iy, half_window), _box(iy * iy, half_window)
    half = 0.5 * (a + d)
    root = np.sqrt(np.maximum(0.25 * (a - d) ** 2 + b * b, 0.0))
    return half + root, np.maximum(half - root, 0.0)


def exponent_conditioning(kparam, hrH, hrW):
    """kappa per HR pixel = 4|k.x| + 4|k.y| + 8|k.z| (the magnitude of the terms of the tap exponent
    px^2 k.x + 2 px py k.z + py^2 k.y, |px|,|py| <= 2; DeBayerKernels.cu:427), maximum over the 3x3 field texels a
    bilinear lookup at the pixel can touch, field upsampled to the HR grid."""
    k = np.abs(kparam[..., 0]) * 4.0 + np.abs(kparam[..., 1]) * 4.0 + np.abs(kparam[..., 2]) * 8.0
    k = np.nan_to_num(k, nan=np.inf, posinf=np.inf)
    p = np.pad(k, 1, mode="edge")
    m = k.copy()
    for dy in range(3):
        for dx in range(3):
            m = np.maximum(m, p[dy:dy + k.shape[0], dx:dx + k.shape[1]])
    fy, fx = hrH // k.shape[0], hrW // k.shape[1]
    return np.repeat(np.repeat(m, fy, 0), fx, 1)[:hrH, :hrW]


_COND_CACHE = {}


def flow_difference_report(flow_h, flow_o, tracking, half_window, thr=2e-4):
    """Where do the two implementations' flows differ by more than `thr` px?  Lucas-Kanade multiplies rounding noise by
    1 / sigma2 (the smaller singular value of the window's normal matrix; the reference only tests the larger one,
    opticalFlow.cu:255) and does not converge where the flow is wild (border rows under MIRROR addressing, occlusions).
    For a burst of pure translations: `well` = interior pixels (>= 8 px from the border) whose window has sigma2 above the
    image's 20th percentile and whose flow is within 1 px of the frame's median flow.  Returns the maximum difference over
    `well`, over the rest, and the fraction of the > thr differences that sit in the rest."""
    d = np.abs(flow_h - flow_o).max(-1)
    # (every frame of a burst is tracked against the same reference image: its conditioning once)
    key = (id(tracking), tracking.shape, half_window)
    if _COND_CACHE.get("key") != key:
        _COND_CACHE.clear()
        _COND_CACHE.update(key=key, keep=tracking, val=flow_conditioning(tracking, half_window))
    s1, s2 = _COND_CACHE["val"]
    if s2.shape != d.shape:     # flow stored at another resolution than the tracking image: nearest
        fy, fx = d.shape[0] / s2.shape[0], d.shape[1] / s2.shape[1]
        yy = np.minimum((np.arange(d.shape[0]) / fy).astype(int), s2.shape[0] - 1)
        xx = np.minimum((np.arange(d.shape[1]) / fx).astype(int), s2.shape[1] - 1)
        s2 = s2[yy][:, xx]
    med = np.median(flow_o.reshape(-1, 2), 0)
    dev = np.abs(flow_o - med).max(-1)
    yy, xx = np.mgrid[0:d.shape[0], 0:d.shape[1]]
    border = np.minimum(np.minimum(yy, d.shape[0] - 1 - yy), np.minimum(xx, d.shape[1] - 1 - xx))
    q = np.percentile(s2, 20)
    well = (s2 >= q) & (dev <= 1.0) & (border >= 8)
    big = d > thr
    return {
        "max_well": float(d[well].max()) if well.any() else 0.0,
        "max_rest": float(d[~well].max()) if (~well).any() else 0.0,
        "well_fraction": float(well.mean()),
        "n_big": int(big.sum()),
        "big_in_rest_fraction": float((big & ~well).sum()) / max(int(big.sum()), 1),
        "sigma2_p20": float(q),
    }
