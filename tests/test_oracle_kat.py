"""Known-answer tests that pin the CPU oracle (CPU only).

The reference ships no tests, golden vectors or expected outputs for this path
(SURVEY.md section 4), so the oracle cannot be pinned against the reference's own
fixtures -- "parity unpinned".  These KATs are derived by hand from the formulae
in the reference sources (file:line cited per test) and from closed-form
properties (a quadratic surface is fitted exactly, a constant image debayers to
itself, Lucas-Kanade must converge on a known translation, ...).
"""
import numpy as np
import pytest

from tests.kernels import F2, F3, Tex, pitch_of

RGGB = [0, 1, 1, 2]


def test_deBayersSubSample3_ramp(orc):
    # DeBayerKernels.cu:244-283: R,B = raw/maxVal; G = sum of the two greens * 0.5 / maxVal
    orc.set_cfa(RGGB)
    raw = np.arange(64, dtype=np.uint16).reshape(8, 8) * 37
    out = np.zeros((4, 4, 3), np.float32)
    orc.call("deBayersSubSample3", raw, out, 4095.0, 4, 4, pitch_of(out))
    for y in range(4):
        for x in range(4):
            r, g1, g2, b = raw[2 * y, 2 * x], raw[2 * y, 2 * x + 1], raw[2 * y + 1, 2 * x], raw[2 * y + 1, 2 * x + 1]
            np.testing.assert_allclose(out[y, x], [r / 4095, (int(g1) + int(g2)) / 2 / 4095, b / 4095], rtol=3e-7)
    # SURVEY.md section 8c probe value: G of quad (0,0) with greens 37 and 296
    assert abs(out[0, 0, 1] - (37 + 296) / 2 / 4095) < 1e-7


@pytest.mark.parametrize("pat", [[0, 1, 1, 2], [2, 1, 1, 0], [1, 0, 2, 1], [1, 2, 0, 1]])
def test_debayer_of_flat_colour_is_that_colour(orc, pat):
    # DeBayerKernels.cu:55-231: every interpolation is a convex/affine combination that
    # reproduces constants, so a mosaic of a flat colour debayers to the colour itself.
    orc.set_cfa(pat)
    H, W = 16, 20
    colour = np.array([0.25, 0.5, 0.75], np.float32)
    black = np.array([64, 64, 64], np.float32)
    white = np.array([1000, 1000, 1000], np.float32)
    raw = np.zeros((H, W), np.float32)
    for y in range(H):
        for x in range(W):
            c = pat[(y % 2) * 2 + (x % 2)]
            raw[y, x] = colour[c] * white[c] + black[c]
    out = np.zeros((H, W, 3), np.float32)
    sc = (1.0 / white).astype(np.float32)
    orc.call("deBayerGreenKernel", W, H, raw, pitch_of(raw), out, pitch_of(out), F3(black), F3(sc))
    orc.call("deBayerRedBlueKernel", W, H, raw, pitch_of(raw), out, pitch_of(out), F3(black), F3(sc))
    np.testing.assert_allclose(out[3:-3, 3:-3], np.broadcast_to(colour, (H - 6, W - 6, 3)), atol=2e-6)
    assert (out[:2] == 0).all() and (out[:, -2:] == 0).all()  # 2-px ring untouched (:61-62)


def test_findMinimum_recovers_analytic_minimum(orc):
    # kernel.cu:503-633: the 3x3 stencils fit a quadratic exactly -> sub-pixel minimum is exact
    S = 4
    R = 2 * S + 1
    y, x = np.mgrid[0:R, 0:R].astype(np.float64)
    cx, cy = 1.3, -0.6
    img = (1.5 * (x - S - cx) ** 2 + 0.8 * (y - S - cy) ** 2 + 0.3 * (x - S - cx) * (y - S - cy) + 2.0).astype(np.float32)
    out = np.zeros((1, 1, 2), np.float32)
    orc.call("findMinimum", img[None].copy(), out, 8, S, 1, 1, 0.0)
    np.testing.assert_allclose(out[0, 0], [cx, cy], atol=2e-5)
    # flat tile: threshold + min > max -> (0,0) (:629-633)
    orc.call("findMinimum", np.ones((1, R, R), np.float32), out, 8, S, 1, 1, 0.5)
    assert (out == 0).all()
    # minimum on the border ring -> (0,0) (:548-553)
    img2 = img.copy()
    img2[0, 3] = -100
    orc.call("findMinimum", img2[None].copy(), out, 8, S, 1, 1, 0.0)
    assert (out == 0).all()


def test_tile_chain_is_the_sum_of_squared_differences(orc):
    # kernel.cu:119-259 + direct correlation: D(s) = sum (ref(p) - moved(p+s))^2
    T, S, n = 8, 2, 3
    L, R = T + 2 * S, 2 * S + 1
    r = np.random.default_rng(0)
    mt = r.random((n, L, L), dtype=np.float32)
    rt = np.zeros_like(mt)
    rt[:, S:S + T, S:S + T] = r.random((n, T, T), dtype=np.float32)
    cc, bx, by = np.zeros_like(mt), np.zeros_like(mt), np.zeros_like(mt)
    sq = np.zeros(n, np.float32)
    dist = np.zeros((n, R, R), np.float32)
    orc.call("crossCorrelateTiles", rt, mt, cc, S, T, n)
    orc.call("squaredSum", rt, sq, S, T, n)
    orc.call("boxFilterWithBorderX", mt, bx, S, T, n)
    orc.call("boxFilterWithBorderY", bx, by, S, T, n)
    orc.call("normalizedCC", cc, sq, by, dist, S, T, n)
    for t in range(n):
        for sy in range(-S, S + 1):
            for sx in range(-S, S + 1):
                ref = rt[t, S:S + T, S:S + T].astype(np.float64)
                mov = mt[t, S + sy:S + sy + T, S + sx:S + sx + T].astype(np.float64)
                assert abs(dist[t, sy + S, sx + S] - ((ref - mov) ** 2).sum()) < 1e-4
    # SURVEY.md section 8c probe: boxFilterWithBorderX at a valid column = direct sum of squares
    assert abs(bx[0, 0, T // 2] - (mt[0, 0, 0:T].astype(np.float64) ** 2).sum()) < 1e-5


def test_gaussin_filter_1D_values(orc):
    # test_opencv/main.cpp:370-391
    t = np.zeros(99, np.float32)
    assert orc.o.gaussin_filter_1D(0.5, t) == 3     # (int)(0.5/0.6-0.4)*2+3
    e = np.exp(-2.0)
    np.testing.assert_allclose(t[:3], [e / (1 + 2 * e), 1 / (1 + 2 * e), e / (1 + 2 * e)], rtol=1e-6)
    assert orc.o.gaussin_filter_1D(0.0, t) == 9 and t[4] == 1 and t[:9].sum() == 1   # delta (:371-373)
    assert orc.o.gaussin_filter_1D(1000.0, t) == 99                                    # clamp (:375)
    n = orc.o.gaussin_filter_1D(2.0, t)
    assert n == 7 and abs(t[:n].sum() - 1) < 1e-6 and np.allclose(t[:n], t[:n][::-1])


def test_texture_semantics(orc):
    # pixel-centre fetch returns the texel; MIRROR reflects; CLAMP repeats the edge
    H, W = 6, 9
    img = np.random.default_rng(1).random((H, W), dtype=np.float32)
    out = np.zeros((H, W), np.float32)
    zero = np.zeros((H, W, 2), np.float32)
    orc.call("WarpingKernel", W, H, pitch_of(out), Tex(zero), out, Tex(img))      # opticalFlow.cu:28-44
    np.testing.assert_allclose(out, img, atol=1e-6)
    uv = zero.copy()
    uv[..., 0] = 1.0                                                               # out(x) = img(x+1), mirror at the end
    orc.call("WarpingKernel", W, H, pitch_of(out), Tex(uv), out, Tex(img))
    np.testing.assert_allclose(out[:, :-1], img[:, 1:], atol=1e-6)
    np.testing.assert_allclose(out[:, -1], img[:, -1], atol=1e-6)                  # reflection of W is W-1
    uv[..., 0] = 0.5
    orc.call("WarpingKernel", W, H, pitch_of(out), Tex(uv), out, Tex(img))
    np.testing.assert_allclose(out[:, :-1], 0.5 * (img[:, :-1] + img[:, 1:]), atol=1e-6)


def test_derivative_sign_and_lk_convergence(orc):
    """opticalFlow.cu:116-119 is MINUS the usual 5-point derivative and Iz = source - target
    (:131): with texSource = warped moved frame the Lucas-Kanade update (:190-325) converges
    to the true translation; with the roles swapped it diverges.  This pins the argument
    order used by the pipeline."""
    H, W, hw = 64, 96, 3
    y, x = np.mgrid[0:H + 8, 0:W + 8].astype(np.float64)
    f = lambda xx, yy: 0.5 + 0.2 * np.sin(0.31 * xx + 0.12 * yy) + 0.2 * np.cos(0.23 * yy - 0.05 * xx) + 0.1 * np.sin(0.11 * xx * 0.5)
    ref = f(x[4:-4, 4:-4], y[4:-4, 4:-4]).astype(np.float32)
    d = np.array([0.37, -0.52])
    mov = f(x[4:-4, 4:-4] - d[0], y[4:-4, 4:-4] - d[1]).astype(np.float32)   # mov(p + d) = ref(p)
    Ix, Iy, Iz = (np.zeros((H, W), np.float32) for _ in range(3))
    orc.call("ComputeDerivatives2Kernel", W, H, pitch_of(Ix), Ix, Iy, Tex(ref))
    true_dx = 0.2 * 0.31 * np.cos(0.31 * x[4:-4, 4:-4] + 0.12 * y[4:-4, 4:-4]) + 0.2 * 0.05 * np.sin(
        0.23 * y[4:-4, 4:-4] - 0.05 * x[4:-4, 4:-4]) + 0.1 * 0.055 * np.cos(0.055 * x[4:-4, 4:-4])
    np.testing.assert_allclose(Ix[4:-4, 4:-4], -true_dx[4:-4, 4:-4], atol=2e-4)   # MINUS d/dx

    def iterate(source_is_warped):
        flow = np.zeros((H, W, 2), np.float32)
        warped = np.zeros((H, W), np.float32)
        for _ in range(4):
            orc.call("WarpingKernel", W, H, pitch_of(warped), Tex(flow), warped, Tex(mov))
            a, b = (warped, ref) if source_is_warped else (ref, warped)
            orc.call("ComputeDerivativesKernel", W, H, pitch_of(Ix), Ix, Iy, Iz, Tex(a), Tex(b))
            orc.call("lucasKanadeOptim", flow, Ix, Iy, Iz, pitch_of(flow), pitch_of(Ix), W, H, hw, 1e-6)
        return flow[10:-10, 10:-10].reshape(-1, 2)

    good = iterate(True)
    np.testing.assert_allclose(np.median(good, 0), d, atol=0.01)
    bad = iterate(False)
    assert np.abs(np.median(bad, 0) - d).max() > 0.3


def test_accumulate_identity_burst(orc):
    # DeBayerKernels.cu:379-468 generalised: flat raw, zero flow, certainty 1 ->
    # imgOut / totalWeights == (raw - black) / white for every channel that received taps
    orc.set_cfa(RGGB)
    W, H, s = 32, 24, 2
    raw = np.full((H, W), 1256, np.uint16)
    img = np.zeros((H * s, W * s, 3), np.float32)
    tw = np.zeros_like(img)
    mask = np.ones((H // 2, W // 2, 4), np.float32)
    kp = np.zeros((H // 2, W // 2, 4), np.float32)
    kp[..., 0] = kp[..., 1] = 1.0
    flow = np.zeros((H // 2, W // 2, 2), np.float32)
    white, black = F3([3839, 3839, 3839]), F3([256, 256, 256])
    orc.call("accumulateSuperResFull", raw, img, tw, mask, Tex(kp), Tex(flow), white, black, W, H, s, pitch_of(img),
             pitch_of(mask))
    inner = (slice(1, -1), slice(1, -1))
    assert (tw[inner] > 0).all()                       # 5x5 HR taps always reach R, G and B sites at s = 2
    np.testing.assert_allclose(img[inner] / tw[inner], (1256 - 256) / 3839, rtol=1e-6)
    assert (tw[0] == 0).all() and (tw[:, 0] == 0).all()   # 1-px ring untouched (:391)
    # total weight of green = sum over the taps landing on green sites of exp(-(px^2+py^2)/2)
    # HR pixel (2,2): X+px in 0..4 -> raw cols 0,0,1,1,2 (parity E,E,O,O,E), same for rows.
    w1 = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    even = w1[[0, 1, 4]].sum()
    odd = w1[[2, 3]].sum()
    np.testing.assert_allclose(tw[2, 2], [even * even, 2 * even * odd, odd * odd], rtol=1e-6)   # R=EE, G=EO+OE, B=OO


def test_non_finite_weight_rule(orc):
    # DeBayerKernels.cu:429-430: non-finite w -> 1 on the axes (px*py == 0), else 0
    orc.set_cfa(RGGB)
    W, H, s = 16, 16, 2
    raw = np.full((H, W), 1256, np.uint16)
    img = np.zeros((H * s, W * s, 3), np.float32)
    tw = np.zeros_like(img)
    mask = np.ones((H // 2, W // 2, 4), np.float32)
    kp = np.full((H // 2, W // 2, 4), np.nan, np.float32)
    flow = np.zeros((H // 2, W // 2, 2), np.float32)
    orc.call("accumulateSuperResFull", raw, img, tw, mask, Tex(kp), Tex(flow), F3([3839] * 3), F3([256] * 3), W, H, s,
             pitch_of(img), pitch_of(mask))
    # only the 9 axis taps survive, each with weight 1.  HR pixel (2,2): X+px, Y+py in 0..4 -> raw
    # cols/rows 0,0,1,1,2 (parity E,E,O,O,E).  Row taps (py=0 -> raw row 1, odd): cols E,E,O,O,E ->
    # G,G,B,B,G.  Column taps (px=0 -> raw col 1, odd), centre excluded: rows E,E,O,E -> G,G,B,G.
    np.testing.assert_array_equal(tw[2, 2], [0.0, 6.0, 3.0])
    assert tw[2, 2].sum() == 9.0


def test_ApplyWeighting_and_gamma(orc):
    # kernel.cu:426-481
    io = np.array([[[0.2, 0.4, 0.6]]], np.float32)
    fin = np.array([[[1.0, 0.0, 3.0]]], np.float32)
    wt = np.array([[[2.0, 0.0, -1.0]]], np.float32)
    orc.call("ApplyWeighting", io, fin, wt, 1, 1, 12, 0.5)
    # ch0: w >= thr -> 1/2; ch1: w < thr -> (0+0.4)/(0+1); ch2: w=-1 < thr -> w+1 == 0 -> 0
    np.testing.assert_allclose(io[0, 0], [0.5, 0.4, 0.0], rtol=1e-7)
    g = np.array([[[0.5, 0.002, 1.7]], [[np.nan, -1.0, 0.0031308]]], np.float32)
    orc.call("GammasRGB", g, 1, 2, 12)                                            # kernel.cu:380-422
    np.testing.assert_allclose(g[0, 0], [1.055 * 0.5 ** (1 / 2.4) - 0.055, 12.92 * 0.002, 1.0], rtol=1e-6)
    np.testing.assert_allclose(g[1, 0], [0.0, 0.0, 12.92 * 0.0031308], atol=1e-7)


def test_kernel_param_is_the_inverse_covariance(orc):
    # kernel.cu:718-790: output = (B^-1)_{00}, (B^-1)_{11}, (B^-1)_{01} of B = k1 e1 e1^T + k2 e2 e2^T
    Dth, Dtr, kDetail, kDenoise, kStretch, kShrink = 0.005, 0.05, 0.3, 2.0, 2.0, 2.0
    t = np.array([[[4e-4, 1e-4, 1e-4], [1e-4, 9e-4, -2e-4], [2.5e-4, 2.5e-4, 0.0]]], np.float32)
    exp = []
    for a11, a22, a12 in t[0].astype(np.float64):
        help_ = np.sqrt((a22 - a11) ** 2 + 4 * a12 * a12)
        c, s = 2 * a12, a22 - a11 + help_
        n = np.hypot(c, s)
        c, s = (c / n, s / n) if n > 0 else (1.0, 0.0)
        l1, l2 = (a11 + a22 + help_) / 2, (a11 + a22 - help_) / 2
        A = 1 + np.sqrt((l1 - l2) ** 2 / (l1 + l2) ** 2)
        D = min(max(1 - np.sqrt(l1) / Dtr + Dth, 0), 1)
        k1 = ((1 - D) * kDetail * kStretch * A + D * kDetail * kDenoise) ** 2
        k2 = ((1 - D) * kDetail / kShrink * A + D * kDetail * kDenoise) ** 2
        e1, e2 = np.array([s, -c]), np.array([c, s])
        B = k1 * np.outer(e1, e1) + k2 * np.outer(e2, e2)
        Bi = np.linalg.inv(B)
        exp.append([Bi[0, 0], Bi[1, 1], Bi[0, 1]])
    k = t.copy()
    orc.call("ComputeKernelParam", k, 3, 1, pitch_of(k), Dth, Dtr, kDetail, kDenoise, kStretch, kShrink)
    np.testing.assert_allclose(k[0], exp, rtol=2e-4)


def test_shift_minimiser_exact_on_consistent_measurements(orc):
    # ShiftMinimizerKernels.cu:81-139,179-218 + the batched solve (C4)
    n_img, tiles = 5, 4
    pairs = [(a, b) for a in range(n_img) for b in range(a + 1, n_img)]
    n1, m = n_img - 1, len(pairs)
    A = np.zeros((tiles, n1, m), np.float32)
    d = np.array([[1, -2], [0.5, 0.25], [-3, 4], [2, 2]], np.float32)
    meas = np.zeros((tiles, m, 2), np.float32)
    for k, (a, b) in enumerate(pairs):
        A[:, a:b, k] = 1
        meas[:, k] = d[a:b].sum(0)
    one = np.zeros((tiles, n1, 2), np.float32)
    opt = np.zeros((tiles, 2, m), np.float32)
    info = np.zeros(tiles, np.int32)
    status = np.zeros(tiles, np.int32)
    orc.call("solveShiftsBatched", A, meas, one, opt, info, tiles, n_img, m)
    orc.call("checkForOutliers", meas, opt, A, status, info, tiles, n_img, m)
    assert (info == 0).all() and (status == -1).all()
    np.testing.assert_allclose(one, np.broadcast_to(d, (tiles, n1, 2)), atol=1e-5)
    out = np.zeros((2, 2, 2), np.float32)
    orc.call("getOptimalShifts", out, one, n_img, 2, 2, pitch_of(out), 1, 4)      # ref 1 -> image 4: d1+d2+d3
    np.testing.assert_allclose(out[0, 0], d[1:4].sum(0), atol=1e-5)
    orc.call("getOptimalShifts", out, one, n_img, 2, 2, pitch_of(out), 3, 0)      # backwards: -(d0+d1+d2)
    np.testing.assert_allclose(out[1, 1], -d[0:3].sum(0), atol=1e-5)


def test_UpSampleShifts_scales_a_constant_field(orc):
    # kernel.cu:642-688: a constant field stays constant, multiplied by oldLevel/newLevel
    inS = np.tile(np.array([1.5, -0.75], np.float32), (4, 6, 1))
    out = np.zeros((8, 12, 2), np.float32)
    orc.call("UpSampleShifts", inS, out, pitch_of(inS), pitch_of(out), 4, 2, 6, 4, 12, 8, 16, 16)
    np.testing.assert_allclose(out, np.broadcast_to(np.array([3.0, -1.5], np.float32), out.shape), rtol=1e-7)


def test_sharpenImg2_small(orc):
    # multi_frame_sr.cpp:90-119: result byte j (row r) = sat(5*c[j+ch] - c[j] - c[j+2ch] - p[j+ch] - n[j+ch]); ring = 0
    img = (np.arange(5 * 6).reshape(5, 6, 1) * 7 % 256).astype(np.uint8)
    out = np.full_like(img, 9)
    orc.call("sharpenImg2", img, out, 5, 6, 1, 6, 6)
    assert (out[0] == 0).all() and (out[-1] == 0).all() and (out[:, 0] == 0).all() and (out[:, -1] == 0).all()
    i = img.astype(np.int32)[..., 0]
    for r in range(1, 4):
        for j in range(1, 4):
            v = 5 * i[r, j + 1] - i[r, j] - i[r, j + 2] - i[r - 1, j + 1] - i[r + 1, j + 1]
            assert out[r, j, 0] == min(max(v, 0), 255)
    assert (out[1:4, 4] == 0).all()   # never written by the reference loop (defined as 0)


def test_robustness_mask_limits(orc):
    # RobustnessModell.cu:29-158: identical images -> d = 0 -> mask = clamp(1.5 - 0.12, 0, 1) = 1;
    # very different images -> exp(-big) -> 0; M > thresholdM forces s = 0 -> 0
    H, W = 12, 16
    ref = np.random.default_rng(3).random((H, W, 3), dtype=np.float32) * 0.5 + 0.25
    uv = np.zeros((H, W, 2), np.float32)
    mask = np.zeros((H, W, 4), np.float32)
    orc.call("ComputeRobustnessMask", ref, ref.copy(), mask, Tex(uv), W, H, pitch_of(ref), pitch_of(mask), 1e-4, 1e-6, 1.0)
    assert (mask[1:-1, 1:-1, :3] == 1.0).all() and (mask[0] == 0).all()
    # textured reference (std ~0.14 >> sigma_md ~0.007), moved brighter by 0.5: d/sigma ~ 3.5 -> 0
    orc.call("ComputeRobustnessMask", ref, ref + 0.5, mask, Tex(uv), W, H, pitch_of(ref), pitch_of(mask), 1e-4, 1e-6, 1.0)
    assert (mask[1:-1, 1:-1, :3] == 0.0).all()
    # flat reference: std = 0 -> the Wiener factor std^2/(std^2+sigma_md^2) shrinks d to 0 -> 1 (:142-144)
    flat = np.full((H, W, 3), 0.2, np.float32)
    orc.call("ComputeRobustnessMask", flat, flat + 0.5, mask, Tex(uv), W, H, pitch_of(ref), pitch_of(mask), 1e-4, 1e-6, 1.0)
    assert (mask[1:-1, 1:-1, :3] == 1.0).all()
    # large local flow range x mean distance > thresholdM -> s = 0 -> mask 0 (:148-149); .w = M
    uv2 = np.zeros((H, W, 2), np.float32)
    uv2[:, (np.arange(W) % 4) < 2, 0] = 6.0   # flow(x) != flow(x+2): the only pair the quirk compares (:62-72)
    orc.call("ComputeRobustnessMask", ref, ref + 0.3, mask, Tex(uv2), W, H, pitch_of(ref), pitch_of(mask), 1e-4, 1e-6, 0.1)
    assert (mask[1:-1, 1:-3, :3] == 0.0).all() and (mask[1:-1, 1:-3, 3] > 0.1).all()


@pytest.mark.parametrize("angle,tx,ty", [(5.0, 1.0, 3.0), (-15.0, -4.0, 2.0), (0.0, 0.0, 0.0), (10.0, 6.0, -6.0)])
def test_prealign_recovers_a_known_rotation(angle, tx, ty):
    """Known answer for the global pre-alignment (oracle/prealign.c): a scene re-sampled under the model
    q = c + R(theta)(p - c - t) comes back with theta to < 0.2 degree and t to the search grid of the finest level."""
    from oracle.bindings import oracle
    from tests.test_parity_kernels import _rotated_pair
    W, H = 512, 256
    ref, mov = _rotated_pair(W, H, angle, tx, ty, 7)
    res = np.zeros(5, np.float32)
    st = np.zeros(4, np.int32)
    n = oracle().preAlign(ref, mov, W, H, ref.strides[0], 20.0, res, st)
    assert n == 4 and st[3] == 0                      # 512 -> 256 -> 128 -> 64: four levels, finest = the image itself
    assert abs(np.degrees(res[2]) - angle) < 0.2
    assert abs(res[0] - tx) <= 1.0 and abs(res[1] - ty) <= 1.0
    assert res[3] == np.cos(np.float32(res[2]), dtype=np.float32) or abs(res[3] - np.cos(res[2])) < 1e-6


def test_sharpenImg_known_answers(orc):
    """sharpenImg (test_opencv/main.cpp:525-534): a constant image is its own blur and comes back unchanged; a step edge is
    sharpened only on its bright side (the uchar subtraction src - blurred saturates at 0, so the dark side always counts
    as low contrast and is copied through) and by exactly src - blurred there."""
    flat = np.full((9, 11, 1), 77, np.uint8)
    out = np.zeros_like(flat)
    orc.call("sharpenImg", flat, out, np.zeros_like(flat), 9, 11, 1, 11, 11)
    assert (out == 77).all()
    edge = np.zeros((8, 20, 1), np.uint8)
    edge[:, :10] = 50
    edge[:, 10:] = 150
    out = np.zeros_like(edge)
    tmp = np.zeros_like(edge)
    orc.call("sharpenImg", edge, out, tmp, 8, 20, 1, 20, 20)
    assert (out[:, :10] == 50).all()                 # dark side: src < blurred -> "low contrast" -> copied
    assert (out[:, 10] > 150).all() and (out[:, 14:] == 150).all()
    # 7 taps exp(-x^2/2) normalised, both passes rounded: column 10 sees three 50s under the left taps
    t = np.exp(-np.arange(-3, 4) ** 2 / 2.0)
    t /= t.sum()
    b = int(np.rint(50 * t[:3].sum() + 150 * t[3:].sum()))
    assert out[0, 10] == min(2 * 150 - b, 255)
