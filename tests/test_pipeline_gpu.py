"""GPU parity of the whole burst pipeline (C-ABI mfsr_burst_*) against the CPU
oracle pipeline on the same synthetic bursts, plus size-independent properties
at BASELINE.json's full frame sizes.

Tolerance (north_star): outputs within +-1 LSB per channel.  Kernel-level parity
is bit-exact or ~1e-6 (tests/test_parity_kernels.py).  At pipeline level the only
non-bit-exact intermediate is the per-pixel flow (the Lucas-Kanade solve evaluates
atan2/cos/sin/sqrt: 1-2 ulp between ocml and glibc, ~1e-5 px on the flow), and the
flow reaches the image only through roundings and thresholds.  tests/flipset.py
recomputes those decisions for both implementations from their own per-frame flows
and masks; the contract asserted here (tests/burst_compare.py::assert_parity) is

  * outside the flip set: NO 8-bit sample differs by more than 1 LSB;
  * the flip set is a small fraction of the image and is printed with its causes;
  * PSNR vs the oracle >= 70 dB, 8-bit samples off by >1 LSB <= 1e-3 overall.
"""
import ctypes

import numpy as np
import pytest

from tests.burst_compare import assert_parity, classify, flow_difference_report, psnr as _psnr, run_hip, run_oracle

pytestmark = pytest.mark.gpu


def _cfg(W, H, N, scale, mono, fused):
    from multi_frame_super_resolution_amd.pipeline import default_config
    cfg = default_config(W, H, N, scale, mono)
    cfg.fused = fused
    return cfg


def _run_hip(cfg, frames):
    return run_hip(cfg, frames)


def _run_oracle(cfg, frames):
    return run_oracle(cfg, frames)


def _check_outputs(h, o, what, cfg=None):
    """HIP vs oracle with the flip-set classification (cfg given), or HIP vs HIP variants (no cfg: plain statistics)."""
    if cfg is not None:
        assert_parity(classify(cfg, h, o), what)
        return
    d16 = np.abs(h["out16"].astype(np.int64) - o["out16"].astype(np.int64))
    d8 = np.abs(np.round(h["out"] * 255.0) - np.round(o["out"] * 255.0))
    frac16 = float(np.mean(d16 > 1))
    frac8 = float(np.mean(d8 > 1))
    psnr = _psnr(h["out"], o["out"])
    print(f"[{what}] PSNR {psnr:.1f} dB; >1 LSB: 8-bit {frac8:.2e}, 16-bit {frac16:.2e}; max16 {d16.max()}")
    assert psnr >= 70.0
    assert frac8 <= 1e-3
    assert frac16 <= 1e-2


@pytest.mark.parametrize("mono,fused,scale", [(False, 1, 2), (False, 0, 2), (True, 1, 2), (False, 1, 4), (True, 0, 3)])
def test_burst_matches_oracle(mono, fused, scale):
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 256, 192, 4
    frames, shifts, gt = make_burst(W, H, N, scale=scale, mono=mono, seed=1234 + scale, max_shift=4.0)
    cfg = _cfg(W, H, N, scale, mono, fused)
    h = _run_hip(cfg, frames)
    o = _run_oracle(cfg, frames)
    # reference-frame products
    # reference-frame products are bit-identical: tracking image (A1 + luma + Gaussian, only + - * /) and the kernel
    # parameters (E1-E3: the pipeline takes the bit-exact structure-tensor chain, csrc/pipeline.cpp)
    assert np.array_equal(h["tracking"], o["tracking"])
    assert np.array_equal(h["kparam"], o["kparam"], equal_nan=True)
    # last frame's flow (raw-pixel units) and robustness mask
    dflow = np.abs(h["flow"] - o["flow"])
    print(f"flow: max |d| {dflow.max():.2e} px, mean {dflow.mean():.2e}")
    assert np.mean(dflow > 1e-3) < 1e-3
    assert np.mean(np.abs(h["mask"][..., :3] - o["mask"][..., :3]) > 1e-3) < 1e-2
    _check_outputs(h, o, f"mono={mono} fused={fused} s={scale}", cfg)
    # the alignment actually locks on: flow ~ -(shift of the last frame) in raw px (centre region)
    fs = 1 if mono else 2
    true = -shifts[N - 1].numpy()
    c = h["flow"][h["flow"].shape[0] // 4: -h["flow"].shape[0] // 4, h["flow"].shape[1] // 4: -h["flow"].shape[1] // 4]
    np.testing.assert_allclose(np.median(c.reshape(-1, 2), 0), true, atol=0.15)
    assert fs in (1, 2)


@pytest.mark.parametrize("W,H,N,scale,cfa", [(392, 264, 3, 2, "GRBG"), (268, 196, 5, 2, "BGGR"), (328, 200, 3, 4, "RGGB")])
def test_burst_matches_oracle_ragged_sizes(W, H, N, scale, cfa):
    """Sizes that are not whole tiles (partial 256/512-pixel tiles, partial tracker tiles, odd frame counts
    so that one frame is fused alone) and the other Bayer phases, whole pipeline against the oracle."""
    from multi_frame_super_resolution_amd.synth import make_burst
    pat = {"RGGB": [0, 1, 1, 2], "BGGR": [2, 1, 1, 0], "GRBG": [1, 0, 2, 1], "GBRG": [1, 2, 0, 1]}[cfa]
    frames, shifts, gt = make_burst(W, H, N, scale=scale, mono=False, seed=4321 + W, max_shift=3.0)
    cfg = _cfg(W, H, N, scale, False, 1)
    for i in range(4):
        cfg.cfa[i] = pat[i]
    h = _run_hip(cfg, frames)
    o = _run_oracle(cfg, frames)
    np.testing.assert_allclose(h["tracking"], o["tracking"], atol=1e-6)
    dflow = np.abs(h["flow"] - o["flow"])
    assert np.mean(dflow > 1e-3) < 1e-3
    _check_outputs(h, o, f"{W}x{H} N={N} s={scale} {cfa}", cfg)


def test_super_resolution_beats_single_frame():
    """Fusing the burst must recover more of the ground truth than the fallback
    (debayer + bilinear x2) of the reference frame alone."""
    import torch
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 384, 256, 8
    frames, shifts, gt = make_burst(W, H, N, scale=2, mono=False, seed=77, max_shift=3.0)
    cfg = _cfg(W, H, N, 2, False, 1)
    h = _run_hip(cfg, frames)
    cfg1 = _cfg(W, H, 1, 2, False, 1)
    h1 = _run_hip(cfg1, frames[:1])
    g = gt.permute(1, 2, 0).numpy()
    sl = (slice(32, -32), slice(32, -32))
    p_burst = _psnr(h["out"][sl], g[sl])
    p_single = _psnr(h1["out"][sl], g[sl])
    print(f"PSNR vs ground truth: burst {p_burst:.2f} dB, single frame {p_single:.2f} dB")
    assert p_burst > p_single + 0.5


def test_fused_equals_unfused_pipeline():
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 320, 192, 3
    frames, _, _ = make_burst(W, H, N, scale=2, mono=False, seed=5)
    a = _run_hip(_cfg(W, H, N, 2, False, 1), frames)
    b = _run_hip(_cfg(W, H, N, 2, False, 0), frames)
    _check_outputs(a, b, "fused vs unfused")


def test_full_size_properties_4k():
    """BASELINE configs[2] frame size (3840x2160 RGGB, x2): properties that need no oracle.
    (a) accumulating the same frame twice doubles both accumulators (to fp32 rounding) and the
        launch sequence is deterministic bit for bit;
    (b) a burst of identical frames normalises to (statistically) the single-frame result;
    (c) border ring of the HR grid falls back to the debayered reference."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    from multi_frame_super_resolution_amd.synth import make_burst
    dev = torch.device("cuda:0")
    W, H = 3840, 2160
    frames, _, _ = make_burst(W, H, 1, scale=2, mono=False, seed=9, device=dev)
    cfg = _cfg(W, H, 3, 2, False, 1)
    pipe = BurstPipeline(cfg, dev)
    f0 = frames[0]
    pipe.reset_accumulators()
    pipe.set_reference(f0)
    pipe.add_frame(f0, True)
    acc1, w1 = pipe.img_out.clone(), pipe.total_weights.clone()
    out1, _ = pipe.finish()
    out1 = out1.clone()
    pipe.add_frame(f0, True)
    # second pass adds the same 25 terms on top of the first sum: 2x up to fp32 rounding
    assert torch.allclose(pipe.img_out, acc1 * 2, rtol=1e-5, atol=1e-6)
    assert torch.allclose(pipe.total_weights, w1 * 2, rtol=1e-5, atol=1e-6)
    # determinism: the same launch sequence reproduces the accumulators bit for bit
    pipe.reset_accumulators()
    pipe.add_frame(f0, True)
    assert torch.equal(pipe.img_out, acc1) and torch.equal(pipe.total_weights, w1)
    # (b) the same frame added again as *moved* frames (flow ~ 0): where every channel is well
    # supported the normalised image does not change; it may differ where a channel's total
    # weight is tiny or the per-tap certainty varies (ratio of sums), so this is statistical.
    pipe.reset_accumulators()
    pipe.add_frame(f0, True)
    pipe.add_frame(f0, False)
    pipe.add_frame(f0, False)
    out3, _ = pipe.finish()
    d = (out3 - out1).abs()
    print("identical-frame burst vs single frame: mean |d|", float(d.mean()), "frac > 0.02:", float((d > 0.02).float().mean()))
    assert float(d.mean()) < 2e-3 and float((d > 0.02).float().mean()) < 0.03
    # (c) ring = fallback only (accumulators never touch it)
    assert float(pipe.total_weights[0].abs().max()) == 0.0 and float(pipe.total_weights[:, 0].abs().max()) == 0.0
    assert torch.isfinite(out3).all()
    assert 0.05 < float(out3.mean()) < 0.95
    pipe.close()


def test_frame_grouping_equals_frame_by_frame():
    """cfg.pairFrames (1, the default: four frames per pass over the accumulators at x2 Bayer; 2 and 3: that many)
    against one launch per frame: same u16 image up to the re-association of the per-pixel sums; a burst that is
    not a multiple of the group leaves the rest to flush/finish."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 384, 256, 5
    frames, _, _ = synth.make_burst(W, H, N, seed=11, device=dev)
    outs = {}
    for pair in (0, 1, 2, 3):
        cfg = default_config(W, H, N, scale=2)
        assert cfg.pairFrames == 1
        cfg.pairFrames = pair
        pipe = BurstPipeline(cfg, dev)
        assert pipe.group_size() == {0: 1, 1: 4, 2: 2, 3: 3}[pair]
        _, o16 = pipe.process(frames)
        outs[pair] = (o16.cpu().numpy().view(np.uint16).astype(np.int32), pipe.total_weights.cpu().numpy())
        pipe.close()
    for pair in (1, 2, 3):
        d = np.abs(outs[0][0] - outs[pair][0])
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, pair   # 16-bit LSB flips of the re-associated sums
        np.testing.assert_allclose(outs[0][1], outs[pair][1], rtol=2e-5, atol=2e-5)


def test_async_fuse_is_bit_identical():
    """cfg.asyncFuse moves the warp+fuse launches to the burst's own stream (event-ordered after the
    alignment); same kernels in the same order on the accumulators, so the result is bit-identical,
    also when accumulators are read between frames (flush joins the streams)."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 384, 256, 7
    frames, _, _ = synth.make_burst(W, H, N, seed=13, device=dev)
    outs, mids = {}, {}
    for mode in (0, 1):
        cfg = default_config(W, H, N, scale=2)
        cfg.asyncFuse = mode
        pipe = BurstPipeline(cfg, dev)
        for rep in range(2):           # second burst re-uses ring slots and the reference products
            pipe.reset_accumulators()
            pipe.set_reference(frames[0])
            for k in range(N):
                pipe.add_frame(frames[k], k == 0)
                if k == 2:
                    mids[(mode, rep)] = pipe.total_weights.clone()   # flush + join in the middle of the burst
            _, o16 = pipe.finish(want_float=False)
            outs[(mode, rep)] = o16.clone()
        pipe.close()
    for rep in range(2):
        assert torch.equal(outs[(0, rep)], outs[(1, rep)])
        assert torch.equal(mids[(0, rep)], mids[(1, rep)])
    assert torch.equal(outs[(0, 0)], outs[(0, 1)])


@pytest.mark.parametrize("kind", ["zeros", "saturated", "noise", "unrelated"])
def test_degenerate_bursts_finish_with_finite_output(kind):
    """Bursts the alignment cannot lock onto (flat, saturated, pure noise, unrelated frames): every kernel
    must still terminate, the accumulators stay finite and the output is a valid image (the reference's
    fallback image wherever no weight was collected)."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 320, 256, 4
    g = torch.Generator(device=dev).manual_seed(99)
    if kind == "zeros":
        frames = [torch.zeros(H, W, dtype=torch.int16, device=dev) for _ in range(N)]
    elif kind == "saturated":
        frames = [torch.full((H, W), -1, dtype=torch.int16, device=dev) for _ in range(N)]   # 0xFFFF
    elif kind == "noise":
        frames = [torch.randint(0, 4096, (H, W), device=dev, generator=g, dtype=torch.int32).to(torch.int16) for _ in range(N)]
    else:
        base = [torch.randint(0, 4096, (H // 8, W // 8), device=dev, generator=g, dtype=torch.int32) for _ in range(N)]
        frames = [b.repeat_interleave(8, 0).repeat_interleave(8, 1).to(torch.int16).contiguous() for b in base]
    for scale in (2, 4):
        cfg = default_config(W, H, N, scale=scale)
        pipe = BurstPipeline(cfg, dev)
        out, o16 = pipe.process(frames)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert torch.isfinite(pipe.img_out).all() and torch.isfinite(pipe.total_weights).all()
        # the float image is not clamped (values below the black level are negative and the edge-directed
        # debayer of the fallback image overshoots a little, exactly as in the oracle); it stays bounded
        assert float(out.min()) > -1.0 and float(out.max()) < 32.0
        pipe.close()


@pytest.mark.parametrize("scale,mono", [(2, False), (4, False), (2, True)])
def test_begin_burst_equals_zeroed_accumulators(scale, mono):
    """mfsr_burst_begin: the first warp+fuse launch overwrites the accumulators instead of adding to
    zeroed ones -- bit-identical, also when they held garbage, for an odd frame count, and when no frame
    follows (then they read as zero)."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 328, 264, 3
    frames, _, _ = synth.make_burst(W, H, N, scale=scale, mono=mono, seed=17, device=dev)
    cfg = default_config(W, H, N, scale=scale, mono=mono)
    pipe = BurstPipeline(cfg, dev)
    pipe.reset_accumulators()
    pipe.set_reference(frames[0])
    for k in range(N):
        pipe.add_frame(frames[k], k == 0)
    ref_acc, ref_w = pipe.img_out.clone(), pipe.total_weights.clone()
    pipe._img_out.fill_(float("nan"))            # garbage that must never be read
    pipe._total_weights.fill_(1e30)
    pipe.begin_burst()
    pipe.set_reference(frames[0])
    for k in range(N):
        pipe.add_frame(frames[k], k == 0)
    assert torch.equal(pipe.img_out, ref_acc) and torch.equal(pipe.total_weights, ref_w)
    pipe._img_out.fill_(7.0)
    pipe.begin_burst()
    assert float(pipe.img_out.abs().max()) == 0.0 and float(pipe.total_weights.abs().max()) == 0.0
    pipe.close()


@pytest.mark.parametrize("scale,radius,host", [(1, 1, False), (2, 1, True), (2, 2, False)])
def test_frame_stream_matches_oracle_per_window(scale, radius, host):
    """mfsr_stream_* (sliding window over a frame stream, also the x1 denoise-merge scale):
    every output t is compared with the ORACLE burst of its window (frames [t-R, t+R] clipped, reference t), and is
    bit-identical to the HIP burst of the same window -- although the stream uploads and prepares each frame once and
    keeps its products in a ring.  One window also goes through the flip-set classification."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, FrameStream, default_config
    from oracle.pipeline import OraclePipeline
    dev = torch.device("cuda:0")
    W, H, N, R = 256, 192, 6, radius
    frames, _, _ = synth.make_burst(W, H, N, scale=scale, seed=23, max_shift=2.0)
    cfg = default_config(W, H, 2 * R + 1, scale=scale)
    st = FrameStream(cfg, R, dev, host_frames=host)
    outs = {}
    feed = [f.pin_memory() for f in frames] if host else [f.to(dev) for f in frames]
    for f in feed:
        r = st.push(f)
        if r is not None:
            outs[r[0]] = r[1].clone()
    for t, o in st.drain():
        outs[t] = o.clone()
    torch.cuda.synchronize()
    st.close()
    assert sorted(outs) == list(range(N))
    for t in range(N):
        lo, hi = max(0, t - R), min(N - 1, t + R)
        window = frames[lo:hi + 1]
        wcfg = default_config(W, H, len(window), scale=scale)
        wcfg.reference = t - lo
        ref = BurstPipeline(wcfg, dev)
        _, o16 = ref.process([f.to(dev) for f in window])
        assert torch.equal(o16, outs[t]), f"t={t}"
        ref.close()
        _, oq = OraclePipeline(wcfg).process([f.numpy().view(np.uint16) for f in window])
        d = np.abs(outs[t].cpu().numpy().view(np.uint16).astype(np.int64) - oq.astype(np.int64))
        d8 = d / 257.0
        assert np.mean(d8 > 1.0) <= 1e-3, (t, float(np.mean(d8 > 1.0)))
        if t == 2:
            h = run_hip(wcfg, window)
            o = run_oracle(wcfg, window)
            assert_parity(classify(wcfg, h, o), f"stream window t={t} scale {scale}")
    assert not torch.equal(outs[1], outs[2])


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs at their own sizes
# ---------------------------------------------------------------------------------------------------------------------
def _flow_differences_are_localised(cfg, h, o, what, thr=5e-4):
    """The HIP and the oracle flow differ by up to ~1e-3 px at these sizes (1e-5 px on small frames): assert WHERE.  Over
    the well-conditioned, converged interior windows (tests/burst_compare.py::flow_difference_report: ~78 % of the pixels)
    the difference stays <= thr px; every larger one sits in a window whose smaller singular value is in the lowest fifth
    (noise x 1/sigma2, opticalFlow.cu:250-266), on the border rows, or where the flow itself is > 1 px off the frame's
    shift (Lucas-Kanade not converged: wild border / low-texture flows).

    thr scales with the frame: the warp takes its sample position as (x + 0.5 + u) / W * W - 0.5 in fp32
    (opticalFlow.cu:38-39), i.e. quantised to one ulp of the pixel COORDINATE -- 1.2e-4 px for 1024 <= x < 2048, 2.4e-4 px for
    2048 <= x < 4096 -- so the 1e-5 px the two solves differ by (ocml against glibc atan2 / cos / sin) either vanishes or
    becomes a whole coordinate ulp in the next iteration's sample.  Measured over the well-conditioned windows: 2.5e-4 px at
    1080p and 3.0e-4 at 4K (tracking image 1920 wide: 2 - 2.5 ulp), 6.1e-4 at 8K (3840 wide: 2.5 ulp); thr = 4 ulp of the
    tracking image's width: 5e-4 up to 4K, 1e-3 at 8K."""
    for k in range(len(h["flows"])):
        if k == cfg.reference:
            continue
        r = flow_difference_report(h["flows"][k], o["flows"][k], o["tracking"], cfg.lkHalfWindow, thr=thr)
        print(f"[{what}] frame {k}: max |flow diff| {r['max_well']:.2e} px over the well-conditioned {r['well_fraction']:.0%}, "
              f"{r['max_rest']:.2e} over the rest; {r['n_big']} px > {thr:.0e}, {r['big_in_rest_fraction']:.0%} of them in the rest "
              f"(sigma2 20th percentile {r['sigma2_p20']:.2e}, lkMinDet {cfg.lkMinDet:.1e})")
        assert r["max_well"] <= thr
        assert r["n_big"] == 0 or r["big_in_rest_fraction"] == 1.0
        assert r["max_rest"] <= 2e-2


def _flow_locks(h, shifts, k, fs):
    """median flow of frame k in the centre region ~ -(true shift) in raw px"""
    f = h["flows"][k]
    c = f[f.shape[0] // 4: -f.shape[0] // 4, f.shape[1] // 4: -f.shape[1] // 4]
    np.testing.assert_allclose(np.median(c.reshape(-1, 2), 0), -shifts[k].cpu().numpy(), atol=0.15)


def test_config1_as_stated_5_frames_1080p_gray_vs_oracle():
    """BASELINE configs[1] AS STATED: the 5-frame 1920x1080 grayscale burst, x2 (monochrome tile kernel, groups of two frames,
    a last group of one) against the oracle, every frame in the classification.  (The 2-frame and 4-frame SAMPLES of configs[1]
    and configs[2] that rounds 1-3 tested are subsets of this test and of the 16-frame one below: removed in round 4.)"""
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 1920, 1080, 5
    frames, shifts, _ = make_burst(W, H, N, scale=2, mono=True, seed=1234 + 1)
    cfg = _cfg(W, H, N, 2, True, 1)
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    np.testing.assert_allclose(h["tracking"], o["tracking"], atol=1e-6)
    for k in range(1, N):
        _flow_locks(h, shifts, k, 1)
    _flow_differences_are_localised(cfg, h, o, "configs[1] x 5")
    assert_parity(classify(cfg, h, o), "configs[1] 1080p gray x2, the full 5-frame burst")


def test_config2_full_burst_16_frames_4k_vs_oracle():
    """BASELINE configs[2] AS STATED: the whole 16-frame 3840x2160 RGGB burst, x2, default configuration (groups of four frames
    per warp+fuse launch, the rings of eight flow / mask slots cycling twice, launches 2..4 accumulating on top of the first
    one's overwrite) against the oracle -- every frame's flow and mask enter the flip-set classification, the
    full +-1 LSB contract and the flow-difference localisation hold.  (~25 s of oracle on the box's host cores.)"""
    import time
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 3840, 2160, 16
    frames, shifts, _ = make_burst(W, H, N, scale=2, mono=False, seed=1234 + 2)
    cfg = _cfg(W, H, N, 2, False, 1)
    assert cfg.asyncFuse == 0 and cfg.pairFrames == 1      # the defaults bench.py measures
    h = run_hip(cfg, frames)
    t0 = time.time()
    o = run_oracle(cfg, frames)
    print(f"oracle: {time.time() - t0:.1f} s for the 16-frame 4K burst")
    for k in range(1, N):
        _flow_locks(h, shifts, k, 2)
    _flow_differences_are_localised(cfg, h, o, "configs[2] x 16")
    assert_parity(classify(cfg, h, o), "configs[2] 4K RGGB x2, the full 16-frame burst")


def test_config3_x4_at_4k_two_frame_sample_vs_oracle():
    """BASELINE configs[3] frame size AND scale (3840x2160 RGGB -> 15360x8640, x4) on a 2-frame sample (reference + one moved
    frame) against the oracle: the x4 tile kernel at the size its launch geometry, the 1.59 GB accumulators and the 32-bit
    offsets inside them are made for (the other x4 oracle check is 1024x768)."""
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 3840, 2160, 2
    frames, shifts, _ = make_burst(W, H, N, scale=4, mono=False, seed=1234 + 3, max_shift=4.0)
    cfg = _cfg(W, H, N, 4, False, 1)
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    _flow_locks(h, shifts, 1, 2)
    _flow_differences_are_localised(cfg, h, o, "configs[3] at 4K")
    # (x4 from two frames: 21 % of the HR samples are farther than a kernel width from every raw sample of their colour -- total
    # weight below TAU_WEIGHT, under the weight-conditioned bound of continuous_checks; assert_parity caps that share at 25 %)
    assert_parity(classify(cfg, h, o), "configs[3] 4K RGGB x4, 2-frame sample")


def test_config3_x4_crop_vs_oracle():
    """BASELINE configs[3] scale (x4) at 1024x768 (12 Mpix HR grid, every tile path of k_accumulate4xTile, partial
    tiles on both axes) against the oracle."""
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 1024, 768, 3
    frames, shifts, _ = make_burst(W, H, N, scale=4, mono=False, seed=1234 + 3, max_shift=4.0)
    cfg = _cfg(W, H, N, 4, False, 1)
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    for k in range(1, N):
        _flow_locks(h, shifts, k, 2)
    assert_parity(classify(cfg, h, o), "configs[3] x4 at 1024x768")


def test_config3_full_size_properties_4k_x4():
    """BASELINE configs[3] at full size (3840x2160 -> 15360x8640, 2 x 1.59 GB accumulators): properties that need no
    oracle.  (a) determinism bit for bit; (b) frame pairing == frame by frame up to the re-association of two sums;
    (c) the fused image of reference + moved frames stays within the sample range and the HR border ring is the
    fallback; (d) a moved frame's flow locks onto its true shift at this size."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, view_as_tensor
    from multi_frame_super_resolution_amd.synth import make_burst
    dev = torch.device("cuda:0")
    W, H, N = 3840, 2160, 3
    frames, shifts, _ = make_burst(W, H, N, scale=4, mono=False, seed=1234 + 3, device=dev)
    outs = {}
    for pair in (1, 0):
        cfg = _cfg(W, H, N, 4, False, 1)
        cfg.pairFrames = pair
        pipe = BurstPipeline(cfg, dev)
        assert pipe.hr_w == 15360 and pipe.hr_h == 8640
        _, o16 = pipe.process(frames)
        first = o16.clone()
        if pair:
            flow_t = pipe.debug_views()[0]
            fl = view_as_tensor(flow_t, 2, dev)
            c = fl[fl.shape[0] // 4: -fl.shape[0] // 4, fl.shape[1] // 4: -fl.shape[1] // 4].reshape(-1, 2)
            med = c.median(0).values.cpu().numpy()
            np.testing.assert_allclose(med, -shifts[N - 1].cpu().numpy(), atol=0.15)
            _, again = pipe.process(frames)
            assert torch.equal(again, first)                       # (a)
            tw = pipe.total_weights
            assert float(tw[0].abs().max()) == 0.0 and float(tw[:, 0].abs().max()) == 0.0   # (c) ring untouched
            assert float(tw[8:-8, 8:-8].sum(-1).min()) > 0.0     # every interior HR pixel collected weight
            out = pipe.out_img
            assert torch.isfinite(out).all() and 0.05 < float(out.mean()) < 0.95
        outs[pair] = first.view(torch.uint8).view(-1)   # u16 bit patterns as bytes
        pipe.close()
        del pipe
        torch.cuda.empty_cache()
    a = outs[1].view(torch.int16).to(torch.int32) & 0xFFFF
    b = outs[0].view(torch.int16).to(torch.int32) & 0xFFFF
    d = (a - b).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3   # (b)


def test_rotated_burst_needs_and_uses_prealign():
    """SURVEY.md section 8d rotation stress variant: frames rotated by up to 10 degrees.  With cfg.preAlign the rotated
    frames lock (robustness mask) and the fused image is closer to the ground truth than the burst without them; the
    HIP pipeline matches the oracle under the flip-set contract."""
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 384, 256, 5
    angles = [0.0, 0.0, 4.0, -7.0, 10.0]
    frames, shifts, gt = make_burst(W, H, N, scale=2, mono=False, seed=31, max_shift=3.0, angles_deg=angles)
    cfg = _cfg(W, H, N, 2, False, 1)
    cfg.preAlign = 1
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    assert_parity(classify(cfg, h, o), "rotated burst with pre-alignment")
    locked = [float(h["masks"][k][8:-8, 8:-8, :3].mean()) for k in range(N)]
    cfg0 = _cfg(W, H, N, 2, False, 1)
    h0 = run_hip(cfg0, frames)
    unlocked = [float(h0["masks"][k][8:-8, 8:-8, :3].mean()) for k in range(N)]
    print("mask means with pre-alignment", locked, "without", unlocked)
    assert min(locked[2:]) > 0.8 and locked[4] > unlocked[4] + 0.2
    g = gt.permute(1, 2, 0).numpy()
    sl = (slice(64, -64), slice(64, -64))
    cfg2 = _cfg(W, H, 2, 2, False, 1)
    h2 = run_hip(cfg2, frames[:2])
    p5, p2, p5_no = _psnr(h["out"][sl], g[sl]), _psnr(h2["out"][sl], g[sl]), _psnr(h0["out"][sl], g[sl])
    print(f"PSNR vs ground truth: 5 frames + pre-alignment {p5:.2f} dB, 5 frames without {p5_no:.2f} dB, 2 frames {p2:.2f} dB")
    assert p5 > p2 and p5 > p5_no


@pytest.mark.parametrize("ring,pair,async_fuse", [(3, 1, 0), (4, 0, 0), (4, 1, 1)])
def test_host_frame_burst_equals_device_frame_burst(ring, pair, async_fuse):
    """mfsr_burst_*_host (frames in pinned host memory, uploaded by the library's copy stream into a ring of device
    slots, result copied back to host memory) is bit-identical to the device-resident burst, over several bursts on
    one context (slot re-use, reference double buffer) and with a reference that is not the first frame."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 384, 256, 7
    frames, _, _ = synth.make_burst(W, H, N, seed=29, device="cpu")
    for ref in (0, 3):
        cfg = default_config(W, H, N, scale=2)
        cfg.reference = ref
        cfg.pairFrames = pair
        cfg.asyncFuse = async_fuse
        plain = BurstPipeline(cfg, dev)
        _, want = plain.process([f.to(dev) for f in frames])
        want = want.cpu()
        plain.close()
        cfg.uploadRing = ring
        pipe = BurstPipeline(cfg, dev)
        pinned = [f.pin_memory() for f in frames]
        for rep in range(3):
            got = pipe.process_host(pinned)
            pipe.host_sync()
            assert torch.equal(got, want), (ref, rep)
            got.zero_()
        pipe.close()


@pytest.mark.parametrize("ring,announce", [(16, "reversed"), (16, "partial"), (4, "all"), (16, "with_reference")])
def test_host_burst_prefetch_announcements_that_do_not_match(ring, announce):
    """mfsr_burst_prefetch_host queues the uploads of the announced frames up front; add_frame_host matches them by host
    pointer IN ORDER.  Frames that come in another order than announced, only partly announced, announced with a ring
    shorter than the burst, or announced together with the reference's own pointer must still give the resident burst's
    image (an unmatched announcement is a wasted copy, never a wrong frame)."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 384, 256, 9
    frames, _, _ = synth.make_burst(W, H, N, seed=61, device="cpu")
    cfg = default_config(W, H, N, scale=2)
    cfg.reference = 1
    plain = BurstPipeline(cfg, dev)
    _, want = plain.process([f.to(dev) for f in frames])
    want = want.cpu().clone()
    plain.close()
    cfg.uploadRing = ring
    pipe = BurstPipeline(cfg, dev)
    pinned = [f.pin_memory() for f in frames]
    out_host = torch.zeros(H * 2, W * 2, 3, dtype=torch.int16).pin_memory()
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        pipe.begin_burst()
        pipe.L.burst_set_reference_host(pipe._h, pinned[cfg.reference].data_ptr(), st)
        ann = {"reversed": pinned[::-1], "partial": pinned[:4], "all": pinned, "with_reference": pinned}[announce]
        ptrs = (ctypes.c_void_p * len(ann))(*[f.data_ptr() for f in ann])
        pipe.L.burst_prefetch_host(pipe._h, ptrs, len(ann), st)
        for k, f in enumerate(pinned):
            pipe.L.burst_add_frame_host(pipe._h, f.data_ptr(), 1 if k == cfg.reference else 0, pipe._img_out.data_ptr(),
                                        pipe._total_weights.data_ptr(), st)
        pipe.L.burst_finish_host(pipe._h, pipe._img_out.data_ptr(), pipe._total_weights.data_ptr(), pipe.out16.data_ptr(),
                                 out_host.data_ptr(), st)
        pipe.host_sync()
        assert torch.equal(out_host, want), (ring, announce, rep)
        out_host.zero_()
    pipe.close()


@pytest.mark.parametrize("ring,group", [(32, 4), (9, 4), (5, 2), (16, 3)])
def test_host_bursts_back_to_back_equal_single_bursts(ring, group):
    """Host bursts enqueued back to back with NO host synchronisation between them (the next burst's uploads run under
    this burst's banded tail and download; a burst that finds the previous one still in flight batches its alignment per
    group, mfsr_burst_set_reference_host): every image equals the device-resident burst.  Two different bursts alternate,
    so a stale upload slot, reference slot or kernel-parameter image of the other burst would show; rings both deeper and
    shallower than a burst."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 512, 384, 11
    bursts = [synth.make_burst(W, H, N, seed=31 + i, device="cpu")[0] for i in range(2)]
    cfg = default_config(W, H, N, scale=2)
    cfg.reference = 2
    cfg.pairFrames = group
    plain = BurstPipeline(cfg, dev)
    want = []
    for fr in bursts:
        _, w = plain.process([f.to(dev) for f in fr])
        want.append(w.cpu().clone())
    plain.close()
    assert not torch.equal(want[0], want[1])
    cfg.uploadRing = ring
    pipe = BurstPipeline(cfg, dev)
    pinned = [[f.pin_memory() for f in fr] for fr in bursts]
    outs = [torch.zeros(H * 2, W * 2, 3, dtype=torch.int16).pin_memory() for _ in range(6)]
    for i in range(6):
        pipe.process_host(pinned[i & 1], outs[i])     # no host_sync in between
    pipe.host_sync()
    torch.cuda.synchronize()
    for i in range(6):
        assert torch.equal(outs[i], want[i & 1]), (ring, group, i)
    pipe.close()


def test_config4_8k_two_frame_sample_vs_oracle():
    """BASELINE configs[4] frame size (7680x4320 RGGB -> 15360x8640, x2) on a 2-frame sample (reference + one moved frame)
    against the oracle: every kernel of the burst at the 8K launch geometry (4x the tiles, bands and strips of 4K, 66 MB raw
    frames, offsets past 2^31 bytes inside the accumulators) -- the 64-frame test below only compares the build with itself."""
    from multi_frame_super_resolution_amd.synth import make_burst
    W, H, N = 7680, 4320, 2
    frames, shifts, _ = make_burst(W, H, N, scale=2, mono=False, seed=1234 + 4, max_shift=4.0)
    cfg = _cfg(W, H, N, 2, False, 1)
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    _flow_locks(h, shifts, 1, 2)
    _flow_differences_are_localised(cfg, h, o, "configs[4] at 8K", thr=1e-3)     # (tracking image 3840 wide: see the helper)
    assert_parity(classify(cfg, h, o), "configs[4] 8K RGGB x2, 2-frame sample")


def test_config4_64_frame_8k_host_burst_equals_resident_burst():
    """BASELINE configs[4] AS STATED on one GPU: a 64-frame 7680x4320 RGGB burst (4.2 GB of raw frames in pinned host memory)
    through the library's upload ring (32 slots of 66 MB: every slot is refilled, the flow / mask rings of eight slots cycle
    eight times, the last two groups are held for the banded finish) equals the same burst resident in HBM, bit for bit.
    (16 distinct frames of one scene, visited in an order without a short period, stand for the 64.)"""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 7680, 4320, 64
    distinct, _, _ = synth.make_burst(W, H, 16, scale=2, seed=1234 + 4, device=dev)
    order = [0] + [1 + ((7 * k + k // 15) % 15) for k in range(N - 1)]      # frame 0 (the reference) once, the others mixed
    cfg = default_config(W, H, N, scale=2)
    plain = BurstPipeline(cfg, dev)
    _, want = plain.process([distinct[i] for i in order])
    want = want.cpu().clone()
    plain.close()
    del plain
    torch.cuda.empty_cache()
    cfg.uploadRing = 32
    pipe = BurstPipeline(cfg, dev)
    host = [f.cpu().pin_memory() for f in distinct]
    pinned = [host[i] for i in order]
    for rep in range(2):
        got = pipe.process_host(pinned)
        pipe.host_sync()
        assert torch.equal(got, want), rep
        got.zero_()
    pipe.close()


@pytest.mark.parametrize("group", [4, 2, 3])
def test_host_burst_with_more_frames_than_configured_and_resident_burst_after_it(group):
    """A group that mfsr_burst_add_frame_host holds back for finish_host (the one cfg.frames' last frame completes) must
    not stay held when frames keep arriving: cfg.frames + k host frames per reference, and a device-resident burst on a
    context that ran a host burst before, both equal the plain resident burst of the same frames (the held group used to
    be overrun: pend.n == group, then a write past the group's arrays)."""
    import torch
    from multi_frame_super_resolution_amd import synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H = 384, 256
    n_cfg = 2 * group                       # the hold triggers when the frame count reaches cfg.frames on a full group
    for extra in (1, 2, group + 1):
        N = n_cfg + extra
        frames, _, _ = synth.make_burst(W, H, N, seed=57 + extra, device="cpu")
        cfg = default_config(W, H, N, scale=2)
        cfg.pairFrames = group
        plain = BurstPipeline(cfg, dev)
        _, want = plain.process([f.to(dev) for f in frames])
        want = want.cpu().clone()
        plain.close()
        cfg = default_config(W, H, n_cfg, scale=2)   # the context believes in a shorter burst than it is fed
        cfg.pairFrames = group
        cfg.uploadRing = 16
        pipe = BurstPipeline(cfg, dev)
        pinned = [f.pin_memory() for f in frames]
        got = pipe.process_host(pinned)
        pipe.host_sync()
        assert torch.equal(got, want), (group, extra, "host burst longer than cfg.frames")
        # a resident burst on the same context afterwards (holdLastGroup must not leak into it)
        _, got2 = pipe.process([f.to(dev) for f in frames])
        torch.cuda.synchronize()
        assert torch.equal(got2.cpu(), want), (group, extra, "resident burst after a host burst")
        # ... and a host burst again
        got3 = pipe.process_host(pinned)
        pipe.host_sync()
        assert torch.equal(got3, want), (group, extra, "host burst after a resident one")
        pipe.close()


@pytest.mark.parametrize("ref_index", [0, 2])
def test_joint_minimiser_burst_matches_oracle(ref_index):
    """Stage C in the pipeline (mfsr_burst_process_joint): neighbouring pairs measured besides the (reference, k) pairs,
    per-tile least squares with outlier rejection inside one launch (mfsr_minimizeShiftsFused), getOptimalShifts feeding
    the flow field -- against the oracle's process_joint (same pairs, host-driven solve / checkForOutliers loop)."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, view_as_tensor
    from multi_frame_super_resolution_amd.synth import make_burst
    from oracle.pipeline import OraclePipeline
    dev = torch.device("cuda:0")
    W, H, N = 320, 256, 5
    frames, shifts, gt = make_burst(W, H, N, scale=2, mono=False, seed=1234 + 7, max_shift=3.0)
    cfg = _cfg(W, H, N, 2, False, 1)
    cfg.reference = ref_index
    nf = [f.numpy().view(np.uint16) for f in frames]
    op = OraclePipeline(cfg)
    o_out, o_q = op.process_joint(nf)
    pipe = BurstPipeline(cfg, dev)
    h_out, h_q = pipe.process_joint([f.to(dev) for f in frames])
    torch.cuda.synchronize()
    flow_t, mask_t, _, _ = pipe.debug_views()
    h = dict(out=h_out.cpu().numpy(), out16=h_q.cpu().numpy().view(np.uint16), tw=pipe.total_weights.cpu().numpy(),
             flows=[None] * (N - 1) + [view_as_tensor(flow_t, 2, dev).cpu().numpy()],
             masks=[None] * (N - 1) + [view_as_tensor(mask_t, 4, dev).cpu().numpy()])
    # the tile shifts out of the minimiser are bit-identical (tracker and solve are); the last frame's flow agrees to LK rounding
    last = N - 1 if ref_index != N - 1 else N - 2
    assert last == N - 1
    dflow = np.abs(h["flows"][-1] - op.flows[-1])
    assert np.mean(dflow > 1e-3) < 1e-3
    d8 = np.abs(np.round(np.clip(h["out"], 0, 1) * 255) - np.round(np.clip(o_out, 0, 1) * 255))
    psnr = _psnr(h["out"], o_out)
    print(f"joint burst ref={ref_index}: PSNR vs oracle {psnr:.1f} dB, 8-bit >1 LSB {float((d8 > 1).mean()):.2e}, "
          f"oracle dropped {op.joint['dropped']} measurements")
    assert psnr >= 70.0 and float((d8 > 1).mean()) <= 1e-3
    # and the joint result is a sensible super-resolution: close to the plain burst's
    plain = BurstPipeline(cfg, dev)
    p_out, _ = plain.process([f.to(dev) for f in frames])
    assert _psnr(p_out.cpu().numpy(), h["out"]) > 35.0
    pipe.close()
    plain.close()


def test_joint_minimiser_rejects_an_inconsistent_measurement_like_the_oracle():
    """Stage C's reject path end to end (checkForOutliers, ShiftMinimizerKernels.cu:81-139): a patch of ONE frame is replaced
    by unrelated texture (an occlusion), so every pair that involves that frame measures an arbitrary shift in the tiles
    under it -- inconsistent with the neighbouring pairs' sums, residual > 1 px^2 -- and the solve / reject loop drops those
    rows.  The oracle's host-driven loop reports how many rows it dropped (> 0 here); the HIP burst (whole loop inside one
    launch) must give the same tile shifts, or every frame's flow under the patch would differ: all frames go through the
    flip-set classification with the continuous checks."""
    import torch
    from multi_frame_super_resolution_amd import capi
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, view_as_tensor
    from multi_frame_super_resolution_amd.synth import make_burst
    from oracle.pipeline import OraclePipeline
    dev = torch.device("cuda:0")
    W, H, N = 320, 256, 5
    frames, shifts, gt = make_burst(W, H, N, scale=2, mono=False, seed=1234 + 9, max_shift=3.0)
    rng = np.random.default_rng(99)
    bad = frames[3].clone()
    patch = torch.from_numpy(rng.integers(300, 3800, size=(96, 128), dtype=np.int64).astype(np.int16))
    bad[80:176, 96:224] = patch
    frames = list(frames)
    frames[3] = bad
    cfg = _cfg(W, H, N, 2, False, 1)
    cfg.reference = 0
    nf = [f.numpy().view(np.uint16) for f in frames]
    op = OraclePipeline(cfg)
    o_out, o_q = op.process_joint(nf)
    print(f"oracle dropped {op.joint['dropped']} measurements (pairs {op.joint['pairs']})")
    assert op.joint["dropped"] > 0, "the injected occlusion did not make any measurement inconsistent"
    pipe = BurstPipeline(cfg, dev)
    h_out, h_q = pipe.process_joint([f.to(dev) for f in frames])
    torch.cuda.synchronize()
    flows, masks = [], []
    for k in range(N):
        ft, mt = capi.Tex2D(), capi.Tex2D()
        pipe.L.burst_debug_frame_views(pipe._h, N - 1 - k, ctypes.byref(ft), ctypes.byref(mt))
        flows.append(view_as_tensor(ft, 2, dev).cpu().numpy())
        masks.append(view_as_tensor(mt, 4, dev).cpu().numpy())
    h = dict(out=h_out.cpu().numpy(), out16=h_q.cpu().numpy().view(np.uint16), img_out=pipe.img_out.cpu().numpy(),
             tw=pipe.total_weights.cpu().numpy(), flows=flows, masks=masks)
    o = dict(out=o_out, out16=o_q, img_out=op.img_out, tw=op.tw, flows=op.flows, masks=op.masks)
    for k in range(1, N):
        d = np.abs(flows[k] - op.flows[k])
        print(f"frame {k}: max |flow diff| {d.max():.2e} px, fraction > 1e-3 px: {np.mean(d > 1e-3):.2e}")
        assert np.mean(d > 1e-2) < 1e-3      # same tile shifts on both sides (a kept / dropped row moves a tile by >= 1 px)
    assert_parity(classify(cfg, h, o), "joint burst with an occluded patch in frame 3", max_flip_fraction=0.1)
    pipe.close()


def test_frame_source_callback_equals_push_api():
    """mfsr_burst_process_source: the pull model of the reference's FrameSource subclass (multi_frame_sr.cpp:18-49: nextFrame
    copies the next device frame into the caller's buffer, leaves it empty when exhausted; reset rewinds).  Same result as
    pushing the same frames with add_frame; a source that runs dry early gives the burst of the frames it delivered."""
    import ctypes
    import torch
    from multi_frame_super_resolution_amd import capi, synth
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config
    dev = torch.device("cuda:0")
    W, H, N = 320, 256, 6
    frames = [f.to(dev) for f in synth.make_burst(W, H, N, seed=37)[0]]
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    NEXT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    RESET = ctypes.CFUNCTYPE(None, ctypes.c_void_p)

    class Source(ctypes.Structure):
        _fields_ = [("next_frame", NEXT), ("reset", RESET), ("user", ctypes.c_void_p)]

    for available in (N, 4):
        state = {"i": 0, "resets": 0}

        def next_frame(user, dst, stream):
            if state["i"] >= available:
                return 0
            f = frames[state["i"]]
            state["i"] += 1
            return 1 if hip.hipMemcpyAsync(dst, f.data_ptr(), W * H * 2, 3, stream) == 0 else -1

        def reset(user):
            state["i"] = 0
            state["resets"] += 1

        src = Source(NEXT(next_frame), RESET(reset), None)
        cfg = default_config(W, H, N, scale=2)
        cfg.uploadRing = 3
        pipe = BurstPipeline(cfg, dev)
        used = ctypes.c_int(0)
        for rep in range(2):
            pipe.L.burst_process_source(pipe._h, ctypes.byref(src), pipe._img_out.data_ptr(), pipe._total_weights.data_ptr(), None,
                                        pipe.out16.data_ptr(), ctypes.byref(used), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert used.value == available and state["resets"] == rep + 1
        got = pipe.out16.clone()
        pipe.close()
        wcfg = default_config(W, H, available, scale=2)
        plain = BurstPipeline(wcfg, dev)
        _, want = plain.process(frames[:available])
        assert torch.equal(got, want), available
        plain.close()


def test_exact_reciprocal_division_equals_the_division():
    """The flow-field, Lucas-Kanade and robustness kernels divide by the image dimensions with the reciprocal sequence the host
    has proven exact for that divisor (common.hpp::mfsr_div, include/mfsr.h::mfsr_exactDivisionOk); MFSR_EXACT_DIV=0 keeps the
    IEEE division.  The switch is read once per process, so each setting runs in its own: same flows, masks and image, bit for
    bit, on a ragged RGGB burst and on a monochrome one (flow at another resolution than the mask)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from multi_frame_super_resolution_amd.pipeline import default_config
from multi_frame_super_resolution_amd.synth import make_burst
from tests.burst_compare import run_hip
h = hashlib.sha256()
for (W, H, N, s, mono) in ((392, 264, 5, 2, False), (320, 200, 3, 2, True)):
    frames, _, _ = make_burst(W, H, N, scale=s, mono=mono, seed=99, max_shift=4.0)
    r = run_hip(default_config(W, H, N, s, mono), frames)
    for a in [r["out16"], r["img_out"], r["tw"]] + r["flows"] + r["masks"]:
        h.update(np.ascontiguousarray(a).tobytes())
print("SHA", h.hexdigest())
''' % root
    outs = []
    for v in ("1", "0"):
        env = dict(os.environ, MFSR_EXACT_DIV=v)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([l for l in p.stdout.splitlines() if l.startswith("SHA")][0])
    assert outs[0] == outs[1]
