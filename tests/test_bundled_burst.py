"""BASELINE configs[0]: the reference's bundled burst test_opencv/img_00000[0-4].png
(512x256 RGB8, copied as data fixtures to tests/golden/city/).  Frame 1 is a pure
translation of frame 0 by (1, 3) px (measured by phase correlation;
the generator floors U(-5,5) shifts, test_opencv/main.cpp:1896-1907); frames 2-4 are
rotated by 5/10/-15 degrees (main.cpp:1896) and lock only with the global pre-alignment
(cfg.preAlign; csrc/prealign.hip, oracle/prealign.c).
CPU: the oracle pipeline recovers the translation and the three rotations and every frame's robustness
mask says "aligned".  GPU: the HIP pipeline finds the same pre-alignment (integer search: identical) and
matches the oracle on all five frames under the flip-set contract."""
import ctypes
import os

import numpy as np
import pytest

CITY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "city")


def _raws(n):
    from PIL import Image
    out = []
    for i in range(n):
        im = np.asarray(Image.open(os.path.join(CITY, f"img_{i:06d}.png")).convert("RGB"))
        H, W = im.shape[:2]
        yy, xx = np.mgrid[0:H, 0:W]
        c = (yy & 1) + (xx & 1)                      # re-mosaic to RGGB, 8 -> 12 bit
        raw = np.take_along_axis(im, c[..., None], 2)[..., 0].astype(np.uint16) * 16
        out.append(np.ascontiguousarray(raw))
    return out, W, H


def _cfg(W, H, n):
    from multi_frame_super_resolution_amd import capi
    cfg = capi.Config()
    assert capi.lib().raw["mfsr_config_default"](ctypes.byref(cfg), W, H, n, 2, 0) == 0
    for c in range(3):
        cfg.black[c] = 0.0
        cfg.white[c] = 4080.0
    cfg.maxVal = 4080.0
    return cfg


def test_oracle_recovers_the_bundled_translation():
    from oracle.pipeline import OraclePipeline
    raws, W, H = _raws(2)
    assert (W, H) == (512, 256)
    op = OraclePipeline(_cfg(W, H, 2))
    out, q = op.process(raws)
    fl = op.flow
    c = fl[fl.shape[0] // 4:-fl.shape[0] // 4, fl.shape[1] // 4:-fl.shape[1] // 4].reshape(-1, 2)
    # moved(p + u) = ref(p): frame 1 is frame 0 shifted by (+1, +3) -> u = (-1, -3) raw pixels
    np.testing.assert_allclose(np.median(c, 0), [-1.0, -3.0], atol=0.1)
    assert out.shape == (512, 1024, 3) and np.isfinite(out).all()


# rotation of frame k in the model q = c + R(theta)(p - c - base) (the generator's angles {0,0,5,10,-15} with its
# sign convention, test_opencv/main.cpp:1896)
ANGLES = [0.0, 0.0, -5.0, -10.0, 15.0]


def _mask_mean(mask):
    return float(mask[4:-4, 4:-4, :3].mean())


def test_oracle_locks_all_five_bundled_frames():
    """BASELINE configs[0] on the CPU path: with cfg.preAlign every frame of the bundled burst locks (mean robustness
    mask > 0.85; without it the rotated frames reach 0.76 / 0.52 / 0.41) and the pre-alignment recovers the
    generator's rotations to the search grid (0.25 degree at this size)."""
    from oracle.pipeline import OraclePipeline
    raws, W, H = _raws(5)
    cfg = _cfg(W, H, 5)
    cfg.preAlign = 1
    op = OraclePipeline(cfg)
    img_out = np.zeros((2 * H, 2 * W, 3), np.float32)
    tw = np.zeros_like(img_out)
    op.set_reference(raws[0])
    for k in range(5):
        op.add_frame(raws[k], k == 0, img_out, tw)
        if k == 0:
            continue
        assert abs(np.degrees(op.prealign["rotation"]) - ANGLES[k]) <= 0.26, (k, op.prealign)
        assert _mask_mean(op.mask) > 0.85, (k, _mask_mean(op.mask))
    out, _ = op.finish(img_out, tw)
    assert np.isfinite(out).all()
    # without pre-alignment the 15 degree frame does not lock
    cfg.preAlign = 0
    op0 = OraclePipeline(cfg)
    op0.set_reference(raws[0])
    op0.add_frame(raws[4], False, np.zeros_like(img_out), np.zeros_like(img_out))
    assert _mask_mean(op0.mask) < 0.6


@pytest.mark.gpu
def test_hip_matches_oracle_on_the_bundled_burst():
    """All five bundled frames (BASELINE configs[0]) through the HIP pipeline with cfg.preAlign, against the oracle."""
    import torch
    from multi_frame_super_resolution_amd import capi
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    from tests.burst_compare import assert_parity, classify, run_hip, run_oracle
    raws, W, H = _raws(5)
    cfg = _cfg(W, H, 5)
    cfg.preAlign = 1
    frames = [torch.from_numpy(r.view(np.int16)) for r in raws]
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    for k in range(1, 5):
        assert _mask_mean(h["masks"][k]) > 0.85, (k, _mask_mean(h["masks"][k]))
    assert_parity(classify(cfg, h, o), "configs[0] bundled 5-frame burst with pre-alignment")
    # the integer search gives the same pre-alignment on both sides, frame by frame
    from oracle.pipeline import OraclePipeline
    dev = torch.device("cuda:0")
    pipe = BurstPipeline(cfg, dev)
    op = OraclePipeline(cfg)
    pipe.begin_burst()
    pipe.set_reference(frames[0].to(dev))
    op.set_reference(raws[0])
    io, tw = np.zeros((2 * H, 2 * W, 3), np.float32), np.zeros((2 * H, 2 * W, 3), np.float32)
    for k in range(1, 5):
        pipe.add_frame(frames[k].to(dev), False)
        pipe.flush()     # frame-batched alignment: a frame is aligned when its group is complete or on flush
        op.add_frame(raws[k], False, io, tw)
        pa = capi.PreAlign()
        pipe.L.burst_prealign_result(pipe._h, ctypes.byref(pa), None)
        assert (pa.angleIndex, pa.tx, pa.ty, pa.level) == (op.prealign["angle_index"], *op.prealign["t"], op.prealign["level"])
        assert (pa.shiftX, pa.shiftY, pa.rotation) == (*op.prealign["shift"], op.prealign["rotation"])
    pipe.close()
