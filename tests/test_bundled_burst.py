"""BASELINE configs[0]: the reference's bundled burst test_opencv/img_00000[0-4].png
(512x256 RGB8, copied as data fixtures to tests/golden/city/).  Frame 1 is a pure
translation of frame 0 by (1, 3) px (measured by phase correlation;
the generator floors U(-5,5) shifts, test_opencv/main.cpp:1896-1907); frames 2-4 are
rotated by 5/10/-15 degrees and need the global pre-alignment that is listed as "next".
CPU: the oracle pipeline recovers that translation.  GPU: the HIP pipeline matches the oracle."""
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


@pytest.mark.gpu
def test_hip_matches_oracle_on_the_bundled_burst():
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    from oracle.pipeline import OraclePipeline
    raws, W, H = _raws(2)
    cfg = _cfg(W, H, 2)
    ref, _ = OraclePipeline(cfg).process(raws)
    dev = torch.device("cuda:0")
    pipe = BurstPipeline(cfg, dev)
    out, _ = pipe.process([torch.from_numpy(r.view(np.int16)).to(dev) for r in raws])
    got = out.cpu().numpy()
    pipe.close()
    d8 = np.abs(np.round(got * 255) - np.round(ref * 255))
    mse = np.mean((got.astype(np.float64) - ref) ** 2)
    print("bundled burst: PSNR vs oracle", 10 * np.log10(1 / mse), "frac > 1 LSB (8 bit)", np.mean(d8 > 1))
    assert np.mean(d8 > 1) < 2e-3 and 10 * np.log10(1 / mse) > 60
