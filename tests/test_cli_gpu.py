"""The reference-style CLI (apps/multi_frame_sr.cpp, drop-in for
finalProject/Project/multi_frame_sr.cpp:122-210): same argv, same prints, same output files."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "apps", "multi_frame_sr")


def test_cli_usage_matches_reference():
    if not os.path.exists(CLI):
        pytest.skip("apps/multi_frame_sr not built")
    p = subprocess.run([CLI, "only", "two"], capture_output=True, text=True)
    assert p.returncode != 0
    assert "./multi_frame_sr optFlowName inputName iterations" in p.stdout       # multi_frame_sr.cpp:138
    p = subprocess.run([CLI, "farneback", "nowhere", "3"], capture_output=True, text=True)
    assert "wrong input" in p.stdout                                             # :161
    p = subprocess.run([CLI, "simple", "city", "3"], capture_output=True, text=True)
    assert "Incorrect Optical Flow algorithm - simple" in p.stderr               # :84


@pytest.mark.gpu
def test_cli_city_burst(tmp_path):
    from PIL import Image
    import torch
    from multi_frame_super_resolution_amd.synth import _scene, _shifted
    assert os.path.exists(CLI), "build apps/multi_frame_sr first (__graft_entry__.build())"
    # 5 frames 512x256 RGB8 like the bundled test_opencv/img_00000[0-4].png, pure translations
    gen = torch.Generator().manual_seed(3)
    scene = _scene(256 * 2 + 64, 512 * 2 + 64, gen, "cpu")
    shifts = [(0, 0), (1.3, -2.1), (-3.2, 0.6), (2.4, 2.9), (-1.1, -1.7)]
    for i, (dx, dy) in enumerate(shifts):
        sh = _shifted(scene, dx * 2, dy * 2)[:, 32:32 + 512, 32:32 + 1024]
        lr = torch.nn.functional.avg_pool2d(sh[None], 2)[0]
        img = (lr.permute(1, 2, 0).clamp(0, 1) * 255).round().byte().numpy()
        Image.fromarray(img).save(tmp_path / f"img_{i + 1:06d}.png")          # reference indexes 1..5 (:171)
    p = subprocess.run([CLI, "farneback", "city", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert " sec" in p.stdout and " FPS" in p.stdout                             # :205-206
    out = np.asarray(Image.open(tmp_path / "city_farneback_sr_result.png"))      # :207
    out2 = np.asarray(Image.open(tmp_path / "city_farneback_sr2_result.png"))    # :209
    assert out.shape == (512, 1024, 3) and out2.shape == out.shape
    gt = (scene[:, 32:32 + 512, 32:32 + 1024].permute(1, 2, 0).clamp(0, 1) * 255).numpy()
    mse = np.mean((out[32:-32, 32:-32].astype(np.float64) - gt[32:-32, 32:-32]) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    print("CLI x2 result PSNR vs scene:", psnr)
    assert psnr > 24.0
    assert (out2[0] == 0).all() and (out2[:, 0] == 0).all()                      # sharpenImg2 zeroes the ring (:114-117)


@pytest.mark.gpu
def test_cli_on_the_bundled_city_frames(tmp_path):
    """The CLI on the reference's own data (tests/golden/city = test_opencv/img_00000[0-4].png, 0-based names): its 8-bit
    result is exactly what the Python host gets from the same library with the same configuration (whose parity with
    the oracle tests/test_bundled_burst.py asserts), so all five frames -- including the rotated ones -- are fused."""
    import ctypes
    import shutil
    import torch
    from PIL import Image
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    from tests.test_bundled_burst import CITY, _cfg, _raws
    assert os.path.exists(CLI), "build apps/multi_frame_sr first (__graft_entry__.build())"
    for i in range(5):
        shutil.copy(os.path.join(CITY, f"img_{i:06d}.png"), tmp_path / f"img_{i:06d}.png")
    p = subprocess.run([CLI, "farneback", "city", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "img_000000.png, [512 x 256]" in p.stdout and "img_000004.png, [512 x 256]" in p.stdout
    out = np.asarray(Image.open(tmp_path / "city_farneback_sr_result.png"))
    raws, W, H = _raws(5)
    cfg = _cfg(W, H, 5)
    cfg.preAlign = 1
    cfg.lkIterations = 3
    dev = torch.device("cuda:0")
    pipe = BurstPipeline(cfg, dev)
    res, _ = pipe.process([torch.from_numpy(r.view(np.int16)).to(dev) for r in raws])
    q = torch.empty(2 * H, 2 * W, 3, dtype=torch.uint8, device=dev)
    pipe.L.quantize(res.data_ptr(), 12 * 2 * W, None, q.data_ptr(), 2 * W, 2 * H, 255.0, None)
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), out)
    pipe.close()


@pytest.mark.gpu
def test_cli_shards_the_burst_from_one_process(tmp_path):
    """MFSR_GPUS=3 (argv unchanged): the CLI drives mfsr_dist_group from its one process -- here with the three ranks on
    device 0 (MFSR_VIRTUAL_RANKS=1).  The bundled city frames are rotated by up to 15 degrees: vertical flows beyond the
    default 64-row halo, so the run also takes the status-1 -> whole-raw-frames path.  Same picture as the 1-GPU CLI (the
    u16 images are bit-identical, tests/test_dist_local_gpu.py; the two 8-bit quantisations may differ by one level where
    the u16 value sits on a rounding boundary)."""
    import shutil
    from PIL import Image
    from tests.test_bundled_burst import CITY
    assert os.path.exists(CLI), "build apps/multi_frame_sr first (__graft_entry__.build())"
    outs = []
    for sub, env in (("one", {}), ("three", {"MFSR_GPUS": "3", "MFSR_VIRTUAL_RANKS": "1"})):
        d = tmp_path / sub
        d.mkdir()
        for i in range(5):
            shutil.copy(os.path.join(CITY, f"img_{i:06d}.png"), d / f"img_{i:06d}.png")
        p = subprocess.run([CLI, "farneback", "city", "3"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        assert " sec" in p.stdout and " FPS" in p.stdout
        outs.append((np.asarray(Image.open(d / "city_farneback_sr_result.png")).astype(int),
                     np.asarray(Image.open(d / "city_farneback_sr2_result.png")).astype(int)))
    d0 = np.abs(outs[0][0] - outs[1][0])
    print("1-GPU vs 3-rank CLI: max 8-bit difference", d0.max(), "fraction differing", float(np.mean(d0 > 0)))
    assert d0.max() <= 1 and np.mean(d0 > 0) < 0.01


def test_image_readers_against_an_independent_decoder(tmp_path):
    """apps/image_io.hpp (PNG, PNM, baseline JPEG readers of the CLI) through apps/imgconv, against PIL: the bundled city
    PNGs decode exactly, the reference's "car" JPEGs (finalProject/Project/car/[1-4].jpg, copied as data fixtures: baseline
    SOF0, YCbCr 4:2:0) to within one level of libjpeg's output."""
    from PIL import Image
    conv = os.path.join(ROOT, "apps", "imgconv")
    if not os.path.exists(conv):
        pytest.skip("apps/imgconv not built")
    gold = os.path.join(ROOT, "tests", "golden")
    for rel, exact in [("city/img_000002.png", True)] + [(f"car/{n}.jpg", False) for n in (1, 2, 3, 4)]:
        out = tmp_path / "o.ppm"
        subprocess.check_call([conv, os.path.join(gold, rel), str(out)])
        a = np.asarray(Image.open(out)).astype(int)
        b = np.asarray(Image.open(os.path.join(gold, rel)).convert("RGB")).astype(int)
        assert a.shape == b.shape
        d = np.abs(a - b)
        if exact:
            assert d.max() == 0
        else:
            assert d.max() <= 2 and d.mean() < 0.05, (rel, d.max(), d.mean())
    # not an image / truncated file: rejected, no crash
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(open(os.path.join(gold, "car/1.jpg"), "rb").read()[:300])
    assert subprocess.run([conv, str(bad), str(tmp_path / "x.ppm")], capture_output=True).returncode != 0


def test_image_readers_refuse_hostile_headers(tmp_path):
    """Untrusted files: headers that promise far more pixels than the file can hold are refused before anything of that
    size is allocated (PNG 65536^2 RGBA = 17 GB, JPEG SOF0 65500^2), a PNM header that ends in white space or overflows
    its integers is refused without reading past the buffer -- exit code 1 from imgconv, never a crash (which would be a
    negative return code) and never a multi-gigabyte allocation (checked through the address-space limit)."""
    import struct
    import zlib
    conv = os.path.join(ROOT, "apps", "imgconv")
    if not os.path.exists(conv):
        pytest.skip("apps/imgconv not built")

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    cases = {
        "huge.png": b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 65536, 65536, 8, 6, 0, 0, 0))
                    + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""),
        "ws.pgm": b"P5 12 12   \n  ",
        "overflow.pgm": b"P5 99999999999999999999 4 255\n" + b"\0" * 64,
        "zero.pgm": b"P5 0 4 255\n" + b"\0" * 64,
        "neg.ppm": b"P6 -4 4 255\n" + b"\0" * 64,
    }
    jpg = bytearray(open(os.path.join(ROOT, "tests", "golden", "car/1.jpg"), "rb").read())
    i = jpg.find(b"\xff\xc0")
    assert i > 0
    jpg[i + 5:i + 9] = struct.pack(">HH", 65500, 65500)      # SOF0: height, width
    cases["huge.jpg"] = bytes(jpg)
    import resource

    def limit():
        resource.setrlimit(resource.RLIMIT_AS, (2 << 30, 2 << 30))   # a 17 GB vector would throw; the readers must not even try

    for name, data in cases.items():
        f = tmp_path / name
        f.write_bytes(data)
        p = subprocess.run([conv, str(f), str(tmp_path / "o.ppm")], capture_output=True, preexec_fn=limit)
        assert p.returncode == 1, (name, p.returncode, p.stderr[-200:])


def _tiff_bytes(arr, big_endian, rows_per_strip):
    """Hand-written baseline TIFF (uncompressed, chunky) of a uint8 / uint16 array [h, w] or [h, w, c], several strips."""
    import struct
    e = ">" if big_endian else "<"
    h, w = arr.shape[:2]
    c = 1 if arr.ndim == 2 else arr.shape[2]
    bits = arr.dtype.itemsize * 8
    data = arr.astype(arr.dtype.newbyteorder(e)).tobytes()
    row = w * c * bits // 8
    strips = [(y, min(rows_per_strip, h - y)) for y in range(0, h, rows_per_strip)]
    n = len(strips)
    entries = []
    extra = b""
    ifd_off = 8 + len(data)
    n_entries = 9
    extra_off = ifd_off + 2 + n_entries * 12 + 4

    def entry(tag, typ, vals):
        nonlocal extra
        sz = {3: 2, 4: 4}[typ]
        raw = b"".join(struct.pack(e + {3: "H", 4: "I"}[typ], v) for v in vals)
        if len(raw) <= 4:
            val = raw.ljust(4, b"\0")
        else:
            val = struct.pack(e + "I", extra_off + len(extra))
            extra += raw
        entries.append(struct.pack(e + "HHI", tag, typ, len(vals)) + val)

    entry(256, 4, [w])
    entry(257, 4, [h])
    entry(258, 3, [bits] * c)
    entry(259, 3, [1])
    entry(262, 3, [1 if c == 1 else 2])
    entry(273, 4, [8 + y * row for y, _ in strips])
    entry(277, 3, [c])
    entry(278, 4, [rows_per_strip])
    entry(279, 4, [r * row for _, r in strips])
    assert len(entries) == n_entries
    head = (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, ifd_off)
    return head + data + struct.pack(e + "H", n_entries) + b"".join(entries) + struct.pack(e + "I", 0) + extra


def test_tiff_reader_against_numpy_and_pil(tmp_path):
    """The CLI's TIFF reader (apps/image_io.hpp: classic TIFF, II / MM, uncompressed, 8 / 16 bit, 1 / 3 samples, any strip
    layout) through apps/imgconv: hand-written files decode to the array they were made from, files written by PIL
    (its own strip layout) likewise; compressed files are refused."""
    from PIL import Image
    conv = os.path.join(ROOT, "apps", "imgconv")
    if not os.path.exists(conv):
        pytest.skip("apps/imgconv not built")
    r = np.random.default_rng(5)

    def decode(path):
        out = tmp_path / "o.pnm"
        subprocess.check_call([conv, str(path), str(out)])
        raw = open(out, "rb").read()
        magic, dims, maxv, body = raw.split(b"\n", 3)
        w, h = (int(v) for v in dims.split())
        c = 1 if magic == b"P5" else 3
        a = np.frombuffer(body, dtype=">u2" if int(maxv) == 65535 else np.uint8).reshape(h, w, c)
        return a.astype(np.int64)[..., 0] if c == 1 else a.astype(np.int64)

    cases = [(r.integers(0, 65536, (37, 53), dtype=np.uint16), True, 5), (r.integers(0, 65536, (40, 31, 3), dtype=np.uint16), True, 7),
             (r.integers(0, 65536, (29, 64), dtype=np.uint16), False, 29), (r.integers(0, 256, (33, 47, 3), dtype=np.uint8), False, 4),
             (r.integers(0, 256, (16, 16), dtype=np.uint8), True, 100)]
    for arr, be, rps in cases:
        f = tmp_path / "t.tif"
        f.write_bytes(_tiff_bytes(arr, be, rps))
        np.testing.assert_array_equal(decode(f), arr.astype(np.int64))
    # PIL's writer: 8-bit RGB and 16-bit gray, large enough for several strips
    rgb = r.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "p8.tif")
    np.testing.assert_array_equal(decode(tmp_path / "p8.tif"), rgb.astype(np.int64))
    g16 = r.integers(0, 65536, (260, 333), dtype=np.uint16)
    Image.fromarray(g16).save(tmp_path / "p16.tif")
    np.testing.assert_array_equal(decode(tmp_path / "p16.tif"), g16.astype(np.int64))
    # compressed / truncated: refused, no crash
    Image.fromarray(rgb).save(tmp_path / "lzw.tif", compression="tiff_lzw")
    assert subprocess.run([conv, str(tmp_path / "lzw.tif"), str(tmp_path / "x.ppm")], capture_output=True).returncode != 0
    (tmp_path / "cut.tif").write_bytes(_tiff_bytes(cases[0][0], True, 5)[:600])
    assert subprocess.run([conv, str(tmp_path / "cut.tif"), str(tmp_path / "x.ppm")], capture_output=True).returncode != 0


@pytest.mark.gpu
def test_cli_reads_tiff_frames(tmp_path):
    """The bundled city burst as TIFF files (the reference's fixed file names, content sniffed like cv::imread does): 8-bit
    RGB TIFFs give exactly the PNG run's result; 16-bit single-channel TIFFs are taken as raw RGGB frames (upper 12 bits)
    and give the same picture up to the white level (4095 instead of 255 * 16)."""
    import shutil
    from PIL import Image
    assert os.path.exists(CLI)
    src = os.path.join(ROOT, "tests", "golden", "city")
    runs = {}
    for kind in ("png", "tif8", "raw16"):
        d = tmp_path / kind
        d.mkdir()
        for i in range(5):
            im = np.asarray(Image.open(os.path.join(src, f"img_{i:06d}.png")).convert("RGB"))
            dst = d / f"img_{i:06d}.png"
            if kind == "png":
                shutil.copy(os.path.join(src, f"img_{i:06d}.png"), dst)
            elif kind == "tif8":
                dst.write_bytes(_tiff_bytes(im, False, 17))
            else:
                yy, xx = np.mgrid[0:im.shape[0], 0:im.shape[1]]
                mosaic = np.take_along_axis(im, ((yy & 1) + (xx & 1))[..., None], 2)[..., 0].astype(np.uint16) * 16   # 12-bit RGGB
                dst.write_bytes(_tiff_bytes((mosaic << 4).astype(np.uint16), True, 64))
        p = subprocess.run([CLI, "farneback", "city", "3"], cwd=d, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        runs[kind] = np.asarray(Image.open(d / "city_farneback_sr_result.png")).astype(np.int64)
    assert np.array_equal(runs["png"], runs["tif8"])
    d = np.abs(runs["png"] - runs["raw16"])
    # white level 4095 vs 4080: -0.4 % in value everywhere; a few pixels sit on the other side of a flow rounding / threshold
    assert d.mean() < 0.6 and (d > 3).mean() < 2e-3, (d.max(), d.mean(), (d > 3).mean())


@pytest.mark.gpu
def test_cli_car_burst_from_jpeg(tmp_path):
    """`multi_frame_sr farneback car 3` on the reference's own car/1..4.jpg (multi_frame_sr.cpp:155-159): decodes the JPEGs,
    fuses the four 130x228 frames (cropped to 128x228) and writes both result PNGs."""
    import shutil
    from PIL import Image
    assert os.path.exists(CLI)
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "car"), tmp_path / "car")
    p = subprocess.run([CLI, "farneback", "car", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "car/1.jpg, [130 x 228]" in p.stdout and "car/4.jpg, [130 x 228]" in p.stdout
    out = np.asarray(Image.open(tmp_path / "car_farneback_sr_result.png"))
    assert out.shape == (456, 256, 3)
    ref = np.asarray(Image.open(tmp_path / "car" / "1.jpg").convert("RGB"))[:, :128]
    # the x2 result is the reference frame up-sampled and denoised: its 2x2-binned version stays close to frame 1
    binned = out.reshape(228, 2, 128, 2, 3).mean((1, 3))
    assert np.abs(binned[8:-8, 8:-8] - ref[8:-8, 8:-8]).mean() < 12.0


def test_image_readers_under_address_and_ub_sanitizers(tmp_path):
    """apps/imgconv rebuilt with -fsanitize=address,undefined (CPU build: the only place sanitizers run on this pool) over
    truncated, header-corrupted and byte-fuzzed PNG / PNM / JPEG files: no sanitizer report, no crash, whatever the verdict."""
    import random
    import shutil
    import struct
    import zlib
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "imgconv_asan"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        os.path.join(ROOT, "apps", "imgconv.cpp"), "-o", str(exe), "-lz"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stderr[-200:])
    gold = os.path.join(ROOT, "tests", "golden")
    png = open(os.path.join(gold, "city/img_000000.png"), "rb").read()
    jpg = open(os.path.join(gold, "car/1.jpg"), "rb").read()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    cases = {
        "huge.png": b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 65536, 65536, 8, 6, 0, 0, 0))
                    + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""),
        "ws.pgm": b"P5 12 12   \n  ", "short.pgm": b"P5 4 4 255", "overflow.pgm": b"P5 99999999999999999999 4 255\n" + b"\0" * 64,
        "trunc.png": png[:5000], "good.png": png, "good.jpg": jpg, "trunc.jpg": jpg[:2000],
    }
    rnd = random.Random(1)
    for k in range(30):
        b = bytearray(jpg)
        for _ in range(8):
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        cases[f"fuzz{k}.jpg"] = bytes(b)
    for k in range(15):
        b = bytearray(png)
        for _ in range(6):
            b[rnd.randrange(40)] = rnd.randrange(256)
        cases[f"fuzz{k}.png"] = bytes(b)
    for name, data in cases.items():
        f = tmp_path / name
        f.write_bytes(data)
        p = subprocess.run([str(exe), str(f), str(tmp_path / "o.ppm")], capture_output=True, text=True)
        assert p.returncode in (0, 1), (name, p.returncode, p.stderr[-300:])
        assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, (name, p.stderr[-600:])
