"""CPU-side checks of the drop-in boundary (no GPU needed): the C-ABI library
loads, exports every symbol include/mfsr.h declares, validates arguments before
touching the device, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from multi_frame_super_resolution_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_parses_every_prototype():
    protos = capi.parse_header()
    text = open(capi.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    declared = set(re.findall(r"\b(mfsr_[A-Za-z0-9_]+)\s*\(", text))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 70


def test_library_exports_every_declared_symbol():
    L = capi.lib()  # raises if the .so is missing or lacks a declared symbol
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.path], text=True)
    exported = set(re.findall(r" T (mfsr_\w+)", out))
    assert set(L.protos) <= exported, set(L.protos) - exported
    assert L.version() == 100


def test_reference_kernel_names_are_all_present():
    """One entry point per on-path reference kernel, same names."""
    names = """deBayerGreenKernel deBayerRedBlueKernel deBayersSubSample3 accumulateImages accumulateImagesSuperRes
    squaredSum boxFilterWithBorderX boxFilterWithBorderY normalizedCC convertToTilesOverlapBorder
    convertToTilesOverlapPreShift GammasRGB ApplyWeighting conjugateComplexMulKernel findMinimum UpSampleShifts
    ComputeStructureTensor ComputeKernelParam fourierFilter fftshift copyShiftMatrix setPointers checkForOutliers
    transposeShifts getOptimalShifts concatenateShifts separateShifts WarpingKernel CreateFlowFieldFromTiles
    ComputeDerivativesKernel ComputeDerivatives2Kernel lucasKanadeOptim ComputeRobustnessMask""".split()
    protos = capi.parse_header()
    for n in names:
        assert "mfsr_" + n in protos, n


def test_argument_validation_happens_on_the_host():
    L = capi.lib()
    raw = L.raw
    # null pointers / bad sizes are rejected with MFSR_E_INVALID before any HIP call
    assert raw["mfsr_deBayersSubSample3"](None, None, 1.0, 4, 4, 48, None) == -1
    assert raw["mfsr_squaredSum"](None, None, 1, 8, 1, None) == -1
    assert raw["mfsr_set_cfa_pattern"](None) == -1
    bad = (ctypes.c_int32 * 4)(0, 1, 1, 9)
    assert raw["mfsr_set_cfa_pattern"](bad) == -1
    ok = (ctypes.c_int32 * 4)(2, 1, 1, 0)
    assert raw["mfsr_set_cfa_pattern"](ok) == 0
    got = (ctypes.c_int32 * 4)()
    assert raw["mfsr_get_cfa_pattern"](got) == 0 and list(got) == [2, 1, 1, 0]
    assert raw["mfsr_set_cfa_pattern"]((ctypes.c_int32 * 4)(0, 1, 1, 2)) == 0
    # entry points added for the MI355X path validate the same way
    assert raw["mfsr_tileSquaredSums"](None, None, 64, 64, 256, 4, 32, 2, 2, None) == -1
    assert raw["mfsr_zeroRing_f32x4"](None, 64, 4, 4, None) == -1
    assert raw["mfsr_burst_flush"](None, None) == -1
    z3, t0 = capi.Float3(0, 0, 0), capi.Tex2D(None, 0, 0, 0)
    assert raw["mfsr_accumulateSuperResFull2"](None, None, None, None, None, None, t0, t0, t0, z3, z3, 64, 64, 2, 768, 512,
                                               None) == -1
    cfgd = capi.Config()
    assert raw["mfsr_config_default"](ctypes.byref(cfgd), 256, 192, 4, 2, 0) == 0
    assert cfgd.pairFrames == 1 and cfgd.asyncFuse == 0 and cfgd.preAlign == 0 and cfgd.uploadRing == 0
    assert L.raw["mfsr_error_string"](-1) == b"invalid argument"
    assert L.raw["mfsr_error_string"](0) == b"success"


def test_config_default_and_workspace():
    L = capi.lib()
    cfg = capi.Config()
    assert L.raw["mfsr_config_default"](ctypes.byref(cfg), 3840, 2160, 16, 2, 0) == 0
    assert (cfg.width, cfg.height, cfg.frames, cfg.scale, cfg.mono) == (3840, 2160, 16, 2, 0)
    assert list(cfg.cfa) == [0, 1, 1, 2] and cfg.fused == 1 and cfg.levelFactor[cfg.levels - 1] == 1
    ws = L.raw["mfsr_burst_workspace_bytes"](ctypes.byref(cfg))
    acc = L.raw["mfsr_burst_accumulator_bytes"](ctypes.byref(cfg))
    assert acc == 12 * 7680 * 4320          # 398 MB per plane-set
    assert 100e6 < ws < 2e9
    cfg.fused = 0   # the unfused chain: its own scratch images instead of the per-frame align sets of the batched path
    assert 100e6 < L.raw["mfsr_burst_workspace_bytes"](ctypes.byref(cfg)) < 2e9
    cfg.width = 30                           # invalid geometry -> 0 bytes
    assert L.raw["mfsr_burst_workspace_bytes"](ctypes.byref(cfg)) == 0


def test_config_struct_mirrors_the_header():
    """capi.Config must have the layout of struct mfsr_config: compile a probe with gcc."""
    probe = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "mfsr.h"
    int main(void){ printf("%zu %zu %zu %zu %zu\n", sizeof(mfsr_config), offsetof(mfsr_config, maxVal),
        offsetof(mfsr_config, lkIterations), offsetof(mfsr_config, fused), sizeof(mfsr_tex2d)); return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(probe)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        size, o_max, o_lk, o_fused, tex = map(int, subprocess.check_output([exe], text=True).split())
    assert ctypes.sizeof(capi.Config) == size
    assert capi.Config.maxVal.offset == o_max
    assert capi.Config.lkIterations.offset == o_lk
    assert capi.Config.fused.offset == o_fused
    assert ctypes.sizeof(capi.Tex2D) == tex


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    L = capi.lib()
    assert L.device_count() == 0
    cfg = capi.Config()
    L.raw["mfsr_config_default"](ctypes.byref(cfg), 256, 128, 3, 2, 0)
    handle = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(4096)
    addr = (ctypes.addressof(buf) + 255) // 256 * 256
    rc = L.raw["mfsr_burst_create"](ctypes.byref(handle), ctypes.byref(cfg), addr, 1 << 40)
    assert rc == -3  # MFSR_E_NODEVICE
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline
    with pytest.raises(RuntimeError):
        BurstPipeline(cfg)


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing in the package or apps/ may reference it."""
    pkg = os.path.join(ROOT, "multi_frame_super_resolution_amd")
    for base in (pkg, os.path.join(ROOT, "apps")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                    assert "libmfsr_oracle" not in txt and "orc_" not in txt, f


def test_dist_library_loads_and_exports_every_declared_symbol():
    """include/mfsr_dist.h <-> lib/libmfsr_dist.so (RCCL layer); no compute without a GPU."""
    D = capi.dist_lib()
    out = subprocess.check_output(["nm", "-D", "--defined-only", D.path], text=True)
    exported = set(re.findall(r" T (mfsr_\w+)", out))
    assert set(D.protos) <= exported, set(D.protos) - exported
    assert {"mfsr_dist_create", "mfsr_dist_process_burst", "mfsr_dist_get_unique_id"} <= set(D.protos)
    needed = subprocess.check_output(["readelf", "-d", D.path], text=True)
    assert "librccl" in needed and "libmfsr_hip" in needed
    cfg = capi.Config()
    assert capi.lib().raw["mfsr_config_default"](ctypes.byref(cfg), 256, 192, 4, 2, 0) == 0
    assert D.raw["mfsr_dist_workspace_bytes"](ctypes.byref(cfg), 2) > capi.lib().raw["mfsr_burst_workspace_bytes"](ctypes.byref(cfg))
    assert D.raw["mfsr_dist_process_burst"](None, None, 0, None, None, None) == -1


@pytest.mark.parametrize("W,H,scale,mono", [(3840, 2160, 2, 0), (3840, 2160, 4, 0), (1920, 1080, 2, 1), (328, 200, 4, 0), (256, 192, 3, 0)])
def test_stripe_plan_covers_the_frame_and_every_row_the_fuse_reads(W, H, scale, mono):
    """mfsr_dist_stripe_plan: stripes tile the HR rows on 16-row boundaries, and the flow / certainty / raw row ranges
    contain every row the fuse of a stripe can read (bilinear field fetch, 5x5 footprint, vertical flow up to maxFlowY)."""
    L = capi.lib()
    cfg = capi.Config()
    assert L.raw["mfsr_config_default"](ctypes.byref(cfg), W, H, 16, scale, mono) == 0
    hrH = H * scale
    th = H if mono else H // 2
    hh = H // 2
    for world in (1, 2, 3, 8, 64):
        halo = 64
        plans = []
        for r in range(world):
            p = capi.StripePlan()
            assert L.raw["mfsr_dist_stripe_plan"](ctypes.byref(cfg), world, r, halo, ctypes.byref(p)) == 0
            plans.append(p)
        assert plans[0].rowBegin == 0 and plans[-1].rowEnd == hrH
        for a, b in zip(plans, plans[1:]):
            assert a.rowEnd == b.rowBegin and a.rowEnd % 16 == 0
        for p in plans:
            if p.rowEnd <= p.rowBegin:
                continue
            assert p.maxFlowY == halo - 3
            for Y in (p.rowBegin, p.rowEnd - 1):
                # field rows of the bilinear fetch at HR row Y (and of the tile kernels' 3-row staging)
                for fh, r0, n in ((th, p.flowRow0, p.flowRows), (hh, p.maskRow0, p.maskRows)):
                    yb = (Y + 0.5) / hrH * fh - 0.5
                    lo, hi = max(int(np.floor(yb)), 0), min(int(np.floor(yb)) + 1, fh - 1)
                    per = hrH // fh
                    lo, hi = max(min(lo, Y // per - 1), 0), min(max(hi, Y // per + 1), fh - 1)
                    assert r0 <= lo and hi < r0 + n
                # certainty sites of the 5x5 footprint, raw rows for |v| <= maxFlowY
                for py in (-2, 2):
                    site = min(max((Y + py) // scale, 0), H - 1) // 2
                    assert p.maskRow0 <= site < p.maskRow0 + p.maskRows
                    for v in (-p.maxFlowY, p.maxFlowY):
                        row = min(max((Y + py + int(round(scale * v))) // scale, 0), H - 1)
                        assert p.rawRow0 <= row < p.rawRow0 + p.rawRows
                        assert p.rawRow0 <= min(row + 2, H - 1) < p.rawRow0 + p.rawRows   # the 3x3 raw sites of a pixel


def test_exact_division_check_is_host_only_and_agrees_with_numpy():
    """mfsr_exactDivisionOk (include/mfsr.h): the host-side proof the kernels rely on before they replace `x / d` by the
    reciprocal sequence.  Runs without a device.  For divisors that pass, the sequence evaluated in numpy float32 equals the
    float32 division on a million random numerators (flows up to +-1e4 px, negative and tiny values included)."""
    import ctypes
    import numpy as np
    from multi_frame_super_resolution_amd import capi
    L = ctypes.CDLL(capi.LIB_PATH)
    L.mfsr_exactDivisionOk.argtypes = [ctypes.c_float]
    L.mfsr_exactDivisionOk.restype = ctypes.c_int
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1e4, 1e4, 500000), rng.uniform(-2.0, 2.0, 400000), rng.standard_normal(100000) * 1e-6]).astype(np.float32)
    for d in (1920.0, 1080.0, 3840.0, 2160.0, 960.0, 540.0, 7680.0, 4320.0, 392.0, 264.0, 328.0, 200.0, 162.0, 3.0):
        assert L.mfsr_exactDivisionOk(d) == 1, d
        dd, r = np.float32(d), np.float32(1.0) / np.float32(d)
        q = x * r
        # fma in float64: the products of two float32 are exact there, one rounding to float32 at the end of each step
        rem = (x.astype(np.float64) - dd.astype(np.float64) * q.astype(np.float64)).astype(np.float32)
        q2 = (q.astype(np.float64) + rem.astype(np.float64) * np.float64(r)).astype(np.float32)
        assert np.array_equal(q2, x / dd), d
    assert L.mfsr_exactDivisionOk(0.0) == 0 and L.mfsr_exactDivisionOk(-5.0) == 0 and L.mfsr_exactDivisionOk(3.0e8) == 0
