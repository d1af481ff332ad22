"""ctypes binding of the C-ABI declared in ``include/mfsr.h``.

The product path is the HIP shared library ``lib/libmfsr_hip.so`` and nothing
else: there is no CPU fallback.  If the library is missing or no HIP device is
present, every compute entry point fails loudly.

Signatures are parsed from ``include/mfsr.h`` itself so that the binding cannot
drift from the header (the header is the contract; each prototype there cites
the reference kernel it replaces, e.g. ``accumulateImagesSuperRes`` =
reference ``test_opencv/DeBayerKernels.cu:379``).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
HEADER_PATH = os.path.join(_ROOT, "include", "mfsr.h")
# MFSR_LIB overrides the library path (A/B builds of the same C-ABI during kernel tuning)
LIB_PATH = os.environ.get("MFSR_LIB") or os.path.join(_PKG_DIR, "lib", "libmfsr_hip.so")


class Float2(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float)]


class Float3(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float)]


class Float4(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float), ("w", ctypes.c_float)]


class Tex2D(ctypes.Structure):
    """mfsr_tex2d: stand-in for cudaTextureObject_t."""

    _fields_ = [("ptr", ctypes.c_void_p), ("pitch", ctypes.c_int32), ("width", ctypes.c_int32),
                ("height", ctypes.c_int32)]


class PreAlign(ctypes.Structure):
    """mfsr_prealign (include/mfsr.h)."""

    _fields_ = [("shiftX", ctypes.c_float), ("shiftY", ctypes.c_float), ("rotation", ctypes.c_float),
                ("cosRotation", ctypes.c_float), ("sinRotation", ctypes.c_float), ("angleIndex", ctypes.c_int32),
                ("tx", ctypes.c_int32), ("ty", ctypes.c_int32), ("level", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


class LkFrame(ctypes.Structure):
    """mfsr_lk_frame (include/mfsr.h): one frame of mfsr_lucasKanadeSweepBatch."""

    _fields_ = [("shiftsIn", ctypes.c_void_p), ("shiftsOut", ctypes.c_void_p), ("movedImg", ctypes.c_void_p),
                ("sumIn", ctypes.c_void_p), ("diffIn", ctypes.c_void_p), ("sumOut", ctypes.c_void_p), ("diffOut", ctypes.c_void_p)]


class PrepareFrame(ctypes.Structure):
    """mfsr_prepare_frame"""
    _fields_ = [("dataIn", ctypes.c_void_p), ("halfOut", ctypes.c_void_p), ("pyr0", ctypes.c_void_p), ("pyr1", ctypes.c_void_p)]


class TrackFrame(ctypes.Structure):
    """mfsr_track_frame"""
    _fields_ = [("movedImg", ctypes.c_void_p), ("coarseShifts", ctypes.c_void_p), ("coordinates", ctypes.c_void_p), ("base", ctypes.c_void_p)]


class FlowFieldFrame(ctypes.Structure):
    """mfsr_flowfield_frame"""
    _fields_ = [("outImg", ctypes.c_void_p), ("tileShifts", ctypes.c_void_p), ("base", ctypes.c_void_p), ("movedImg", ctypes.c_void_p),
                ("sumOut", ctypes.c_void_p), ("diffOut", ctypes.c_void_p)]


class RobustnessFrame(ctypes.Structure):
    """mfsr_robustness_frame"""
    _fields_ = [("movedHalf", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("flow", ctypes.c_void_p)]


class Config(ctypes.Structure):
    """mfsr_config (include/mfsr.h)."""

    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("frames", ctypes.c_int32),
        ("reference", ctypes.c_int32), ("scale", ctypes.c_int32), ("mono", ctypes.c_int32),
        ("cfa", ctypes.c_int32 * 4), ("black", ctypes.c_float * 3), ("white", ctypes.c_float * 3),
        ("maxVal", ctypes.c_float),
        ("levels", ctypes.c_int32), ("levelFactor", ctypes.c_int32 * 4), ("tileSize", ctypes.c_int32 * 4),
        ("maxShift", ctypes.c_int32 * 4), ("minimumThreshold", ctypes.c_float), ("sigmaTracking", ctypes.c_float),
        ("lkIterations", ctypes.c_int32), ("lkHalfWindow", ctypes.c_int32), ("lkMinDet", ctypes.c_float),
        ("alpha", ctypes.c_float), ("beta", ctypes.c_float), ("thresholdM", ctypes.c_float),
        ("sigmaTensor", ctypes.c_float),
        ("Dth", ctypes.c_float), ("Dtr", ctypes.c_float), ("kDetail", ctypes.c_float), ("kDenoise", ctypes.c_float),
        ("kStretch", ctypes.c_float), ("kShrink", ctypes.c_float),
        ("weightThreshold", ctypes.c_float), ("applyGamma", ctypes.c_int32), ("fused", ctypes.c_int32),
        ("pairFrames", ctypes.c_int32), ("asyncFuse", ctypes.c_int32),
        ("preAlign", ctypes.c_int32), ("preAlignMaxAngle", ctypes.c_float), ("uploadRing", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 2),
    ]


_BY_VALUE = {
    "mfsr_float2": Float2, "mfsr_float3": Float3, "mfsr_float4": Float4, "mfsr_tex2d": Tex2D,
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "mfsr_stream_t": ctypes.c_void_p, "long long": ctypes.c_longlong,
}
_RET = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "void": None, "const char*": ctypes.c_char_p,
        "mfsr_burst*": ctypes.c_void_p, "mfsr_dist*": ctypes.c_void_p}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """Return {name: (return_type, [(ctype_string, arg_name), ...])} for every
    ``mfsr_*`` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"#[^\n]*", " ", text)                       # preprocessor lines
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    text = re.sub(r"\{[^{}]*\}", " ", text)                    # struct / enum bodies
    protos = {}
    for stmt in text.split(";"):
        m = re.match(r"^\s*((?:const\s+char\s*\*|mfsr_burst\s*\*|mfsr_dist\s*\*|int|size_t|void))\s+(mfsr_\w+)\s*\((.*)\)\s*$", stmt, flags=re.S)
        if not m:
            continue
        ret = re.sub(r"\s+", " ", m.group(1)).replace(" *", "*").strip()
        name = m.group(2)
        args = []
        argtxt = re.sub(r"\s+", " ", m.group(3)).strip()
        if argtxt and argtxt != "void":
            for a in argtxt.split(","):
                a = a.strip()
                am = re.match(r"^(.*?)(\w+)(\[\d*\])?$", a)
                typ = am.group(1).strip()
                if am.group(3):
                    typ += "*"
                typ = typ.replace(" *", "*").replace("* ", "*")
                args.append((typ, am.group(2)))
        protos[name] = (ret, args)
    return protos


def _ctype_of(typ: str):
    if "*" in typ:
        return ctypes.c_void_p
    base = typ.replace("const", "").strip()
    if base not in _BY_VALUE:
        raise TypeError(f"mfsr.h: unhandled parameter type {typ!r}")
    return _BY_VALUE[base]


class MfsrError(RuntimeError):
    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed with code {code}: {msg}")
        self.code = code


class StripePlan(ctypes.Structure):
    """mfsr_stripe_plan (include/mfsr.h)."""

    _fields_ = [("rowBegin", ctypes.c_int32), ("rowEnd", ctypes.c_int32), ("flowRow0", ctypes.c_int32),
                ("flowRows", ctypes.c_int32), ("maskRow0", ctypes.c_int32), ("maskRows", ctypes.c_int32),
                ("rawRow0", ctypes.c_int32), ("rawRows", ctypes.c_int32), ("maxFlowY", ctypes.c_float),
                ("reserved", ctypes.c_int32 * 3)]


DIST_HEADER_PATH = os.path.join(_ROOT, "include", "mfsr_dist.h")
DIST_LIB_PATH = os.path.join(_PKG_DIR, "lib", "libmfsr_dist.so")
DIST_STRIPES, DIST_REDUCE, DIST_REDUCE_SCATTER = 0, 1, 2
DIST_ID_BYTES = 128


class _Lib:
    """Loaded C-ABI; attribute access returns checked wrappers (raise on rc != 0)."""

    def __init__(self, path: str = LIB_PATH, header: str = HEADER_PATH, mode: int = ctypes.DEFAULT_MODE):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        self.path = path
        self.cdll = ctypes.CDLL(path, mode=mode)
        self.protos = parse_header(header)
        self.raw = {}
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = _RET[ret]
            fn.argtypes = [_ctype_of(t) for t, _ in args]
            self.raw[name] = fn

    def __getattr__(self, name: str):
        raw = self.__dict__["raw"].get("mfsr_" + name) or self.__dict__["raw"].get(name)
        if raw is None:
            raise AttributeError(name)
        full = raw.__name__ if hasattr(raw, "__name__") else name
        ret = self.protos[full][0]
        if ret != "int" or full in ("mfsr_version", "mfsr_device_count", "mfsr_gaussin_filter_1D", "mfsr_burst_group_size",
                                     "mfsr_trackTilesFastSupported"):
            return raw

        def checked(*a):
            rc = raw(*a)
            if rc != 0:
                es = self.raw.get("mfsr_error_string") or lib().raw["mfsr_error_string"]
                msg = es(rc)
                raise MfsrError(full, rc, msg.decode() if msg else "?")
            return rc

        checked.__name__ = full
        return checked


_lib = None
_dist_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def dist_lib() -> _Lib:
    """libmfsr_dist.so (include/mfsr_dist.h): the multi-GPU layer, RCCL + libmfsr_hip.so."""
    global _dist_lib
    if _dist_lib is None:
        lib()   # libmfsr_hip.so first, so that both bind to the one copy of its process-wide state
        _dist_lib = _Lib(DIST_LIB_PATH, DIST_HEADER_PATH)
    return _dist_lib


def f3(v) -> Float3:
    return Float3(float(v[0]), float(v[1]), float(v[2]))


def f2(v) -> Float2:
    return Float2(float(v[0]), float(v[1]))


def tex(t, width: int | None = None, height: int | None = None, pitch: int | None = None) -> Tex2D:
    """Texture descriptor over a torch tensor laid out [H, W(, C)] (dense rows
    unless pitch is given)."""
    h = t.shape[0] if height is None else height
    w = t.shape[1] if width is None else width
    p = t.stride(0) * t.element_size() if pitch is None else pitch
    return Tex2D(t.data_ptr(), p, w, h)
