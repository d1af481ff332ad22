"""ctypes binding of oracle/libmfsr_oracle.so (numpy in / numpy out).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Prototypes are parsed from
the oracle's C sources so the binding cannot drift from them.
"""
from __future__ import annotations

import ctypes
import glob
import os
import re
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libmfsr_oracle.so")

_SCALAR = {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "int32_t": ctypes.c_int32}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def _parse():
    protos = {}
    for src in sorted(glob.glob(os.path.join(_DIR, "*.c"))):
        text = open(src).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        for m in re.finditer(r"^(void|int)\s+(orc_\w+)\s*\(([^)]*)\)\s*\{", text, flags=re.M | re.S):
            args = []
            for a in re.sub(r"\s+", " ", m.group(3)).split(","):
                a = a.strip()
                if not a or a == "void":
                    continue
                am = re.match(r"^(.*?)(\w+)$", a)
                args.append((am.group(1).strip(), am.group(2)))
            protos[m.group(2)] = (m.group(1), args)
    return protos


class _Oracle:
    def __init__(self):
        build()
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = _parse()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)
            fn.restype = ctypes.c_int if ret == "int" else None
            fn.argtypes = [ctypes.c_void_p if "*" in t else _SCALAR[t.replace("const", "").strip()] for t, _ in args]

    def __getattr__(self, name):
        fn = getattr(self.__dict__["cdll"], "orc_" + name)

        def call(*a):
            conv = []
            for v in a:
                if isinstance(v, np.ndarray):
                    assert v.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
                    conv.append(v.ctypes.data)
                else:
                    conv.append(v)
            return fn(*conv)

        return call


_o = None


def oracle() -> _Oracle:
    global _o
    if _o is None:
        _o = _Oracle()
    return _o
