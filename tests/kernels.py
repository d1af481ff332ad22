"""Uniform numpy-level access to the CPU oracle and to the HIP C-ABI.

Both runners expose ``call(name, *args)`` with the SAME argument list (the
reference kernel's argument order); numpy arrays are device buffers that the
kernel may update in place.  Marker classes cover the places where the two
C signatures differ only in spelling:

    F3(v)          float3 by value (HIP)        / const float[3] (oracle)
    F2(v)          float2 by value (HIP)        / two floats (oracle)
    Tex(a, w, h)   mfsr_tex2d by value (HIP)    / ptr, pitch, w, h (oracle)
    Host(a)        host array on both sides (e.g. filter taps)

``OracleKernels`` is test infrastructure; ``HipKernels`` is the product path
(torch is used only to own device memory).
"""
from __future__ import annotations

import ctypes

import numpy as np


class F3:
    def __init__(self, v):
        self.v = np.asarray(v, np.float32).copy()


class F2:
    def __init__(self, v):
        self.v = np.asarray(v, np.float32).copy()


class Tex:
    def __init__(self, arr, width=None, height=None, pitch=None):
        self.arr = arr
        self.height = arr.shape[0] if height is None else height
        self.width = arr.shape[1] if width is None else width
        self.pitch = arr.strides[0] if pitch is None else pitch


class Host:
    def __init__(self, arr):
        self.arr = np.ascontiguousarray(arr)


def pitch_of(a: np.ndarray) -> int:
    return int(a.strides[0])


class OracleKernels:
    name = "oracle"

    def __init__(self):
        from oracle.bindings import oracle

        self.o = oracle()

    def call(self, fname, *args):
        conv = []
        keep = []
        for a in args:
            if isinstance(a, F3):
                keep.append(a.v)
                conv.append(a.v)
            elif isinstance(a, F2):
                conv.extend([float(a.v[0]), float(a.v[1])])
            elif isinstance(a, Tex):
                conv.extend([a.arr, int(a.pitch), int(a.width), int(a.height)])
            elif isinstance(a, Host):
                conv.append(a.arr)
            elif a is None:
                conv.append(None)
            else:
                conv.append(a)
        return getattr(self.o, fname)(*conv)

    def set_cfa(self, pattern):
        self.o.set_cfa_pattern(np.asarray(pattern, np.int32))


class HipKernels:
    name = "hip"

    def __init__(self):
        import torch

        from multi_frame_super_resolution_amd import capi

        self.torch = torch
        self.capi = capi
        self.L = capi.lib()
        self.dev = torch.device("cuda:0")

    def set_cfa(self, pattern):
        arr = (ctypes.c_int32 * 4)(*[int(p) for p in pattern])
        self.L.set_cfa_pattern(arr)

    def call(self, fname, *args):
        torch = self.torch
        uploaded = {}

        def up(a: np.ndarray):
            key = id(a)
            if key not in uploaded:
                assert a.flags["C_CONTIGUOUS"]
                if a.dtype == np.uint16:
                    t = torch.from_numpy(a.view(np.int16)).to(self.dev)
                else:
                    t = torch.from_numpy(a).to(self.dev)
                uploaded[key] = (a, t)
            return uploaded[key][1]

        conv = []
        for a in args:
            if isinstance(a, F3):
                conv.append(self.capi.f3(a.v))
            elif isinstance(a, F2):
                conv.append(self.capi.f2(a.v))
            elif isinstance(a, Tex):
                t = up(a.arr)
                conv.append(self.capi.Tex2D(t.data_ptr(), int(a.pitch), int(a.width), int(a.height)))
            elif isinstance(a, Host):
                conv.append(a.arr.ctypes.data)
            elif isinstance(a, np.ndarray):
                conv.append(up(a).data_ptr())
            elif a is None:
                conv.append(None)
            else:
                conv.append(a)
        conv.append(None)  # stream: default
        rc = getattr(self.L, fname)(*conv)
        torch.cuda.synchronize()
        for a, t in uploaded.values():
            if a.flags["WRITEABLE"]:
                h = t.cpu().numpy()
                if a.dtype == np.uint16:
                    h = h.view(np.uint16)
                np.copyto(a, h)
        return rc
