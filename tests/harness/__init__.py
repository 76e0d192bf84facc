"""ctypes access to tests/harness/libngsc_harness.so: CPU unit-test harness for the NGSC_HD format
helpers of phyngsc_amd/csrc/ngsc_common.h (test infrastructure, g++; never loaded by the product)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngsc_harness.so")
_SRC = os.path.join(_HERE, "harness.cpp")
_HDR = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "phyngsc_amd", "csrc", "ngsc_common.h")
_lib = None


def lib():
    global _lib
    if _lib is None:
        stale = not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))
        if stale:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", _SO, _SRC])
        L = ctypes.CDLL(_SO)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.h_huffman.restype = u32
        L.h_huffman.argtypes = [vp, u32, vp, vp, ctypes.POINTER(u32)]
        L.h_title_stream.restype = ctypes.c_int64
        L.h_title_stream.argtypes = [vp, u32, vp, vp, u32, u32, vp, u64]
        L.h_plan_file.restype = ctypes.c_int64
        L.h_plan_file.argtypes = [vp, u64, u32, vp, u32]
        _lib = L
    return _lib
