"""Host build of the product's per-lane trellis code (tests/host_emul/vit_emul.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_emul", "vit_emul.cpp")
LIB = os.path.join(HERE, "host_emul", "libvit_emul.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [SRC] + [os.path.join(ROOT, "osmo-tetra_amd", "csrc", h) for h in ("vit_core.h", "tg_layout.h")]
        if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
            subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared",
                                   "-I" + os.path.join(ROOT, "osmo-tetra_amd", "csrc"), SRC, "-o", LIB])
        _lib = C.CDLL(LIB)
        _lib.emul_decode_words.restype = C.c_uint
    return _lib


def decode_block(kind, type4):
    """kind 0/1/2 (SB1/216/432); type4 = descrambled bits.  returns (type2 bits, crc)"""
    t4 = np.ascontiguousarray(type4, np.uint8)
    w = np.zeros(18, np.uint32)
    out = np.zeros(288, np.uint8)
    lib().emul_pack_block(kind, t4.ctypes.data_as(u8p), w.ctypes.data_as(u32p))
    crc = lib().emul_decode_words(kind, w.ctypes.data_as(u32p), out.ctypes.data_as(u8p))
    return out, crc


def pack_slot(btype, slot):
    s = np.ascontiguousarray(slot, np.uint8)
    w = np.zeros(20, np.uint32)
    lib().emul_pack_slot(btype, s.ctypes.data_as(u8p), w.ctypes.data_as(u32p))
    return w
