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
        deps = [SRC] + [os.path.join(ROOT, "osmo-tetra_amd", "csrc", h) for h in ("vit_core.h", "slot_core.h", "tg_layout.h", "tg_conv.h")]
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


def decode_block_soft(kind, soft4, maskwords=None, packed=False, want_max=False):
    """kind 0/1/2; soft4 = int8 values in type-4 (stream) order.  returns (type2 bits, crc); packed: the 16-bit
    trellis the kernels run (tg_pvit_*) instead of the 32-bit statement (tg_svit_*); want_max: also the largest
    12-bit metric seen"""
    nblk = {0: 10, 1: 18, 2: 36}[kind]
    s4 = np.ascontiguousarray(soft4, np.int8)
    area = np.zeros(8 + 12 * nblk + 8, np.int8)
    out = np.zeros(288, np.uint8)
    i8p = C.POINTER(C.c_int8)
    lib().emul_soft_layout(kind, s4.ctypes.data_as(i8p), area.ctypes.data_as(i8p))
    mw = None if maskwords is None else np.ascontiguousarray(maskwords, np.uint32).ctypes.data_as(u32p)
    if packed:
        mx = C.c_uint(0)
        lib().emul_decode_psoft.restype = C.c_uint
        crc = lib().emul_decode_psoft(kind, area.ctypes.data_as(i8p), mw, out.ctypes.data_as(u8p), C.byref(mx))
        if want_max:
            return out, crc, mx.value
        return out, crc
    crc = lib().emul_decode_soft(kind, area.ctypes.data_as(i8p), mw, out.ctypes.data_as(u8p))
    return out, crc


def conv_decode(pu, mother, type3, type2_len):
    """generic trellis: puncturer pu on the rate-1/mother code; None for a rejected shape"""
    t3 = np.ascontiguousarray(type3, np.uint8)
    out = np.zeros(type2_len, np.uint8)
    rc = lib().emul_conv_decode(pu, mother, len(t3), type2_len, t3.ctypes.data_as(u8p), out.ctypes.data_as(u8p))
    return None if rc else out


# ---- classification words / SYNC summaries of a 0 / 1 stream (tests/host_emul/cls_emul.c) ----
CLS_SRC = os.path.join(HERE, "host_emul", "cls_emul.c")
CLS_LIB = os.path.join(HERE, "host_emul", "libcls_emul.so")
_cls_lib = None


def cls_ysum(stream, anchor, chunk, view=None):
    """(cls, ysum) as tests/test_stream_sync_cpu.py's emul_cls / emul_ysum give them, in C: streams of bench size"""
    global _cls_lib
    if _cls_lib is None:
        if not os.path.exists(CLS_LIB) or os.path.getmtime(CLS_SRC) > os.path.getmtime(CLS_LIB):
            subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", CLS_SRC, "-o", CLS_LIB])
        _cls_lib = C.CDLL(CLS_LIB)
        _cls_lib.emul_cls_ysum.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u32p, C.POINTER(C.c_uint16)]
        _cls_lib.emul_cls_ysum.restype = None
    if view is None:        # TG_VIEW_OF (csrc/tg_layout.h)
        view = 640 if chunk <= 64 else 832 if chunk <= 128 else 1088
    s = np.ascontiguousarray(stream, np.uint8)
    L = len(s)
    n = (L - anchor) // 510 if L >= anchor + 510 else 0
    cls = np.zeros(max(n, 1), np.uint32)
    ys = np.zeros(max(n, 1), np.uint16)
    _cls_lib.emul_cls_ysum(s.ctypes.data_as(u8p), L, anchor, chunk, view, cls.ctypes.data_as(u32p),
                           ys.ctypes.data_as(C.POINTER(C.c_uint16)))
    return cls[:n], ys[:n]


def decode_slot(btype, words20):
    """the lane-per-slot schedule (slot_core.h) on a packed slot whose blocks are descrambled: returns (decoded bytes [36], crc [2],
    record bytes 48..319 [272], SYNC PDU words [3])"""
    w = np.ascontiguousarray(words20, np.uint32)
    od = np.zeros(36, np.uint8)
    crc = np.zeros(2, np.uint32)
    bits = np.zeros(272, np.uint8)
    sy = np.zeros(3, np.uint32)
    rc = lib().emul_decode_slot(int(btype), w.ctypes.data_as(u32p), od.ctypes.data_as(u8p), crc.ctypes.data_as(u32p),
                                bits.ctypes.data_as(u8p), sy.ctypes.data_as(u32p))
    assert rc == 0
    return od, crc, bits, sy
