"""ctypes access to the oracle (oracle/liboracle.so) and, when it has been built in
this container, to the real reference objects (oracle/_ref/libtetra_ref.so).

TEST INFRASTRUCTURE.  Nothing under osmo-tetra_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFERENCE_SRC = "/root/reference/src"

# enum values (mirrors of phy/tetra_burst.h)
TRAIN_NORM_1, TRAIN_NORM_2, TRAIN_NORM_3, TRAIN_SYNC, TRAIN_EXT = range(5)
T_SB1, T_SB2, T_NDB, T_BBK, T_SCH_HU, T_SCH_F = range(6)
LC_UNKNOWN, LC_SCH_F, LC_AACH, LC_BSCH, LC_BNCH = 0, 1, 8, 10, 11
CRC_OK = 0x1D0F

BLK = {  # type345, type2, type1, a   (lower_mac/tetra_lower_mac.c:55-102)
    T_SB1: (120, 80, 60, 11),
    T_SB2: (216, 144, 124, 101),
    T_NDB: (216, 144, 124, 101),
    T_BBK: (30, 30, 14, 0),
    T_SCH_HU: (168, 112, 92, 13),
    T_SCH_F: (432, 288, 268, 103),
}

u8p = C.POINTER(C.c_uint8)


def _p(a):
    return a.ctypes.data_as(u8p)


def build_oracle():
    """(re)build liboracle.so, and the _ref objects when the reference tree is present."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if os.path.isdir(REFERENCE_SRC):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


class TdmaTime(C.Structure):
    _fields_ = [("hn", C.c_uint16), ("sn", C.c_uint32), ("tn", C.c_uint32), ("fn", C.c_uint32), ("mn", C.c_uint32)]

    def tup(self):
        return (self.hn, self.sn, self.tn, self.fn, self.mn)


class Record(C.Structure):
    _fields_ = [
        ("burst_seq", C.c_uint32),
        ("burst_type", C.c_uint8),
        ("type", C.c_uint8),
        ("blk_num", C.c_uint8),
        ("lchan", C.c_uint8),
        ("crc_ok", C.c_uint8),
        ("traffic_dumped", C.c_uint8),
        ("crc", C.c_uint16),
        ("scrambling_code", C.c_uint32),
        ("time", TdmaTime),
        ("type1_len", C.c_uint16),
        ("type1", C.c_uint8 * 268),
        ("type4", C.c_uint8 * 432),
        ("time_str", TdmaTime),
    ]


class Rx(C.Structure):
    pass


UPPER_CB = C.CFUNCTYPE(C.c_int, C.POINTER(Rx), C.POINTER(Record), C.c_uint, C.c_void_p)
EVENT_CB = C.CFUNCTYPE(None, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p)

Rx._fields_ = [
    ("state", C.c_int),
    ("bits_in_buf", C.c_uint),
    ("bitbuf", C.c_uint8 * 4096),
    ("bitbuf_start_bitnum", C.c_uint),
    ("next_frame_start_bitnum", C.c_uint),
    ("phy_time", TdmaTime),
    ("mcc", C.c_uint16),
    ("mnc", C.c_uint16),
    ("colour_code", C.c_uint8),
    ("cell_time", TdmaTime),
    ("scramb_init", C.c_uint32),
    ("is_traffic", C.c_int),
    ("blk1_stolen", C.c_int),
    ("blk2_stolen", C.c_int),
    ("use_acc", C.c_int),
    ("burst_seq", C.c_uint32),
    ("cur_burst_type", C.c_uint8),
    ("upper", UPPER_CB),
    ("event", EVENT_CB),
    ("priv", C.c_void_p),
]


class BlockResult(C.Structure):
    _fields_ = [
        ("type1", C.c_uint8 * 432),
        ("type2", C.c_uint8 * 288),
        ("type4", C.c_uint8 * 432),
        ("crc", C.c_uint16),
        ("crc_ok", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build_oracle()
    L = C.CDLL(path)
    L.orc_scramb_get_init.restype = C.c_uint32
    L.orc_scramb_get_init.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8]
    L.orc_scramb_get_bits.argtypes = [C.c_uint32, u8p, C.c_int]
    L.orc_scramb_bits.argtypes = [C.c_uint32, u8p, C.c_int]
    L.orc_block_interleave.argtypes = [C.c_uint32, C.c_uint32, u8p, u8p]
    L.orc_block_deinterleave.argtypes = [C.c_uint32, C.c_uint32, u8p, u8p]
    L.orc_conv_encode.argtypes = [u8p, C.c_int, u8p]
    L.orc_puncture.argtypes = [C.c_int, u8p, C.c_int, u8p]
    L.orc_depuncture.argtypes = [C.c_int, u8p, C.c_int, u8p]
    L.orc_gsmtap_makemsg.argtypes = [C.POINTER(TdmaTime), C.c_int, C.c_uint8, C.c_uint8, C.c_int8, C.c_uint8, u8p, C.c_uint, u8p]
    L.orc_conv_encode_tch.argtypes = [u8p, C.c_int, u8p]
    L.orc_conv_decode_block.argtypes = [C.c_int, C.c_int, u8p, C.c_uint, C.c_uint, C.c_int, u8p]
    L.orc_code_output.argtypes = [C.c_int, C.c_uint, C.c_uint]
    L.orc_code_output.restype = C.c_uint
    i8p = C.POINTER(C.c_int8)
    L.orc_viterbi_generic.argtypes = [i8p, u8p, C.c_int]
    L.orc_viterbi_acc.argtypes = [i8p, u8p, C.c_int]
    L.orc_viterbi_dec_wrapper.argtypes = [u8p, u8p, C.c_uint, C.c_int]
    L.orc_viterbi_soft.argtypes = [i8p, u8p, C.c_uint]
    L.orc_crc16_itut_bits.restype = C.c_uint16
    L.orc_crc16_itut_bits.argtypes = [C.c_uint16, u8p, C.c_int]
    L.orc_crc16_ccitt_bits.restype = C.c_uint16
    L.orc_crc16_ccitt_bits.argtypes = [u8p, C.c_uint]
    L.orc_rm3014_row.restype = C.c_uint32
    L.orc_rm3014_row.argtypes = [C.c_int]
    L.orc_rm3014_compute.restype = C.c_uint32
    L.orc_rm3014_compute.argtypes = [C.c_uint16]
    L.orc_tdma_add_tn.argtypes = [C.POINTER(TdmaTime), C.c_uint32]
    L.orc_find_train_seq.argtypes = [u8p, C.c_uint, C.c_uint32, C.POINTER(C.c_uint)]
    L.orc_train_bits.restype = u8p
    L.orc_train_bits.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    L.orc_build_sync_burst.argtypes = [u8p, u8p, u8p, u8p]
    L.orc_build_norm_burst.argtypes = [u8p, u8p, u8p, u8p, C.c_int]
    L.orc_encode_block.argtypes = [C.c_int, u8p, C.c_uint32, u8p]
    L.orc_encode_bbk.argtypes = [u8p, C.c_uint32, u8p]
    L.orc_decode_block.argtypes = [C.c_int, u8p, C.c_uint32, C.c_int, C.POINTER(BlockResult)]
    L.orc_rx_init.argtypes = [C.POINTER(Rx), UPPER_CB, EVENT_CB, C.c_void_p]
    L.orc_burst_sync_in.argtypes = [C.POINTER(Rx), u8p, C.c_uint]
    L.orc_rx_feed.argtypes = [C.POINTER(Rx), u8p, C.c_size_t, C.c_uint]
    L.orc_burst_rx_cb.argtypes = [C.POINTER(Rx), u8p, C.c_uint, C.c_int]
    L.orc_tp_sap_udata_ind.argtypes = [C.POINTER(Rx), C.c_int, C.c_int, u8p, C.c_uint]
    L.orc_float_to_bits.argtypes = [C.POINTER(C.c_float), C.c_size_t, u8p, C.c_int, C.c_float, C.c_float,
                                    C.POINTER(C.c_float)]
    L.orc_float_to_soft.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int8)]
    L.orc_decode_block_soft.argtypes = [C.c_int, C.POINTER(C.c_int8), C.c_uint32, C.POINTER(BlockResult)]
    L.orc_bench_decode_slots.restype = C.c_uint64
    L.orc_bench_decode_slots.argtypes = [u8p, u8p, C.c_size_t, C.c_uint32, C.c_int, u8p, C.POINTER(C.c_uint16)]
    _lib = L
    return L


# ---------------------------------------------------------------------------
# numpy-level helpers
# ---------------------------------------------------------------------------
def bits(s):
    """'0101' -> uint8 array"""
    return np.frombuffer(s.encode(), dtype=np.uint8) - ord("0")


def bitstr(a):
    return "".join(str(int(x)) for x in a)


def scramb_get_init(mcc, mnc, cc):
    return lib().orc_scramb_get_init(mcc, mnc, cc)


def scramb_seq(init, n):
    out = np.zeros(n, np.uint8)
    lib().orc_scramb_get_bits(init, _p(out), n)
    return out


def scramb(init, a):
    out = np.array(a, dtype=np.uint8, copy=True)
    lib().orc_scramb_bits(init, _p(out), len(out))
    return out


def interleave(K, a, x):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros(K, np.uint8)
    lib().orc_block_interleave(K, a, _p(x), _p(out))
    return out


def deinterleave(K, a, x):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros(K, np.uint8)
    lib().orc_block_deinterleave(K, a, _p(x), _p(out))
    return out


def conv_encode(x):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros(4 * len(x), np.uint8)
    lib().orc_conv_encode(_p(x), len(x), _p(out))
    return out


def puncture(pu, mother, n):
    mother = np.ascontiguousarray(mother, np.uint8)
    out = np.zeros(n, np.uint8)
    assert lib().orc_puncture(pu, _p(mother), n, _p(out)) == 0
    return out


def depuncture(pu, x, mother_len, fill=0xFF):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.full(mother_len, fill, np.uint8)
    assert lib().orc_depuncture(pu, _p(x), len(x), _p(out)) == 0
    return out


# lower_mac/tetra_conv_enc.c:257-267 (punct_test_params): (type2_len, type3_len, mother rate, puncturer)
PUNCT_SHAPES = [(80, 120, 4, 0), (292, 432, 4, 2), (148, 432, 4, 3), (144, 216, 4, 0), (112, 168, 4, 0),
                (288, 432, 4, 0), (112, 168, 3, 4), (72, 162, 3, 5), (38, 80, 3, 6)]


def gsmtap_makemsg(tm, lchan, ts, bits, ss=0, signal_dbm=0, snr=0):
    b = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(16 + (len(b) + 7) // 8, np.uint8)
    n = lib().orc_gsmtap_makemsg(C.byref(TdmaTime(*tm)), lchan, ts, ss, signal_dbm, snr, _p(b), len(b), _p(out))
    return out[:n].tobytes()


def conv_encode_tch(bits):
    x = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(3 * len(x), np.uint8)
    lib().orc_conv_encode_tch(_p(x), len(x), _p(out))
    return out


def conv_decode_block(pu, mother, type3, type2_len, use_acc=0):
    """depuncture + Viterbi for any puncturer on either mother code; None for an invalid shape"""
    x = np.ascontiguousarray(type3, np.uint8)
    out = np.zeros(type2_len, np.uint8)
    rc = lib().orc_conv_decode_block(pu, mother, _p(x), len(x), type2_len, use_acc, _p(out))
    return None if rc else out


def conv_encode_block(pu, mother, type2, type3_len):
    m = conv_encode_tch(type2) if mother == 3 else conv_encode(type2)
    return puncture(pu, m, type3_len)


def viterbi_hard(type3dp, n, use_acc=0):
    """type3dp: uint8 mother-code array with 0/1/0xff; returns n decoded bits"""
    x = np.ascontiguousarray(type3dp, np.uint8)
    out = np.zeros(n, np.uint8)
    lib().orc_viterbi_dec_wrapper(_p(x), _p(out), n, use_acc)
    return out


def viterbi_soft(sb, n):
    x = np.ascontiguousarray(sb, np.int8)
    out = np.zeros(n, np.uint8)
    lib().orc_viterbi_soft(x.ctypes.data_as(C.POINTER(C.c_int8)), _p(out), n)
    return out


def crc16(x):
    x = np.ascontiguousarray(x, np.uint8)
    return lib().orc_crc16_ccitt_bits(_p(x), len(x))


def encode_block(t, type1, scramb_init):
    type1 = np.ascontiguousarray(type1, np.uint8)
    assert len(type1) == BLK[t][2]
    out = np.zeros(BLK[t][0], np.uint8)
    lib().orc_encode_block(t, _p(type1), scramb_init, _p(out))
    return out


def encode_bbk(type1_14, scramb_init):
    type1_14 = np.ascontiguousarray(type1_14, np.uint8)
    out = np.zeros(30, np.uint8)
    lib().orc_encode_bbk(_p(type1_14), scramb_init, _p(out))
    return out


def decode_block(t, type5, scramb_init, use_acc=0):
    type5 = np.ascontiguousarray(type5, np.uint8)
    res = BlockResult()
    lib().orc_decode_block(t, _p(type5), scramb_init, use_acc, C.byref(res))
    n1 = BLK[t][2]
    return (np.frombuffer(res.type1, np.uint8, n1).copy(), res.crc, bool(res.crc_ok),
            np.frombuffer(res.type2, np.uint8, BLK[t][1]).copy())


def rm3014_decode_ml(rx30):
    L = lib()
    L.orc_rm3014_decode_ml.restype = C.c_uint16
    n = C.c_uint(0)
    d = L.orc_rm3014_decode_ml(C.c_uint32(int(rx30)), C.byref(n))
    return int(d), int(n.value)


def rm3014_compute(data14):
    L = lib()
    L.orc_rm3014_compute.restype = C.c_uint32
    return int(L.orc_rm3014_compute(C.c_uint16(int(data14))))


def traffic_block(type4):
    t = np.ascontiguousarray(type4, np.uint8)
    out = np.zeros(690, np.int16)
    lib().orc_traffic_block(_p(t), len(t), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def _acelp_args(cls):
    arrs = [np.ascontiguousarray(c, np.uint8) for c in cls]
    ptrs = (u8p * 3)(*[_p(a) for a in arrs])
    ns = (C.c_uint * 3)(*[len(a) for a in arrs])
    return arrs, ptrs, ns


def acelp_type2_to_codec(bits, cls, fill=0):
    """lower_mac/tch_reordering.c:94-117 with the class position tables cls = (class0, class1, class2)"""
    arrs, ptrs, ns = _acelp_args(cls)
    n = 2 * sum(len(a) for a in arrs)
    b = np.ascontiguousarray(bits, np.uint8)
    assert len(b) == n
    out = np.full(n, fill, np.uint8)
    lib().orc_acelp_type2_to_codec(_p(b), _p(out), ptrs, ns)
    return out


def acelp_codec_to_acelp(bits, cls, fill=0):
    arrs, ptrs, ns = _acelp_args(cls)
    n = 2 * sum(len(a) for a in arrs)
    b = np.ascontiguousarray(bits, np.uint8)
    assert len(b) == n
    out = np.full(n, fill, np.uint8)
    lib().orc_acelp_codec_to_acelp(_p(b), _p(out), ptrs, ns)
    return out


def ref_acelp_tables():
    """the class position tables of the REAL lower_mac/tch_reordering.c object, read off its behaviour (unit vectors
    through tetra_acelp_codec_to_acelp); None when oracle/_ref is not built.  Class sizes 51 / 56 / 30 are the
    #defines of tch_reordering.c:27,55,79."""
    R = ref()
    if R is None:
        return None
    n = 274
    # codec_to_acelp reads one position per table entry: label every position of frame 0 with its own number
    # (an entry 0 reads in[-1]: padded buffer, label 0 there) and read the table off the output
    buf = np.zeros(n + 16, np.uint8)
    buf[8:8 + 137] = np.arange(1, 138)
    out = np.zeros(n, np.uint8)
    R.tetra_acelp_codec_to_acelp(C.cast(buf[8:].ctypes.data, u8p), _p(out))
    pos = out[0::2]
    return [pos[:51].copy(), pos[51:107].copy(), pos[107:137].copy()]


def float_to_soft(phi):
    phi = np.ascontiguousarray(phi, np.float32)
    out = np.zeros(2 * len(phi), np.int8)
    lib().orc_float_to_soft(phi.ctypes.data_as(C.POINTER(C.c_float)), len(phi), out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out


def float_to_bits(phi, afc=False, fval=0.0001, goal=0.0):
    phi = np.ascontiguousarray(phi, np.float32)
    out = np.zeros(2 * len(phi), np.uint8)
    st = C.c_float(0)
    lib().orc_float_to_bits(phi.ctypes.data_as(C.POINTER(C.c_float)), len(phi), _p(out), int(afc), C.c_float(fval),
                            C.c_float(goal), C.byref(st))
    return out


def decode_block_soft(t, soft5, scramb_init):
    soft5 = np.ascontiguousarray(soft5, np.int8)
    res = BlockResult()
    lib().orc_decode_block_soft(t, soft5.ctypes.data_as(C.POINTER(C.c_int8)), scramb_init, C.byref(res))
    n1 = BLK[t][2]
    return (np.frombuffer(res.type1, np.uint8, n1).copy(), res.crc, bool(res.crc_ok),
            np.frombuffer(res.type2, np.uint8, BLK[t][1]).copy())


def bits_to_phase(bits):
    """inverse of sym_int2bits (float_to_bits.c:50-72): 00 -> +1, 01 -> +3, 10 -> -1, 11 -> -3"""
    b = np.asarray(bits, np.int64).reshape(-1, 2)
    return np.where(b[:, 0] == 0, 1.0, -1.0) * np.where(b[:, 1] == 0, 1.0, 3.0)


def build_sync_burst(sb, bb, bkn):
    buf = np.zeros(510, np.uint8)
    n = lib().orc_build_sync_burst(_p(buf), _p(np.ascontiguousarray(sb, np.uint8)),
                                   _p(np.ascontiguousarray(bb, np.uint8)), _p(np.ascontiguousarray(bkn, np.uint8)))
    assert n == 510
    return buf


def build_norm_burst(b1, bb, b2, two):
    buf = np.zeros(510, np.uint8)
    n = lib().orc_build_norm_burst(_p(buf), _p(np.ascontiguousarray(b1, np.uint8)),
                                   _p(np.ascontiguousarray(bb, np.uint8)), _p(np.ascontiguousarray(b2, np.uint8)), two)
    assert n == 510
    return buf


def find_train_seq(buf, end, mask):
    """buf must extend >= 21 bytes past 'end' (the reference reads cur[21])."""
    buf = np.ascontiguousarray(buf, np.uint8)
    assert len(buf) >= end + 22
    off = C.c_uint(0)
    rc = lib().orc_find_train_seq(_p(buf), end, mask, C.byref(off))
    return rc, off.value


def record_to_dict(r):
    n = r.type1_len
    return dict(burst_seq=r.burst_seq, burst_type=r.burst_type, type=r.type, blk_num=r.blk_num, lchan=r.lchan,
                crc_ok=r.crc_ok, traffic=r.traffic_dumped, crc=r.crc, scramb=r.scrambling_code,
                time=(r.time.tn, r.time.fn, r.time.mn), time_str=(r.time_str.tn, r.time_str.fn, r.time_str.mn),
                type1=bytes(r.type1[:n]),
                type4=bytes(r.type4[:BLK[r.type][0]]))


def run_rx(stream, chunk=64, use_acc=0, upper=None):
    """Feed a whole stream through the oracle receiver.

    upper(rx, recdict, offset) -> int emulates upper_mac_prim_recv(); default returns -1.
    Returns (records, events) where records has one dict per tp_sap_udata_ind() call."""
    L = lib()
    stream = np.ascontiguousarray(stream, np.uint8)
    recs, events = [], []

    def _upper(rxp, recp, offset, priv):
        d = record_to_dict(recp.contents)
        if offset in (0, 0xFFFFFFFF):
            recs.append(d)
        if offset == 0xFFFFFFFF:
            return -1
        if upper is not None:
            return int(upper(rxp.contents, d, offset))
        return -1

    def _event(ev, bitnum, arg, priv):
        events.append((ev, bitnum, arg))

    ucb, ecb = UPPER_CB(_upper), EVENT_CB(_event)
    rx = Rx()
    L.orc_rx_init(C.byref(rx), ucb, ecb, None)
    rx.use_acc = use_acc
    L.orc_rx_feed(C.byref(rx), _p(stream), len(stream), chunk)
    return recs, events


def bench_decode_slots(slots, types, scramb_init=0, use_acc=0, want_out=False, want_crc=False):
    slots = np.ascontiguousarray(slots, np.uint8)
    types = np.ascontiguousarray(types, np.uint8)
    n = len(types)
    out = np.zeros((n, 288), np.uint8) if want_out else None
    crc = np.zeros((n, 2), np.uint16) if want_out else None
    ok = lib().orc_bench_decode_slots(_p(slots), _p(types), n, scramb_init, use_acc, _p(out) if want_out else None,
                                      crc.ctypes.data_as(C.POINTER(C.c_uint16)) if want_out else None)
    if want_crc:
        return ok, out, crc
    return ok, out


def bench_decode_slots_soft(soft_slots, types, scramb_init=0):
    """(ok count, type-1 bits laid out as bench_decode_slots, crc words) of the oracle's soft chain"""
    s = np.ascontiguousarray(soft_slots, np.int8)
    types = np.ascontiguousarray(types, np.uint8)
    n = len(types)
    out = np.zeros((n, 288), np.uint8)
    crc = np.zeros((n, 2), np.uint16)
    L = lib()
    L.orc_bench_decode_slots_soft.restype = C.c_uint64
    ok = L.orc_bench_decode_slots_soft(s.ctypes.data_as(C.POINTER(C.c_int8)), _p(types), C.c_size_t(n), C.c_uint32(scramb_init),
                                       _p(out), crc.ctypes.data_as(C.POINTER(C.c_uint16)))
    return ok, out, crc


# ---------------------------------------------------------------------------
# the real reference objects (this container only)
# ---------------------------------------------------------------------------
class RefCall(C.Structure):
    _fields_ = [("type", C.c_int), ("blk_num", C.c_int), ("len", C.c_uint), ("bits", C.c_uint8 * 432)]


_ref = None


def ref():
    """libtetra_ref.so or None when it has not been built (e.g. no /root/reference)."""
    global _ref
    if _ref is not None:
        return _ref
    path = os.path.join(ORACLE_DIR, "_ref", "libtetra_ref.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.tetra_scramb_get_init.restype = C.c_uint32
    R.tetra_scramb_get_init.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8]
    R.tetra_scramb_get_bits.argtypes = [C.c_uint32, u8p, C.c_int]
    R.tetra_scramb_bits.argtypes = [C.c_uint32, u8p, C.c_int]
    R.crc16_ccitt_bits.restype = C.c_uint16
    R.crc16_ccitt_bits.argtypes = [u8p, C.c_uint]
    R.crc16_itut_bits.restype = C.c_uint16
    R.crc16_itut_bits.argtypes = [C.c_uint16, u8p, C.c_int]
    R.tetra_rm3014_compute.restype = C.c_uint32
    R.tetra_rm3014_compute.argtypes = [C.c_uint16]
    R.build_sync_c_d_burst.argtypes = [u8p, u8p, u8p, u8p]
    R.build_norm_c_d_burst.argtypes = [u8p, u8p, u8p, u8p, C.c_int]
    R.tetra_find_train_seq.argtypes = [u8p, C.c_uint, C.c_uint32, C.POINTER(C.c_uint)]
    R.tetra_burst_rx_cb.argtypes = [u8p, C.c_uint, C.c_int, C.c_void_p]
    R.tetra_tdma_time_add_tn.argtypes = [C.POINTER(TdmaTime), C.c_uint32]
    R.viterbi_dec_sb1_wrapper.argtypes = [u8p, u8p, C.c_uint]
    R.tetra_acelp_type2_to_codec.argtypes = [u8p, u8p]
    R.tetra_acelp_codec_to_acelp.argtypes = [u8p, u8p]
    R.ref_glue_set_decoder.argtypes = [C.c_void_p]
    R.ref_glue_vit_input.restype = C.POINTER(C.c_int8)
    R.ref_glue_get.restype = C.POINTER(RefCall)
    R.ref_glue_get.argtypes = [C.c_int]
    R.tetra_rm3014_init()
    _ref = R
    return R


def ref_float_to_bits():
    p = os.path.join(ORACLE_DIR, "_ref", "float_to_bits")
    return p if os.path.exists(p) else None
