"""Synthetic TETRA downlink streams for the tests, built with the ORACLE's TX side
(rows E of SURVEY.md section 8(a)).  Test infrastructure only."""
import numpy as np

import oraclelib as O


def splitmix64(seed):
    """deterministic 64-bit generator (SURVEY 8(d) config 2 names splitmix64(seed=1))"""
    x = np.uint64(seed)
    with np.errstate(over="ignore"):
        while True:
            x = x + np.uint64(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            yield int(z ^ (z >> np.uint64(31)))


def sync_pdu(cc, tn, fn, mn, mcc, mnc):
    """60 type-1 bits of a SYNC PDU; layout from testpdu.c:43-58 / tetra_lower_mac.c:284-297"""
    def f(v, n):
        return [(v >> (n - 1 - i)) & 1 for i in range(n)]
    b = f(0, 4) + f(cc, 6) + f(tn - 1, 2) + f(fn, 5) + f(mn, 6) + f(0, 2) + f(0, 3) + [0, 0, 0] + f(mcc, 10) + f(mnc, 14) + f(0, 5)
    assert len(b) == 60
    return np.array(b, np.uint8)


NULL_PDU_HDR = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0], np.uint8)  # MAC-RESOURCE, null address


def payload(rng, n, hdr=True):
    p = rng.integers(0, 2, n).astype(np.uint8)
    if hdr:
        p[:16] = NULL_PDU_HDR
    return p


class Cell:
    def __init__(self, mcc=262, mnc=42, cc=1):
        self.mcc, self.mnc, self.cc = mcc, mnc, cc
        self.code = O.scramb_get_init(mcc, mnc, cc)


def make_sb(rng, cell, tn, fn, mn, code=None, aach=None):
    code = cell.code if code is None else code
    sb1 = O.encode_block(O.T_SB1, sync_pdu(cell.cc, tn, fn, mn, cell.mcc, cell.mnc), 3)
    sb2p = payload(rng, 124, hdr=False)
    sb2p[:2] = (1, 0)  # BROADCAST
    sb2 = O.encode_block(O.T_SB2, sb2p, code)
    a = np.zeros(14, np.uint8) if aach is None else aach
    bb = O.encode_bbk(a, code)
    return O.build_sync_burst(sb1, bb, sb2)


def make_norm1(rng, code, aach=None):
    blk = O.encode_block(O.T_SCH_F, payload(rng, 268), code)
    a = np.zeros(14, np.uint8) if aach is None else aach
    return O.build_norm_burst(blk[:216], O.encode_bbk(a, code), blk[216:], 0)


def make_norm2(rng, code, aach=None):
    b1 = O.encode_block(O.T_NDB, payload(rng, 124), code)
    b2 = O.encode_block(O.T_NDB, payload(rng, 124), code)
    a = np.zeros(14, np.uint8) if aach is None else aach
    return O.build_norm_burst(b1, O.encode_bbk(a, code), b2, 1)


# bit ranges of a slot that carry coded payload (noise goes only here: training
# sequences are matched exactly, SURVEY 8(d) "Noise placement")
FIELDS = {
    O.TRAIN_SYNC: [(94, 214), (252, 282), (282, 498)],
    O.TRAIN_NORM_1: [(14, 230), (230, 244), (266, 282), (282, 498)],
    O.TRAIN_NORM_2: [(14, 230), (230, 244), (266, 282), (282, 498)],
}


def add_field_noise(rng, slot, btype, ber):
    if ber <= 0:
        return slot
    s = slot.copy()
    for (a, b) in FIELDS[btype]:
        flips = rng.random(b - a) < ber
        s[a:b] ^= flips.astype(np.uint8)
    return s


def frame_stream(seed=1, nframes=4, cell=None, ber=0.0, lead_in=100, pad=700, tn0=1, fn0=1, mn0=1):
    """lead-in, one lock-only SB, then repeating 8-slot frames
    [SB, N1, N2, N1, N2, N1, N2, N1]; returns (stream, list of (type, slot_bits))."""
    rng = np.random.default_rng(seed)
    cell = cell or Cell()
    parts = [rng.integers(0, 2, lead_in).astype(np.uint8)]
    slots = []
    tm = O.TdmaTime(0, 0, tn0, fn0, mn0)
    parts.append(make_sb(rng, cell, tn0, fn0, mn0))  # lock-only SB
    pattern = [O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_1,
               O.TRAIN_NORM_2, O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_1]
    for _ in range(nframes):
        for t in pattern:
            O.lib().orc_tdma_add_tn(tm, 1)
            if t == O.TRAIN_SYNC:
                s = make_sb(rng, cell, tm.tn, tm.fn, tm.mn)
            elif t == O.TRAIN_NORM_1:
                s = make_norm1(rng, cell.code)
            else:
                s = make_norm2(rng, cell.code)
            s = add_field_noise(rng, s, t, ber)
            slots.append((t, s))
            parts.append(s)
    parts.append(np.zeros(pad, np.uint8))
    return np.concatenate(parts), slots


def ndb_slots(n, seed=1, ber=0.0, code=0, mix=(0.5, 0.5)):
    """n aligned NDB slots, ~50 % NORM_1 / 50 % NORM_2, scramb_init = code."""
    rng = np.random.default_rng(seed)
    types = np.where(rng.random(n) < mix[0], O.TRAIN_NORM_1, O.TRAIN_NORM_2).astype(np.uint8)
    out = np.zeros((n, 510), np.uint8)
    for i in range(n):
        s = make_norm1(rng, code) if types[i] == O.TRAIN_NORM_1 else make_norm2(rng, code)
        out[i] = add_field_noise(rng, s, int(types[i]), ber)
    return out, types
