#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.json from the REAL reference objects.

Runs only in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  The output is data only: inputs and the outputs the
reference's own compiled functions gave for them.  Re-run:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oraclelib as O  # noqa: E402


def bs(a):
    return "".join(str(int(x)) for x in a)


def main():
    O.build_oracle()
    R = O.ref()
    assert R is not None, "oracle/_ref not built (no /root/reference?)"
    rng = np.random.default_rng(20260928)
    out = {}

    # --- scrambler: tetra_scramb_get_init / tetra_scramb_get_bits ---------
    inits = [0, 3, 0x41802A07, 0xFFFFFFFF, 0x80000000, 1]
    cells = [(262, 42, 1), (0, 0, 0), (1023, 16383, 63), (901, 9999, 17), (234, 14, 33)]
    sc = {"get_init": [], "seq432": []}
    for (mcc, mnc, cc) in cells:
        v = R.tetra_scramb_get_init(mcc, mnc, cc)
        sc["get_init"].append([mcc, mnc, cc, v])
        inits.append(v)
    for init in inits + [int(x) for x in rng.integers(0, 2**32, 6)]:
        buf = np.zeros(432, np.uint8)
        R.tetra_scramb_get_bits(init, O._p(buf), 432)
        sc["seq432"].append([init, bs(buf)])
    out["scramb"] = sc

    # --- CRC-16 -----------------------------------------------------------
    crc = []
    for n in (1, 16, 60, 76, 124, 140, 268, 284):
        for _ in range(3):
            b = rng.integers(0, 2, n).astype(np.uint8)
            crc.append([bs(b), R.crc16_ccitt_bits(O._p(b), n)])
    z = np.zeros(76, np.uint8)
    crc.append([bs(z), R.crc16_ccitt_bits(O._p(z), 76)])
    out["crc16_ccitt"] = crc

    # --- RM(30,14) --------------------------------------------------------
    rm = []
    for v in [0, 1, 0x1001, 0x3FFF, 0x2000, 0x1555, 0x2AAA] + [int(x) for x in rng.integers(0, 1 << 14, 9)]:
        rm.append([v, R.tetra_rm3014_compute(v)])
    out["rm3014"] = rm

    # --- TDMA time: sequences of add_tn(1) from odd starting points --------
    tdma = []
    for (tn, fn, mn) in [(0, 0, 0), (1, 1, 1), (4, 18, 60), (3, 17, 59), (4, 25, 61), (7, 31, 63), (2, 0, 0)]:
        t = O.TdmaTime(0, 0, tn, fn, mn)
        seq = []
        for _ in range(12):
            R.tetra_tdma_time_add_tn(C.byref(t), 1)
            seq.append([t.tn, t.fn, t.mn])
        tdma.append([[tn, fn, mn], seq])
    out["tdma_add_tn"] = tdma

    # --- training sequence search incl. the offset<21 blind spot ----------
    fts = []
    ylen = C.c_uint()
    seqs = {t: np.ctypeslib.as_array(O.lib().orc_train_bits(t, C.byref(ylen)), (ylen.value,)).copy()
            for t in range(5)}
    masks = [1 << 3, (1 << 0) | (1 << 1) | (1 << 3), 0x1F]
    for trial in range(60):
        end = int(rng.integers(60, 700))
        buf = rng.integers(0, 2, end + 64).astype(np.uint8)
        if trial % 3:
            t = int(rng.integers(0, 5))
            pos = int(rng.integers(0, max(1, end - 10)))
            if trial % 6 == 1:
                pos = int(rng.integers(0, 24))
            s = seqs[t]
            buf[pos:pos + len(s)] = s[: len(buf) - pos]
        mask = masks[trial % 3]
        off = C.c_uint(0)
        rc = R.tetra_find_train_seq(O._p(buf), end, mask, C.byref(off))
        fts.append([bs(buf), end, mask, rc, off.value if rc >= 0 else 0])
    # glitch-window cases: p-sequence planted so that the skewed prefilter window can also match
    for pos in range(0, 23):
        end = 120
        buf = np.zeros(end + 64, np.uint8)
        buf[pos:pos + 22] = seqs[1]
        off = C.c_uint(0)
        rc = R.tetra_find_train_seq(O._p(buf), end, 0x1F, C.byref(off))
        fts.append([bs(buf), end, 0x1F, rc, off.value if rc >= 0 else 0])
    out["find_train_seq"] = fts

    # --- burst demux: tetra_burst_rx_cb -----------------------------------
    demux = []
    for t in (O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_3):
        burst = rng.integers(0, 2, 510).astype(np.uint8)
        R.ref_glue_reset()
        R.tetra_burst_rx_cb(O._p(burst), 510, t, None)
        calls = []
        for i in range(R.ref_glue_count()):
            c = R.ref_glue_get(i).contents
            calls.append([c.type, c.blk_num, c.len, bs(c.bits[: c.len])])
        demux.append([t, bs(burst), calls])
    out["burst_rx_cb"] = demux

    # --- burst builders (phase-adjustment bits 12,13,498,499 masked: the
    #     reference indexes its phase table out of range there) -------------
    builds = []
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the builders printf()
    try:
        for _ in range(3):
            sb = rng.integers(0, 2, 120).astype(np.uint8)
            bb = rng.integers(0, 2, 30).astype(np.uint8)
            b1 = rng.integers(0, 2, 216).astype(np.uint8)
            b2 = rng.integers(0, 2, 216).astype(np.uint8)
            buf = np.zeros(510, np.uint8)
            R.build_sync_c_d_burst(O._p(buf), O._p(sb), O._p(bb), O._p(b2))
            m = buf.copy(); m[[12, 13, 498, 499]] = 0
            builds.append(["sync", bs(sb), bs(bb), bs(b2), 0, bs(m)])
            for two in (0, 1):
                R.build_norm_c_d_burst(O._p(buf), O._p(b1), O._p(bb), O._p(b2), two)
                m = buf.copy(); m[[12, 13, 498, 499]] = 0
                builds.append(["norm", bs(b1), bs(bb), bs(b2), two, bs(m)])
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(devnull)
    out["build_burst"] = builds

    # --- viterbi.c: the ubit -> sbit map in front of the decoder (recorded by ref_glue.c's conv_cch_decode) ---
    vw = []
    for n in (80, 144, 288, 7):
        u = rng.choice([0, 1, 0xff, 2, 0x80, 0x7f], 4 * n, p=[.4, .4, .1, .04, .03, .03]).astype(np.uint8)
        out_bits = np.zeros(n + 8, np.uint8)
        R.ref_glue_set_decoder(None)
        R.viterbi_dec_sb1_wrapper(O._p(u), O._p(out_bits), n)
        assert R.ref_glue_vit_n() == n
        rec = np.ctypeslib.as_array(R.ref_glue_vit_input(), ((n + 4) * 4,)).copy()
        vw.append([n, u.tobytes().hex(), rec.astype(np.int8).tobytes().hex()])
    out["viterbi_wrapper"] = vw

    # --- tch_reordering.c: in/out pairs of both directions ----------------
    ac = {"type2_to_codec": [], "codec_to_acelp": []}
    for _ in range(6):     # (padded buffers: a table entry 0 makes the reference index [-1])
        b = np.zeros(274 + 16, np.uint8)
        b[8:282] = rng.integers(0, 2, 274)
        o = np.full(274 + 16, 7, np.uint8)
        R.tetra_acelp_type2_to_codec(C.cast(b[8:].ctypes.data, O.u8p), C.cast(o[8:].ctypes.data, O.u8p))
        ac["type2_to_codec"].append([bs(b[8:282]), bs(o[8:282])])
        o = np.full(274 + 16, 7, np.uint8)
        b[7] = 7           # what codec_to_acelp reads at in[-1] stays recognisable
        R.tetra_acelp_codec_to_acelp(C.cast(b[8:].ctypes.data, O.u8p), C.cast(o[8:].ctypes.data, O.u8p))
        ac["codec_to_acelp"].append([bs(b[8:282]), bs(o[8:282])])
    out["acelp_reorder"] = ac

    # --- float_to_bits binary ----------------------------------------------
    f2b = O.ref_float_to_bits()
    fl = []
    base = np.array([0.0, 2.0, -2.0, 2.0000002, -2.0000002, 1e-30, -1e-30, 3.0, -3.0, 1.0, -1.0, 4.99, 5.0, -5.0,
                     7.5, -9.0, np.nan, np.inf, -np.inf], np.float32)
    noisy = (rng.choice([-3, -1, 1, 3], 300) + rng.normal(0, 0.4, 300) + 0.35).astype(np.float32)
    sig = np.concatenate([base, noisy])
    for args in ([], ["-a"], ["-a", "-f", "0.01"], ["-a", "-f", "0.05", "-F", "0.2"]):
        with tempfile.TemporaryDirectory() as td:
            fi, fo = os.path.join(td, "in.f32"), os.path.join(td, "out.bits")
            sig.tofile(fi)
            subprocess.check_call([f2b] + args + [fi, fo], stdout=subprocess.DEVNULL)
            res = np.fromfile(fo, np.uint8)
        fl.append([args, bs(res)])
    out["float_to_bits"] = {"input_f32_hex": sig.tobytes().hex(), "runs": fl}

    path = os.path.join(HERE, "ref_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
