#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors_osmo.json from the reference's own sources built against a REAL libosmocore.

Run by tools/pin_with_libosmocore.sh on a machine that has libosmocore (this repository's build container has not: the
file this script writes does not exist here, and tests/test_oracle_golden.py::test_osmo_* are skipped until it does).
Data only: inputs, and what the reference's compiled functions / its tetra-rx program gave for them.

    python tests/golden/make_golden_osmo.py oracle/_ref_osmo        (holds libtetra_ref_osmo.so and tetra-rx)
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
import oraclelib as O  # noqa: E402   (stream / block GENERATORS only: what is recorded is the reference's answer)
import synth  # noqa: E402

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)


def p(a):
    return a.ctypes.data_as(u8p)


def hx(a):
    return np.ascontiguousarray(a, np.uint8).tobytes().hex()


def main():
    d = sys.argv[1]
    R = C.CDLL(os.path.join(d, "libtetra_ref_osmo.so"))
    rng = np.random.default_rng(20260929)
    out = {"versions": open(os.path.join(d, "versions.txt")).read().split("\n")[:2]}

    # ---- row I: block (de)interleaver, lower_mac/tetra_interleave.c:36-59, for the (K, a) of every block type ----
    il = []
    for K, a in ((120, 11), (216, 101), (432, 103), (168, 13)):
        src = (np.arange(K) % 251).astype(np.uint8)
        fwd, back = np.zeros(K, np.uint8), np.zeros(K, np.uint8)
        R.block_interleave(K, a, p(src), p(fwd))
        R.block_deinterleave(K, a, p(src), p(back))
        il.append([K, a, fwd.tolist(), back.tolist()])
    out["interleave"] = il

    # ---- row U: the seven puncturers on the nine shapes of tetra_conv_enc.c:257-267 ----
    pu = []
    for t2, t3, rate, punct in O.PUNCT_SHAPES:
        mother = (np.arange(t2 * 4) % 199 + 1).astype(np.uint8)
        got = np.zeros(t3, np.uint8)
        rc1 = R.get_punctured_rate(punct, p(mother), t3, p(got))
        marks = (np.arange(t3) % 250 + 1).astype(np.uint8)
        dp = np.full(t2 * 4, 0xFF, np.uint8)
        rc2 = R.tetra_rcpc_depunct(punct, p(marks), t3, p(dp))
        pu.append([t2, t3, rate, punct, int(rc1), got.tolist(), int(rc2), dp.tolist()])
    out["puncture"] = pu

    # ---- the rate-1/4 encoder (conv_enc_input, tetra_conv_enc.c:78-88: g1 g2 g3 g4 per input bit, one byte each) ----
    enc = []
    for n in (80, 112, 144, 288):
        x = rng.integers(0, 2, n).astype(np.uint8)
        ces = (C.c_uint8 * 4)()
        R.conv_enc_init(ces)
        y = np.zeros(4 * n, np.uint8)
        R.conv_enc_input(ces, p(x), n, p(y))
        enc.append([hx(x), hx(y)])
    out["conv_enc"] = enc

    # ---- row V: osmo_conv_decode() through the reference's own wrapper on NOISY blocks of every block type ----
    # (lower_mac/viterbi.c:6-25 maps 0 -> +127, 0xff -> 0 (erased), else -> -127 and calls conv_cch_decode(), viterbi_cch.c:58-66)
    vit = []
    for t2len, t3len in ((80, 120), (144, 216), (112, 168), (288, 432)):
        for ber in (0.0, 0.02, 0.05, 0.08, 0.15):
            for rep in range(6):
                x = np.concatenate([rng.integers(0, 2, t2len - 4), np.zeros(4)]).astype(np.uint8)
                ces = (C.c_uint8 * 4)()
                R.conv_enc_init(ces)
                mother = np.zeros(4 * t2len, np.uint8)
                R.conv_enc_input(ces, p(x), t2len, p(mother))    # (as conv_enc_test.c:119-122 does)
                t3 = np.zeros(t3len, np.uint8)
                assert R.get_punctured_rate(0, p(mother), t3len, p(t3)) == 0
                t3 ^= (rng.random(t3len) < ber).astype(np.uint8)
                dp = np.full(4 * t2len, 0xFF, np.uint8)
                assert R.tetra_rcpc_depunct(0, p(t3), t3len, p(dp)) == 0
                if rep >= 3:                                     # received positions erased as well
                    er = np.flatnonzero(dp != 0xFF)
                    dp[rng.choice(er, max(1, len(er) // 30), replace=False)] = 0xFF
                dec = np.zeros(t2len, np.uint8)
                R.viterbi_dec_sb1_wrapper(p(dp), p(dec), t2len)
                vit.append([t2len, hx(dp), hx(dec), float(ber)])
    out["viterbi_cch"] = vit

    # the two decoders on raw int8 input (soft values included: what a soft front end would hand over)
    raw = []
    R.conv_cch_decode.argtypes = [i8p, u8p, C.c_int]
    R.conv_tch_decode.argtypes = [i8p, u8p, C.c_int]
    for code, fn, n in ((0, R.conv_cch_decode, 144), (0, R.conv_cch_decode, 288), (1, R.conv_tch_decode, 112), (1, R.conv_tch_decode, 72)):
        for rep in range(8):
            sb = rng.integers(-127, 128, 4 * (n + 4)).astype(np.int8)
            if rep < 4:
                sb = np.where(rng.random(len(sb)) < 0.3, 0, np.sign(sb).astype(np.int8) * 127).astype(np.int8)
            dec = np.zeros(n, np.uint8)
            rc = fn(sb.ctypes.data_as(i8p), p(dec), n)
            raw.append([code, n, sb.tobytes().hex(), hx(dec), int(rc)])
    out["osmo_conv_decode_raw"] = raw

    # ---- rows S and L: the reference's receiver on the stream list of SURVEY.md 8(c) ----
    rx = os.path.join(d, "tetra-rx")
    streams = []
    cases = [("sb_plumbing", dict(seed=1, nframes=1, ber=0.0)), ("mixed", dict(seed=2, nframes=6, ber=0.0)),
             ("noise_2pct", dict(seed=3, nframes=10, ber=0.02)), ("noise_5pct", dict(seed=4, nframes=10, ber=0.05)),
             ("relock", dict(seed=5, nframes=8, ber=0.01))]
    for name, kw in cases:
        s, _ = synth.frame_stream(**kw)
        s = s.copy()
        if name == "relock":
            s[100 + 510 + 510 * 9 + 244 + 7] ^= 1
            s[100 + 510 + 510 * 30 + 244 + 3] ^= 1
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "cap.bits")
            s.tofile(f)
            r = subprocess.run([rx, "-d", td, f], capture_output=True, text=True, timeout=300)
        keep = [ln for ln in r.stdout.splitlines() if ln.startswith(("CRC COMP", "SB1 ", "SB2 ", "NDB ", "SCH/F ", "SCH/HU ", "found SYNC", "TMB-SAP SYNC"))]
        err = [ln for ln in r.stderr.splitlines() if ln.startswith("####")]
        streams.append(dict(name=name, bits=np.packbits(s).tobytes().hex(), nbits=int(len(s)), rc=r.returncode, stdout=keep, stderr=err))
    out["tetra_rx"] = streams

    # ---- row P's uplink shape / row L on single blocks, and (f)3's GSMTAP bytes: tools/pin_harness.c in front of the reference's archives ----
    hx_bits = lambda a: np.ascontiguousarray(a, np.uint8).tobytes().hex()
    ph = subprocess.Popen([os.path.join(d, "pin_harness")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def ask(cmd):
        ph.stdin.write(cmd + "\n")
        ph.stdin.flush()
        lines = []
        while True:
            ln = ph.stdout.readline().rstrip("\n")
            if ln in ("done", "error", "") or ln.startswith("gsmtap "):
                lines.append(ln)
                return lines
            lines.append(ln)

    udata = []
    cell = (262, 42, 1)
    code = O.scramb_get_init(*cell)
    sy = synth.sync_pdu(cell[2], 2, 5, 7, cell[0], cell[1])       # (cc, tn, fn, mn, mcc, mnc): the SYNC PDU's 60 type-1 bits
    for ber in (0.0, 0.03):
        ask("time 2 5 7")
        seq = []
        if sy is not None:      # an SB1 first: it sets the cell's scrambling code for the blocks behind it (tetra_lower_mac.c:291-300)
            seq.append((0, 0, O.encode_block(O.T_SB1, sy, 3)))
        for t, tp in ((O.T_SCH_HU, 4), (O.T_NDB, 2), (O.T_SCH_F, 5), (O.T_SB2, 1)):
            t1 = rng.integers(0, 2, O.BLK[t][2]).astype(np.uint8)
            t1[:16] = synth.NULL_PDU_HDR      # MAC-RESOURCE null PDU header
            blk = O.encode_block(t, t1, code if sy is not None else 0)
            blk ^= (rng.random(len(blk)) < ber).astype(np.uint8)
            seq.append((tp, 1, blk))
        for tp, blk_num, blk in seq:
            out = ask("udata %d %d %s" % (tp, blk_num, hx_bits(blk)))
            udata.append(dict(type=tp, blk_num=blk_num, bits=hx_bits(blk), ber=ber, prims=[ln for ln in out if ln.startswith("prim ")]))
    out["lower_mac_udata"] = udata
    gs = []
    for (tn, fn, mn, lchan, ts, ss) in ((1, 1, 1, 1, 0, 0), (2, 18, 60, 10, 1, 0), (4, 7, 33, 8, 3, 1), (3, 9, 12, 3, 2, 2), (1, 18, 4, 11, 0, 0)):
        for nbits in (14, 60, 124, 268):
            b = rng.integers(0, 2, nbits).astype(np.uint8)
            r = ask("gsmtap %d %d %d %d %d %d %d %d %s" % (tn, fn, mn, lchan, ts, ss, -47, 12, hx_bits(b)))
            gs.append(dict(tm=[tn, fn, mn], lchan=lchan, ts=ts, ss=ss, signal_dbm=-47, snr=12, bits=hx_bits(b), msg=r[-1].split(" ", 1)[1] if r and r[-1].startswith("gsmtap ") else ""))
    out["gsmtap_makemsg"] = gs
    ph.stdin.close()
    ph.wait(timeout=10)

    path = os.path.join(HERE, "ref_vectors_osmo.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, {k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
