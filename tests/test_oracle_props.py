"""Properties the reference's own tests check, applied to the oracle,
plus the cross-check between the two restated libosmocore Viterbi algorithms."""
import ctypes as C

import numpy as np
import pytest

import oraclelib as O
import synth

# lower_mac/tetra_conv_enc.c:257-267 (punct_test_params)
PUNCT_CASES = [(80, 120, 4, 0), (292, 432, 4, 2), (148, 432, 4, 3), (144, 216, 4, 0), (112, 168, 4, 0),
               (288, 432, 4, 0), (112, 168, 3, 4), (72, 162, 3, 5), (38, 80, 3, 6)]


@pytest.mark.parametrize("t2,t3,rate,pu", PUNCT_CASES)
def test_punct_roundtrip(t2, t3, rate, pu):
    """tetra_punct_test(): every transmitted position lands back on its mother position, exactly t3 filled."""
    mlen = t2 * rate
    mother = (np.arange(mlen) % 255).astype(np.uint8)
    tx = O.puncture(pu, mother, t3)
    dp = O.depuncture(pu, tx, mlen)
    filled = dp != 0xFF
    # 0xff sentinel collides with nothing because the ramp stops at 254
    assert (dp[filled] == mother[filled]).all()
    assert filled.sum() == t3


@pytest.mark.parametrize("K,a", [(120, 11), (216, 101), (432, 103), (168, 13)])
def test_interleave_bijection(K, a):
    x = np.arange(K) % 251
    y = O.interleave(K, a, x.astype(np.uint8))
    assert sorted(y.tolist()) == sorted((x % 256).tolist())
    assert (O.deinterleave(K, a, y) == x).all()


@pytest.mark.parametrize("t", [O.T_SB1, O.T_NDB, O.T_SCH_F, O.T_SCH_HU])
def test_loopback_noise_free(t):
    """conv_enc_test.c:336-349: encode -> decode gives CRC OK (here also: the same bits back)."""
    rng = np.random.default_rng(11 + t)
    for i in range(25):
        code = [0, 3, 0x41802A07][i % 3]
        t1 = rng.integers(0, 2, O.BLK[t][2]).astype(np.uint8)
        t5 = O.encode_block(t, t1, code)
        for acc in (0, 1):
            d1, crc, ok, _ = O.decode_block(t, t5, code, acc)
            assert ok and crc == O.CRC_OK
            assert (d1 == t1).all()


def test_crc_test_vector():
    """crc_test.c:43-72: 60-bit SYNC-like vector + appended CRC -> residue 0x1d0f"""
    b = [0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 0, 0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
         1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 1, 0, 0, 1, 1, 0, 0, 1, 1]
    b = np.array(b, np.uint8)
    crc = (~O.crc16(b)) & 0xFFFF
    full = np.concatenate([b, [(crc >> (15 - i)) & 1 for i in range(16)]]).astype(np.uint8)
    assert O.crc16(full) == 0x1D0F
    # the vector's own checksum field in crc_test.c is 1101111011110001
    assert f"{crc:016b}" == "1101111011110001"


@pytest.mark.parametrize("ber", [0.01, 0.02, 0.05, 0.08, 0.2, 0.5])
def test_viterbi_generic_equals_acc(ber):
    """Both published libosmocore algorithms must give the same block, ties included, on hard input."""
    rng = np.random.default_rng(int(ber * 1000))
    ndiff_vs_tx = 0
    for n, k in ((80, 120), (144, 216), (288, 432)):
        for _ in range(120):
            t2 = np.concatenate([rng.integers(0, 2, n - 4), np.zeros(4)]).astype(np.uint8)
            t3 = O.puncture(0, O.conv_encode(t2), k)
            t3 ^= (rng.random(k) < ber).astype(np.uint8)
            dp = O.depuncture(0, t3, 4 * n)
            a = O.viterbi_hard(dp, n, 0)
            b = O.viterbi_hard(dp, n, 1)
            assert (a == b).all()
            ndiff_vs_tx += int((a != t2).any())
    if ber >= 0.05:
        assert ndiff_vs_tx > 0  # the noise was strong enough to matter


def test_viterbi_tie_rule_matters():
    """flipping the tie rule changes a large share of noisy blocks, so the
    property above really exercises ties.  Here: an opposite-rule decoder written in numpy."""
    rng = np.random.default_rng(3)

    def decode(dp, n, prefer_p1):
        INF = 10 ** 6
        pm = np.full(16, INF); pm[0] = 0
        hist = np.zeros((n + 4, 16), np.int64)
        for i in range(n + 4):
            sym = dp[4 * i:4 * i + 4] if i < n else np.full(4, 0xFF)
            new = np.full(16, INF)
            for t in range(16):
                b = t & 1
                if i >= n and b:
                    continue
                best, bp = INF, 0
                for p in ((t >> 1), (t >> 1) | 8):
                    d1, d2, d3, d4 = p & 1, (p >> 1) & 1, (p >> 2) & 1, (p >> 3) & 1
                    o = [(b + d1 + d4) & 1, (b + d2 + d3 + d4) & 1, (b + d1 + d2 + d4) & 1, (b + d1 + d3 + d4) & 1]
                    m = pm[p] + sum(1 for j in range(4) if sym[j] != 0xFF and sym[j] != o[j])
                    if m < best or (prefer_p1 and m == best):
                        best, bp = m, p
                new[t], hist[i, t] = best, bp
            pm = new
        s, out = 0, np.zeros(n, np.uint8)
        for i in range(n + 3, -1, -1):
            if i < n:
                out[i] = s & 1
            s = hist[i, s]
        return out

    changed = 0
    for _ in range(12):
        t2 = np.concatenate([rng.integers(0, 2, 76), np.zeros(4)]).astype(np.uint8)
        t3 = O.puncture(0, O.conv_encode(t2), 120)
        t3 ^= (rng.random(120) < 0.06).astype(np.uint8)
        dp = O.depuncture(0, t3, 320)
        ours = O.viterbi_hard(dp, 80, 0)
        assert (decode(dp, 80, False) == ours).all()      # the canonical rule, third independent statement
        changed += int((decode(dp, 80, True) != ours).any())
    assert changed > 0


@pytest.mark.parametrize("L,K,mother,pu", [(8, 12, 4, 0), (4, 6, 4, 0), (8, 24, 4, 1)])
def test_viterbi_is_maximum_likelihood_with_the_stated_tie_rule_exhaustively(L, K, mother, pu):
    """EVERY received word of a short block (2^K hard words; with erasures: every single position, halves, every other
    position, random patterns) through both restated libosmocore algorithms: the decoded sequence has the brute-force
    minimum distance, and among the sequences at that distance it is the one the stated tie rule names (ml_exhaustive.py:
    smallest when read from the last bit to the first).  4096 words x 21 erasure patterns for the 2/3-punctured block of 8
    bits; more than half of the words have ties."""
    import ml_exhaustive as ML
    rng = np.random.default_rng(L * 100 + K)
    xs, cb = ML.codebook(L, K, mother, pu)
    words = ((np.arange(1 << K)[:, None] >> np.arange(K)[None, :]) & 1).astype(np.uint8) if K <= 12 else \
        rng.integers(0, 2, (4096, K)).astype(np.uint8)
    tied = total = 0
    for pat in ML.erasure_patterns(K, rng):
        rx = words.copy()
        rx[:, pat] = 0xff
        rx = np.unique(rx, axis=0)
        want, dmin, nties = ML.ml_decode(xs, cb, rx)
        for i in range(len(rx)):
            for acc in (0, 1):
                got = O.conv_decode_block(pu, mother, rx[i], L, acc)
                assert (got == want[i]).all(), (pat.nonzero()[0].tolist(), rx[i].tolist(), acc, got.tolist(), want[i].tolist(), int(nties[i]))
        tied += int((nties > 1).sum())
        total += len(rx)
    assert tied > total // 4          # the tie rule decided a large share of these blocks


def test_stream_plumbing_config1():
    """BASELINE config 1 / SURVEY 8(d): 64 zero bytes + SB + SB + NDB(SCH/F) + SB + 700 zero bytes.
    The first SB only gives lock; SB#2 must decode to the golden SYNC PDU."""
    rng = np.random.default_rng(1)
    cell = synth.Cell(262, 42, 0)
    sb = lambda: synth.make_sb(rng, cell, 1, 1, 1)
    stream = np.concatenate([np.zeros(64, np.uint8), sb(), sb(), synth.make_norm1(rng, cell.code), sb(),
                             np.zeros(700, np.uint8)])
    recs, events = O.run_rx(stream)
    sb1 = [r for r in recs if r["type"] == O.T_SB1]
    assert len(sb1) == 2 and all(r["crc_ok"] for r in sb1)
    assert O.bitstr(np.frombuffer(sb1[0]["type1"], np.uint8)) == \
        "000000000000000010000010000000001000001100000000010101000000"
    assert [r["type"] for r in recs] == [O.T_SB1, O.T_BBK, O.T_SB2, O.T_BBK, O.T_SCH_F, O.T_SB1, O.T_BBK, O.T_SB2]
    assert all(r["crc_ok"] for r in recs)
    assert recs[1]["scramb"] == O.scramb_get_init(262, 42, 0)
    assert recs[0]["time"] == (1, 1, 1)
    assert events[0][0] == 1  # found SYNC first


def test_stream_relock_and_drop():
    """loss of lock: everything up to and including the next SB is skipped."""
    stream, slots = synth.frame_stream(seed=4, nframes=3)
    pos0 = 100 + 510                 # first decoded slot starts here
    bad = 3                          # corrupt the training sequence of slot 3 (a NORM_1)
    s = stream.copy()
    s[pos0 + 510 * bad + 244 + 5] ^= 1
    recs, events = O.run_rx(s)
    seqs = sorted({r["burst_seq"] for r in recs})
    evs = [e[0] for e in events]
    assert 5 in evs                   # "could not find successive burst training sequence"
    assert evs.count(1) == 2          # found SYNC twice (initial + re-lock)
    clean, _ = O.run_rx(stream)
    assert len(recs) < len(clean)
    # bursts before the corruption are identical
    n_before = sum(1 for r in clean if r["burst_seq"] <= bad)
    assert recs[:n_before] == clean[:n_before]
