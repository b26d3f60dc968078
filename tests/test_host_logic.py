"""CPU-side checks of the PRODUCT: the C-ABI library loads and exports what include/tetra_gpu.h
declares, its host-side functions agree with the oracle / golden vectors, and the per-lane
trellis code the HIP kernels are built from (compiled here for the host) is bit-exact.
No GPU compute is called."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import emul
import oraclelib as O
import synth

import osmo_tetra_amd as T

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def _built():
    T.build_library()


def test_library_exports_header_symbols():
    L = T.lib()
    names = T.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(T.TgpuError):
        T.Engine(0)


def test_find_train_seq_golden_and_random():
    refv = json.load(open(os.path.join(G, "ref_vectors.json")))
    for b, end, mask, rc, off in refv["find_train_seq"]:
        got = T.find_train_seq(O.bits(b), end, mask)
        assert got[0] == rc and (rc < 0 or got[1] == off)
    rng = np.random.default_rng(9)
    # low-entropy streams make prefilter/blind-spot hits likely
    for _ in range(300):
        end = int(rng.integers(30, 640))
        buf = (rng.random(end + 40) < rng.choice([0.5, 0.2, 0.8])).astype(np.uint8)
        mask = int(rng.choice([8, 11, 31]))
        assert T.find_train_seq(buf, end, mask) == O.find_train_seq(buf, end, mask)


def test_tdma_and_scramb_init():
    L = T.lib()
    rng = np.random.default_rng(2)
    for _ in range(200):
        tn, fn, mn = (int(x) for x in rng.integers(0, 70, 3))
        a, b = T.binding.TdmaTime(0, 0, tn, fn, mn), O.TdmaTime(0, 0, tn, fn, mn)
        for _ in range(8):
            L.tetra_tdma_time_add_tn(C.byref(a), 1)
            O.lib().orc_tdma_add_tn(C.byref(b), 1)
            assert (a.tn, a.fn, a.mn) == (b.tn, b.fn, b.mn)
        mcc, mnc, cc = int(rng.integers(0, 2000)), int(rng.integers(0, 40000)), int(rng.integers(0, 256))
        assert L.tetra_scramb_get_init(mcc, mnc, cc) == O.scramb_get_init(mcc & 0xFFFF, mnc & 0xFFFF, cc)


def test_synth_decodes_in_oracle():
    """product TX -> oracle RX: every block CRC OK and equal to the generator's type-1 bits"""
    types = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2] * 20, np.uint8)
    code = O.scramb_get_init(262, 42, 1)
    slots, t1 = T.synth_slots(types, seed=5, scramb_init=code, want_type1=True)
    ok, out = O.bench_decode_slots(slots, types, code, want_out=True)
    assert ok == 40 * 2 + 20
    assert (out[:, :14 + 268] == t1[:, :14 + 268]).all()
    # training sequences sit where the synchroniser expects them
    for s, t in zip(slots, types):
        rc, off = O.find_train_seq(np.concatenate([s, np.zeros(40, np.uint8)]), 510, 0b1011)
        assert (rc, off) == (t, 214 if t == O.TRAIN_SYNC else 244)
    # SYNC PDU carries the cell identity
    r = O.decode_block(O.T_SB1, slots[0][94:214], 3)
    assert r[2] and int("".join(map(str, r[0][31:41])), 2) == 262


@pytest.mark.parametrize("ber", [0.0, 0.02, 0.05, 0.12, 0.5])
def test_kernel_core_on_host_bit_exact(ber):
    """the packed-u16 trellis (vit_core.h) == oracle, ties included, on all four block kinds"""
    rng = np.random.default_rng(int(ber * 100) + 1)
    for kind, t in ((0, O.T_SB1), (1, O.T_NDB), (2, O.T_SCH_F), (3, O.T_SCH_HU)):
        K, n2, n1, a = O.BLK[t]
        for _ in range(150):
            t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), 0)
            t5 ^= (rng.random(K) < ber).astype(np.uint8)
            got, crc = emul.decode_block(kind, t5)
            want1, wcrc, ok, want2 = O.decode_block(t, t5, 0)
            assert (got[:n2] == want2).all()
            assert crc == wcrc


def test_kernel_core_worst_case_metrics():
    """all-mismatch inputs drive the 8-bit path metrics as high as they can get"""
    for kind, t in ((0, O.T_SB1), (1, O.T_NDB), (2, O.T_SCH_F), (3, O.T_SCH_HU)):
        K, n2, n1, a = O.BLK[t]
        for pattern in (np.ones(K, np.uint8), np.zeros(K, np.uint8), (np.arange(K) % 2).astype(np.uint8),
                        (np.arange(K) % 3 == 0).astype(np.uint8)):
            got, crc = emul.decode_block(kind, pattern)
            _, wcrc, _, want2 = O.decode_block(t, pattern, 0)
            assert (got[:n2] == want2).all() and crc == wcrc


def test_difference_form_never_meets_a_metric_below_its_floor():
    """vit_core.h: the two-bit step of the difference form adds n - 2m >= -2 to a predecessor's metric in UNSIGNED packed
    arithmetic, so every metric it meets must sit at or above TG_VIT_FLOOR (the start state starts there, the SCH/F trellis' one
    normalisation returns the minimum to it).  The host build counts violations at the entry of every such step: none on clean,
    noisy, pure-noise and constant blocks of every kind (a kernel that normalised to 0 instead would show up here)."""
    L = emul.lib()
    L.emul_floor_violations.restype = emul.C.c_ulong
    L.emul_floor_violations()
    rng = np.random.default_rng(77)
    for kind, t in ((0, O.T_SB1), (1, O.T_NDB), (2, O.T_SCH_F), (3, O.T_SCH_HU)):
        K, n2, n1, a = O.BLK[t]
        blocks = [np.ones(K, np.uint8), np.zeros(K, np.uint8), (np.arange(K) % 2).astype(np.uint8)]
        for ber in (0.0, 0.05, 0.5):
            for _ in range(40):
                t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), 0)
                blocks.append(t5 ^ (rng.random(K) < ber).astype(np.uint8))
        for b in blocks:
            emul.decode_block(kind, b)
    assert L.emul_floor_violations() == 0


def test_slot_packing_matches_block_packing():
    """front-kernel gather table == demux (phy/tetra_burst.c:341-379) + per-block layout"""
    rng = np.random.default_rng(4)
    for bt in (O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2):
        slot = rng.integers(0, 2, 510).astype(np.uint8)
        w = emul.pack_slot(bt, slot)
        bbk = np.concatenate([slot[230:244], slot[266:282]]) if bt != O.TRAIN_SYNC else slot[252:282]
        assert w[18] == sum(int(b) << i for i, b in enumerate(bbk))
        parts = {O.TRAIN_SYNC: [(0, 0, slot[94:214]), (1, 9, slot[282:498])],
                 O.TRAIN_NORM_2: [(1, 0, slot[14:230]), (1, 9, slot[282:498])],
                 O.TRAIN_NORM_1: [(2, 0, np.concatenate([slot[14:230], slot[282:498]]))]}[bt]
        for kind, wbase, blk in parts:
            ww = np.zeros(18, np.uint32)
            emul.lib().emul_pack_block(kind, np.ascontiguousarray(blk).ctypes.data_as(emul.u8p),
                                       ww.ctypes.data_as(emul.u32p))
            nw = {0: 5, 1: 9, 2: 18}[kind]
            assert (w[wbase:wbase + nw] == ww[:nw]).all()


@pytest.mark.parametrize("sigma", [0.0, 0.6, 1.2, 4.0])
def test_soft_kernel_core_on_host_bit_exact(sigma):
    """the soft trellis (tg_svit_*, config 5) == the oracle's accelerated-decoder restatement on soft values,
    incl. pure noise and erasures (ties everywhere)"""
    rng = np.random.default_rng(int(sigma * 10) + 2)
    for kind, t in ((0, O.T_SB1), (1, O.T_NDB), (2, O.T_SCH_F)):
        K, n2, n1, a = O.BLK[t]
        for i in range(40):
            t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), 0)
            pad = np.concatenate([t5, np.zeros(K % 2, np.uint8)])
            soft = O.float_to_soft(O.bits_to_phase(pad) + rng.normal(0, sigma, len(pad) // 2))[:K]
            if i % 7 == 0:
                soft = rng.integers(-127, 128, K).astype(np.int8)
            if i % 11 == 0:
                soft[rng.random(K) < 0.3] = 0
            got, crc = emul.decode_block_soft(kind, soft)
            w1, wcrc, ok, w2 = O.decode_block_soft(t, soft, 0)
            assert (got[:n2] == w2).all() and crc == wcrc
            gotp, crcp = emul.decode_block_soft(kind, soft, packed=True)
            assert (gotp[:n2] == w2).all() and crcp == wcrc


def test_packed_soft_trellis_extremes_and_flips():
    """the packed 16-bit soft trellis the kernels run (tg_pvit_*): equal to the 32-bit statement under random sign
    flips (the scrambling mask words), exact on full-scale inputs (+-127, -128 everywhere, long constant runs, all
    zero), and its 12-bit metric never reaches 4096 under the kernel's normalisation schedule"""
    rng = np.random.default_rng(77)
    worst = 0
    for kind, t in ((0, O.T_SB1), (1, O.T_NDB), (2, O.T_SCH_F)):
        K, n2, n1, a = O.BLK[t]
        nw = {0: 5, 1: 9, 2: 18}[kind]
        cases = [np.full(K, -128, np.int8), np.full(K, 127, np.int8), np.zeros(K, np.int8),
                 np.where(rng.random(K) < 0.5, -128, 127).astype(np.int8),
                 np.where(np.arange(K) % 3 == 0, -128, 0).astype(np.int8)]
        for i in range(30):
            v = rng.integers(-128, 128, K).astype(np.int8)
            if i % 3 == 0:
                v = np.where(rng.random(K) < 0.5, -128, v).astype(np.int8)
            if i % 5 == 0:      # a valid code word at full scale with a few strong errors
                t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), 0)
                v = np.where(t5[:K] == 1, -128, 127).astype(np.int8)
                e = rng.random(K) < 0.08
                v[e] = -v[e].clip(-127, 127)
            cases.append(v)
        for soft in cases:
            w1, wcrc, ok, w2 = O.decode_block_soft(t, soft, 0)
            gotp, crcp, mx = emul.decode_block_soft(kind, soft, packed=True, want_max=True)
            assert (gotp[:n2] == w2).all() and crcp == wcrc
            worst = max(worst, mx)
            mw = rng.integers(0, 1 << 30, nw).astype(np.uint32)
            a32 = emul.decode_block_soft(kind, soft, mw)
            a16 = emul.decode_block_soft(kind, soft, mw, packed=True)
            assert (a32[0] == a16[0]).all() and a32[1] == a16[1]
    assert 700 < worst < 4096


def test_soft_definition_consistent_with_hard_slicer():
    """away from the decision boundaries the sign of the soft values is float_to_bits' hard decision"""
    rng = np.random.default_rng(1)
    phi = (rng.choice([-3, -1, 1, 3], 5000) + rng.normal(0, 0.8, 5000)).astype(np.float32)
    phi = phi[(np.abs(phi) > 0.02) & (np.abs(np.abs(phi) - 2) > 0.02)]
    assert ((O.float_to_soft(phi) < 0).astype(np.uint8) == O.float_to_bits(phi)).all()


def test_traffic_dump_block_format():
    """tgpu_traffic_block == the oracle's restatement of tetra_lower_mac.c:213-231; structure of the block
    (six 115-word frames, markers 0x6b21+i, +-127 soft bits, 432 bits used, tail zero)"""
    rng = np.random.default_rng(9)
    for n in (432, 216):
        t4 = rng.integers(0, 2, n).astype(np.uint8)
        got = T.traffic_block(t4)
        assert got.tolist() == O.traffic_block(t4).tolist()
        assert [int(got[115 * i]) for i in range(6)] == [0x6b21 + i for i in range(6)]
        body = np.concatenate([got[1:115], got[116:230], got[231:345], got[346:436]])
        full = np.concatenate([t4, np.zeros(432 - n, np.uint8)])
        assert (body == np.where(full == 1, -127, 127)).all()
        assert not got[436:460].any() and not got[461:575].any() and not got[576:].any()


def test_rm3014_decoder_host():
    """the optional (30,14) decoder (syndrome table) == exhaustive minimum-distance search incl. the tie rule;
    every pattern of up to 3 bit errors is corrected (d_min = 8)"""
    rng = np.random.default_rng(30)
    for _ in range(300):
        d = int(rng.integers(0, 1 << 14))
        cw = O.rm3014_compute(d)
        for nflip in (0, 1, 2, 3):
            e = 0
            for b in rng.choice(30, nflip, replace=False):
                e |= 1 << int(b)
            got, n = T.rm3014_decode(cw ^ e)
            assert got == d and n == nflip
    for _ in range(400):                       # arbitrary words, many of them beyond the guaranteed radius
        rx = int(rng.integers(0, 1 << 30))
        assert T.rm3014_decode(rx) == O.rm3014_decode_ml(rx)


def test_tch_code_tables():
    """the restated speech code has the reference's trellis (spot values of lower_mac/viterbi_tch.c:34-39) and
    the symmetry the butterfly form needs (every generator holds 1 and D^4)"""
    out = lambda s, b: O.lib().orc_code_output(1, s, b)
    assert [out(0, 0), out(0, 1), out(1, 0), out(1, 1), out(3, 0), out(8, 0), out(15, 1)] == [0, 7, 6, 1, 3, 7, 5]
    for code, full in ((0, 15), (1, 7)):
        for s in range(8):
            for b in (0, 1):
                o = O.lib().orc_code_output(code, s, b)
                assert O.lib().orc_code_output(code, s, b ^ 1) == o ^ full
                assert O.lib().orc_code_output(code, s + 8, b) == o ^ full


@pytest.mark.parametrize("shape", O.PUNCT_SHAPES)
def test_generic_trellis_on_host_bit_exact(shape):
    """k_conv's per-lane code (step programs of tg_conv.h + tg_step_gen) == depuncture + both restated
    libosmocore decoders, for every (puncturer, mother code) pair of tetra_conv_enc.c:257-267: clean, noisy,
    pure noise, erased (0xff) and non-binary input bytes"""
    L, K, mother, pu = shape
    rng = np.random.default_rng(L * 7 + pu)
    for i in range(60):
        t2 = rng.integers(0, 2, L).astype(np.uint8)
        t2[-4:] = 0 if i % 3 else t2[-4:]
        t3 = O.conv_encode_block(pu, mother, t2, K)
        ber = (0.0, 0.03, 0.08, 0.2, 0.5)[i % 5]
        t3 = t3 ^ (rng.random(K) < ber).astype(np.uint8)
        if i % 7 == 3:
            t3[rng.random(K) < 0.2] = 0xFF
        if i % 11 == 5:
            t3[t3 == 1] = rng.integers(1, 255, int((t3 == 1).sum())).astype(np.uint8)
        want = O.conv_decode_block(pu, mother, t3, L, 0)
        assert (O.conv_decode_block(pu, mother, t3, L, 1) == want).all()
        got = emul.conv_decode(pu, mother, t3, L)
        assert (got == want).all()
        if ber == 0.0 and i % 7 != 3:
            assert (want == t2).all() or t2[-4:].any()


def test_generic_trellis_rejects_what_the_reference_cannot_index():
    assert emul.conv_decode(7, 4, np.zeros(120, np.uint8), 80) is None          # unknown puncturer (-EINVAL)
    assert emul.conv_decode(0, 5, np.zeros(120, np.uint8), 80) is None
    assert emul.conv_decode(0, 4, np.zeros(432, np.uint8), 80) is None          # positions beyond the mother buffer
    assert O.conv_decode_block(0, 4, np.zeros(432, np.uint8), 80) is None
    assert emul.conv_decode(0, 4, np.zeros(15, np.uint8), 10) is None            # 10 % 8 = 2: unsupported history tail


def test_puncturer_entry_points_host():
    """get_punctured_rate / tetra_rcpc_depunct of the product (the reference's names) == the oracle's for all
    seven puncturers; -EINVAL beyond them (lower_mac/tetra_conv_enc.c:209-210,234-235)"""
    import errno
    for L, K, mother, pu in O.PUNCT_SHAPES:
        m = (np.arange(L * mother) % 251).astype(np.uint8)
        rc, tx = T.get_punctured_rate(pu, m, K)
        assert rc == 0 and tx.tolist() == O.puncture(pu, m, K).tolist()
        rc, dp = T.rcpc_depunct(pu, tx, L * mother)
        assert rc == 0 and dp.tolist() == O.depuncture(pu, tx, L * mother).tolist()
        assert int((dp != 0xFF).sum()) == K - int((tx == 0xFF).sum())
    assert T.get_punctured_rate(7, np.zeros(8, np.uint8), 2)[0] == -errno.EINVAL
    assert T.rcpc_depunct(9, np.zeros(8, np.uint8), 8)[0] == -errno.EINVAL


def test_gsmtap_message():
    """tgpu_gsmtap_makemsg == the oracle's restatement of tetra_gsmtap.c:31-63, plus the documented structure
    (GSMTAP v2 header of 16 bytes, type TETRA_I1, frame number in network order, MSB-first packed bits)"""
    rng = np.random.default_rng(3)
    for lchan in range(0, 13):
        for nbits in (0, 1, 7, 8, 14, 60, 124, 268):
            tm = (int(rng.integers(0, 4)), 0, int(rng.integers(1, 5)), int(rng.integers(1, 19)), int(rng.integers(1, 61)))
            bits = rng.integers(0, 2, nbits).astype(np.uint8)
            got = T.gsmtap_makemsg(tm, lchan, tm[2] - 1, bits, ss=int(rng.integers(0, 2)), signal_dbm=-int(rng.integers(0, 110)),
                                   snr=int(rng.integers(0, 40)))
            want = O.gsmtap_makemsg(tm, lchan, tm[2] - 1, bits, ss=got[14], signal_dbm=np.int8(got[6]).item() if got[6] < 128 else got[6] - 256,
                                    snr=got[7])
            assert got == want
            assert len(got) == 16 + (nbits + 7) // 8 and got[0] == 2 and got[1] == 4 and got[2] == 5 and got[3] == tm[2] - 1
            assert int.from_bytes(got[8:12], "big") == ((tm[0] * 60) + tm[4]) * 18 + tm[3]
            assert got[4:6] == b"\x00\x00" and got[13] == 0 and got[15] == 0
            assert np.unpackbits(np.frombuffer(got[16:], np.uint8))[:nbits].tolist() == bits.tolist()
    sub = {1: 5, 2: 4, 3: 3, 4: 7, 8: 2, 9: 8, 10: 1, 11: 6}
    for lchan in range(0, 13):
        assert T.gsmtap_makemsg((0, 0, 1, 1, 1), lchan, 0, [1])[12] == sub.get(lchan, 0)
    with pytest.raises(T.TgpuError):
        T.gsmtap_makemsg((0, 0, 1, 1, 1), 1, 0, np.ones(100, np.uint8), out_size=20)


def test_branch_metric_table_equals_arithmetic():
    """tg_vit_block_bm (table look-ups, what the kernels run) == tg_vit_block (the arithmetic it replaces):
    path metrics and history bytes after every block of random sequences"""
    for seed in range(1, 200):
        assert emul.lib().emul_bm_selfcheck(seed * 2654435761 % (1 << 32), 36) == 0


def _acelp_tables(seed):
    """synthetic class position tables (the real ones are EN 300 395-2 data and stay with the caller): a random split
    of 1..nbits into three classes, optionally damaged the way the reference's own table is (a position listed twice,
    an entry 0)"""
    rng = np.random.default_rng(seed)
    nbits = int(rng.integers(20, 138))
    perm = rng.permutation(nbits) + 1
    a, b = sorted(rng.choice(np.arange(1, nbits), 2, replace=False))
    cls = [perm[:a].astype(np.uint8), perm[a:b].astype(np.uint8), perm[b:].astype(np.uint8)]
    if seed & 1:
        cls[1][0] = cls[0][0]
        cls[0][-1] = 0
    return cls, nbits


@pytest.mark.parametrize("seed", range(6))
def test_acelp_reordering_host_entry_points(seed):
    """tetra_acelp_type2_to_codec() / tetra_acelp_codec_to_acelp() (the reference's names, caller-supplied tables) and
    the index maps behind them == the oracle's restatement of lower_mac/tch_reordering.c:94-140"""
    cls, nbits = _acelp_tables(seed)
    T.acelp_set_tables(cls)
    rng = np.random.default_rng(100 + seed)
    m1, m0 = T.acelp_build_map(cls, True), T.acelp_build_map(cls, False)
    for _ in range(4):
        b = rng.integers(0, 2, 2 * nbits).astype(np.uint8)
        want = O.acelp_type2_to_codec(b, cls, fill=7)
        got = T.acelp_type2_to_codec(b, out=np.full(2 * nbits, 7, np.uint8))
        assert got.tolist() == want.tolist()
        assert np.where(m1 >= 0, b[np.maximum(m1, 0)], 7).tolist() == want.tolist()
        want = O.acelp_codec_to_acelp(b, cls, fill=7)
        got = T.acelp_codec_to_acelp(b, out=np.full(2 * nbits, 7, np.uint8))
        assert got.tolist() == want.tolist()
        assert np.where(m0 >= 0, b[np.maximum(m0, 0)], 7).tolist() == want.tolist()
    if not seed & 1:   # a proper table: the two directions are inverse permutations
        assert sorted(m1.tolist()) == list(range(2 * nbits)) and (m0[m1] == np.arange(2 * nbits)).all()


def test_acelp_reordering_with_the_reference_tables():
    """with the class tables of the real lower_mac/tch_reordering.c object (read off its behaviour, build container
    and GPU box only) the product's entry points give the real functions' outputs"""
    cls = O.ref_acelp_tables()
    if cls is None:
        pytest.skip("oracle/_ref not built")
    import ctypes as C
    R = O.ref()
    T.acelp_set_tables(cls)
    rng = np.random.default_rng(9)
    for _ in range(5):
        b = np.zeros(274 + 16, np.uint8)
        b[8:282] = rng.integers(0, 2, 274)
        want = np.full(274 + 16, 7, np.uint8)
        R.tetra_acelp_type2_to_codec(C.cast(b[8:].ctypes.data, O.u8p), C.cast(want[8:].ctypes.data, O.u8p))
        got = T.acelp_type2_to_codec(b[8:282].copy(), out=np.full(274, 7, np.uint8))
        assert got.tolist() == want[8:282].tolist()


def test_pack_bits_host():
    """tgpu_pack_bits: bit i of packed byte k = bytes[8 k + i] & 1 for every length and thread count; a byte other than 0 / 1
    is reported"""
    import osmo_tetra_amd as T
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 8, 9, 127, 128, 129, 1000, 100003, 1_000_000):
        x = rng.integers(0, 2, n).astype(np.uint8)
        for th in (1, 3, 8):
            p, bad = T.pack_bits(x, nthreads=th)
            assert bad == 0 and (p == np.packbits(x, bitorder="little")).all(), (n, th)
    x = rng.integers(0, 2, 100000).astype(np.uint8)
    x[77777] = 3
    p, bad = T.pack_bits(x, nthreads=4)
    assert bad > 0 and (p == np.packbits(x & 1, bitorder="little")).all()


def test_adapter_uses_only_what_the_header_declares(tmp_path):
    """tools/tgpu_adapter.c is the file a maintainer of the reference adds; it needs libosmocore
    and the reference's headers, which are not here -- and nothing imitates them.  What CAN rot on this side is its use
    of include/tetra_gpu.h: every tgpu_* function it calls and every struct tgpu_unitdata field it reads is put into a
    probe that includes only tetra_gpu.h (the field must exist, the function must be declared and take that many
    arguments) and compiled with gcc -fsyntax-only.  Also: tetra_gpu.h behind the reference headers' include guards
    (the adapter includes those first) still declares the library's own API."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "tgpu_adapter.c")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    fields = sorted(set(re.findall(r"\bud->(\w+)", code)))
    calls = {}
    for m in re.finditer(r"\b(tgpu_(?!adapter_)\w+)\s*\(", code):
        depth, i, nargs, start = 1, m.end(), 0, m.end()
        while depth:
            c = code[i]
            depth += c == "("
            depth -= c == ")"
            nargs += (c == "," and depth == 1)
            i += 1
        calls[m.group(1)] = nargs + (1 if code[start:i - 1].strip() else 0)
    assert {"type1", "type1_len", "type4", "type4_len", "traffic", "lchan", "crc_ok", "scrambling_code", "blk_num", "tdma_time"} <= set(fields)
    assert {"tgpu_traffic_block", "tgpu_engine_create", "tgpu_channel_create", "tgpu_channel_bind_flags"} <= set(calls)
    probe = ["#include \"tetra_gpu.h\"", "static void probe(const struct tgpu_unitdata *ud)", "{"]
    probe += ["\t(void)sizeof(ud->%s);" % f for f in fields]
    probe += ["}"]
    for fn, n in sorted(calls.items()):       # a call with n null arguments type-checks iff the function is declared with n parameters
        probe += ["static void call_%s(void) { (void)%s(%s); }" % (fn, fn, ", ".join(["0"] * n))] if fn != "tgpu_traffic_block" and fn != "tgpu_channel_bind_flags" else \
                 ["static void call_%s(void) { %s(%s); }" % (fn, fn, ", ".join(["0"] * n))]
    # the callback has upper_mac_prim_recv()'s contract: the adapter's function must be assignable to tgpu_unitdata_cb
    sig = re.search(r"int\s+tgpu_adapter_unitdata\s*\(([^)]*)\)", code).group(1)
    probe += ["static int cb(%s) { (void)ud; (void)offset; (void)priv; return -1; }" % sig, "static tgpu_unitdata_cb check_cb = cb;",
              "int main(void) { (void)probe; (void)check_cb; %s return 0; }" % " ".join("(void)call_%s;" % fn for fn in sorted(calls))]
    f = tmp_path / "probe.c"
    f.write_text("\n".join(probe) + "\n")
    gcc = shutil.which("gcc")
    assert gcc
    r = subprocess.run([gcc, "-std=gnu11", "-Wall", "-Werror", "-Wno-unused-function", "-Wno-unused-variable", "-fsyntax-only", "-I" + os.path.join(root, "include"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the mirrored reference types stand behind the reference headers' include guards: with those defined (as when the
    # adapter has included the reference's headers first) the header must still parse given the types from elsewhere
    g = tmp_path / "guards.c"
    g.write_text("\n".join([
        "#include <stdint.h>", "#include <stdbool.h>",
        "#define TETRA_BURST_H", "#define TETRA_COMMON_H", "#define TETRA_TDMA_H", "#define TETRA_BURST_SYNC_H", "#define TETRA_SCRAMB_H",
        "/* what the reference's own headers would have declared (phy/tetra_burst.h, tetra_common.h, tetra_tdma.h, phy/tetra_burst_sync.h) */",
        "enum tp_sap_data_type { TPSAP_T_SB1, TPSAP_T_SB2, TPSAP_T_NDB, TPSAP_T_BBK, TPSAP_T_SCH_HU, TPSAP_T_SCH_F };",
        "enum tetra_train_seq { TETRA_TRAIN_NORM_1, TETRA_TRAIN_NORM_2, TETRA_TRAIN_NORM_3, TETRA_TRAIN_SYNC, TETRA_TRAIN_EXT };",
        "enum tetra_log_chan { TETRA_LC_UNKNOWN };", "struct tetra_tdma_time { uint16_t hn; uint32_t sn, tn, fn, mn; };",
        "struct tetra_phy_state { struct tetra_tdma_time time; };", "struct tetra_rx_state;",
        "#include \"tetra_gpu.h\"", "int main(void) { return (int)sizeof(struct tgpu_unitdata) * 0; }"]) + "\n")
    r = subprocess.run([gcc, "-std=gnu11", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"), str(g)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("ber", [0.0, 0.03, 0.08, 0.5])
def test_lane_per_slot_schedule_on_host_bit_exact(ber):
    """slot_core.h (round 6: one lane = one slot; what k_slot / k_slot_t run): a NORM_1, NORM_2 or SYNC burst through the ONE
    36-block schedule -- SB1 started at block slot 8, the first block of a two-block burst flushed in the middle of slot 17 with
    the second block's lead-in behind it, traceback and CRC register re-started there -- gives the oracle's type-2 bits and CRC
    words for every block, ties included, and the record pieces hold the type-1 bits where tg_layout.h puts them"""
    rng = np.random.default_rng(int(ber * 1000) + 5)
    code = O.scramb_get_init(262, 42, 1)

    def block(t, c):
        K, n2, n1, a = O.BLK[t]
        t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), c)
        t5 ^= (rng.random(K) < ber).astype(np.uint8)
        return t5

    for rep in range(60):
        for bt in (O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_SYNC):
            slot = rng.integers(0, 2, 510).astype(np.uint8)
            if bt == O.TRAIN_NORM_1:
                blks = [(O.T_SCH_F, code, block(O.T_SCH_F, code))]
                slot[14:230], slot[282:498] = O.scramb(code, blks[0][2])[:216], O.scramb(code, blks[0][2])[216:]
            elif bt == O.TRAIN_NORM_2:
                blks = [(O.T_NDB, code, block(O.T_NDB, code)), (O.T_NDB, code, block(O.T_NDB, code))]
                slot[14:230], slot[282:498] = O.scramb(code, blks[0][2]), O.scramb(code, blks[1][2])
            else:
                blks = [(O.T_SB1, 3, block(O.T_SB1, 3)), (O.T_SB2, code, block(O.T_SB2, code))]
                slot[94:214], slot[282:498] = O.scramb(3, blks[0][2]), O.scramb(code, blks[1][2])
            od, crc, bits, sy = emul.decode_slot(bt, emul.pack_slot(bt, slot))
            odbits = np.unpackbits(od, bitorder="little")
            for i, (t, c, t5) in enumerate(blks):
                want1, wcrc, ok, want2 = O.decode_block(t, t5, c)
                n2, n1 = O.BLK[t][1], O.BLK[t][2]
                o0 = 0 if (i == 0 and bt != O.TRAIN_SYNC) else 64 if i == 0 else 144
                assert (odbits[o0:o0 + n2] == want2).all(), (rep, bt, i)
                assert int(crc[i]) == wcrc, (rep, bt, i)
                r0 = 0 if i == 0 else 128
                assert (bits[r0:r0 + n1] == want1).all() and not bits[r0 + n1:r0 + ((n1 + 15) & ~15)].any()
            if bt == O.TRAIN_NORM_1:
                assert int(crc[1]) == 0
            if bt == O.TRAIN_SYNC:
                w = odbits[64:]
                f = lambda a, n: int("".join(str(int(x)) for x in w[a:a + n]), 2)
                cc, tn, fn, mn, mcc, mnc = f(4, 6), f(10, 2) + 1, f(12, 5), f(17, 6), f(31, 10), f(41, 14)
                assert int(sy[0]) == cc | tn << 8 | fn << 16 | mn << 24 and int(sy[1]) == mcc | mnc << 16
                assert int(sy[2]) == O.scramb_get_init(mcc, mnc, cc)
                assert not bits[64:128].any()
