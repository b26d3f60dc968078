"""GPU parity tests: the HIP path, called through the C ABI, against the oracle.

Bit-exact bar: type-1 bits, crc16, crc_ok, scrambling code, lchan, TDMA time, call order.
Run on the GPU box with:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

import oraclelib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (the HIP path has no fallback)")
    import osmo_tetra_amd as T
    T.lib()  # must load the in-tree libtetra_gpu.so
    return T


@pytest.fixture(scope="module")
def eng(T):
    e = T.Engine(0)
    yield e
    e.close()


@pytest.fixture
def topt(T):
    """a process-wide library switch (tgpu_engine_set_option) for the duration of one test"""
    saved = {}

    def set_(name, value):
        o = getattr(T, "OPT_" + name)
        saved.setdefault(o, T.get_option(o))
        T.set_option(o, value)
    yield set_
    for o, v in saved.items():
        T.set_option(o, v)


def run_plan(T, eng, slots, types, chan=None, codes=None, stride=510, offsets=None):
    """slots: (n,510) host array -> parsed records"""
    import torch
    n = len(types)
    if offsets is None:
        buf = np.zeros(n * stride + 64, np.uint8)
        offsets = np.arange(n, dtype=np.uint64) * stride
        for i in range(n):
            buf[i * stride:i * stride + 510] = slots[i]
    else:
        buf = slots
    d_stream = torch.from_numpy(buf).cuda()
    d_rec = torch.zeros(max(n, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, max(n, 1), 1 if chan is None else int(max(chan)) + 1)
    plan.load(offsets, types, chan, codes)
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)[:n]
    codes_out = plan.final_codes()
    plan.close()
    return rec, T.parse_records(rec), codes_out


def check_against_oracle(T, rec, types, slots, code, use_acc=0):
    ok, want, wcrc = O.bench_decode_slots(slots, types, code, use_acc=use_acc, want_out=True, want_crc=True)
    p = T.parse_records(rec)
    n1 = types == O.TRAIN_NORM_1
    n2 = types == O.TRAIN_NORM_2
    sb = types == O.TRAIN_SYNC
    assert (p["type"] == types).all()
    assert (p["bbk"] == want[:, :14]).all()
    assert (p["bits1"][n1] == want[n1, 14:14 + 268]).all()
    assert (p["bits1"][n2][:, :124] == want[n2, 14:14 + 124]).all()
    assert (p["bits2"][n2] == want[n2, 138:138 + 124]).all()
    assert (p["bits1"][sb][:, :60] == want[sb, 14:14 + 60]).all()
    assert (p["bits2"][sb] == want[sb, 138:138 + 124]).all()
    assert (p["crc"][:, 0] == wcrc[:, 0]).all()
    two = n2 | sb
    assert (p["crc"][two, 1] == wcrc[two, 1]).all()
    assert ((p["crc"][:, 0] == 0x1D0F) == (p["crc_ok"][:, 0] == 1)).all()
    assert ((p["crc"][two, 1] == 0x1D0F) == (p["crc_ok"][two, 1] == 1)).all()
    return ok, p


@pytest.mark.parametrize("ber", [0.0, 0.02, 0.06])
def test_config2_ndb_parity(T, eng, ber):
    """BASELINE config 2 at oracle-sized n: aligned NDB slots, scramb_init = 0, BER 0 and noisy (ties!)"""
    n = 6000
    rng = np.random.default_rng(7)
    types = np.where(rng.random(n) < 0.5, O.TRAIN_NORM_1, O.TRAIN_NORM_2).astype(np.uint8)
    slots = T.synth_slots(types, seed=11, scramb_init=0, ber=ber)
    rec, p, _ = run_plan(T, eng, slots, types)
    ok, p = check_against_oracle(T, rec, types, slots, 0)
    nblocks = int((types == O.TRAIN_NORM_1).sum() + 2 * (types == O.TRAIN_NORM_2).sum())
    if ber == 0.0:
        assert ok == nblocks
    else:
        assert 0 < ok < nblocks or ber < 0.03


def _codes_in_force(slots, types, code0=0):
    """the scrambling code the reference would use for BBK / BLK / SB2 of every slot (oracle): a SYNC slot whose SB1
    passes its CRC sets it from the SYNC PDU, starting with its own BBK and SB2 (tetra_lower_mac.c:291-300)"""
    codes = np.zeros(len(types), np.uint32)
    cur = code0
    for i, t in enumerate(types):
        if t == O.TRAIN_SYNC:
            t1, _, ok, _ = O.decode_block(O.T_SB1, slots[i][94:214], 3)
            if ok:
                f = lambda a, n: int("".join(str(int(b)) for b in t1[a:a + n]), 2)
                cur = O.scramb_get_init(f(31, 10), f(41, 14), f(4, 6))
        codes[i] = cur
    return codes


def test_garbage_and_extreme_inputs(T, eng):
    """random bits, all-ones, all-zeros in the coded fields: worst-case metrics and ties everywhere; checked
    unconditionally -- whatever code an SB1 of garbage produces (one that passes its CRC switches the code, and a
    planted valid SB1 in the middle does so for sure) is what the oracle decodes the following slots with"""
    rng = np.random.default_rng(3)
    n = 900
    types = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2] * (n // 3), np.uint8)
    slots = rng.integers(0, 2, (n, 510)).astype(np.uint8)
    slots[0:3] = 0
    slots[3:6] = 1
    slots[6:9, ::2] = 0
    good = T.synth_slots(np.array([O.TRAIN_SYNC], np.uint8), seed=5, scramb_init=O.scramb_get_init(901, 77, 9), mcc=901, mnc=77, cc=9)
    slots[450, 94:214] = good[0, 94:214]           # one valid SB1 among the garbage: the code changes here
    rec, p, _ = run_plan(T, eng, slots, types)
    codes = _codes_in_force(slots, types)
    assert (p["code"] == codes).all() and len(set(codes.tolist())) >= 2
    for c in sorted(set(codes.tolist())):
        m = codes == c
        check_against_oracle(T, rec[m], types[m], slots[m], int(c))


def test_sync_slots_set_scrambling_code(T, eng):
    """feedback loop 1: a CRC-OK SB1 switches the code for its own BBK/SB2 and everything after"""
    cells = [(262, 42, 1), (901, 77, 9)]
    rng = np.random.default_rng(5)
    parts, types = [], []
    for (mcc, mnc, cc) in cells:
        code = O.scramb_get_init(mcc, mnc, cc)
        ty = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_2, O.TRAIN_NORM_1] * 6, np.uint8)
        parts.append(T.synth_slots(ty, seed=int(rng.integers(1 << 30)), scramb_init=code, mcc=mcc, mnc=mnc, cc=cc,
                                   ber=0.01))
        types.append(ty)
    slots, types = np.concatenate(parts), np.concatenate(types)
    # prepend NDB slots scrambled with code 0: before the first SB the channel has no code
    pre_t = np.array([O.TRAIN_NORM_2, O.TRAIN_NORM_1], np.uint8)
    pre = T.synth_slots(pre_t, seed=99, scramb_init=0)
    slots, types = np.concatenate([pre, slots]), np.concatenate([pre_t, types])
    rec, p, codes_out = run_plan(T, eng, slots, types)
    # expected code per slot: replay the oracle's SB1 decodes (a noisy SB1 may fail its CRC)
    exp_code = np.zeros(len(types), np.uint32)
    cur, nfail = 0, 0
    for i, t in enumerate(types):
        if t == O.TRAIN_SYNC:
            t1, crc, ok, _ = O.decode_block(O.T_SB1, slots[i][94:214], 3)
            if ok:
                f = lambda a, n: int("".join(map(str, t1[a:a + n])), 2)
                cur = O.scramb_get_init(f(31, 10), f(41, 14), f(4, 6))
            else:
                nfail += 1
        exp_code[i] = cur
    assert (p["code"] == exp_code).all()
    assert codes_out[0] == O.scramb_get_init(*cells[1])
    assert set(exp_code.tolist()) == {0, O.scramb_get_init(*cells[0]), O.scramb_get_init(*cells[1])}
    # every run of slots that shares a code must match the oracle decoding with that code
    bounds = [0] + [i for i in range(1, len(types)) if exp_code[i] != exp_code[i - 1]] + [len(types)]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        check_against_oracle(T, rec[lo:hi], types[lo:hi], slots[lo:hi], int(exp_code[lo]))
    sb = np.where(types == O.TRAIN_SYNC)[0]
    assert int((p["crc_ok"][sb, 0] == 1).sum()) == len(sb) - nfail
    assert (p["sbf1"][sb[-1]] >> 16) == 77 and (p["sbf1"][sb[-1]] & 0xFFFF) == 901


def test_failed_sb1_keeps_previous_code(T, eng):
    code = O.scramb_get_init(262, 42, 1)
    ty = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_SYNC, O.TRAIN_NORM_2], np.uint8)
    slots = T.synth_slots(ty, seed=3, scramb_init=code)
    slots[2, 94:214] ^= (np.arange(120) % 3 == 0).astype(np.uint8)   # destroy the second SB1
    rec, p, _ = run_plan(T, eng, slots, ty)
    assert p["crc_ok"][2, 0] == 0
    assert (p["code"] == code).all()
    check_against_oracle(T, rec, ty, slots, code)


def test_multi_channel_plan_and_carry_in(T, eng):
    """several channels in one batch, each with its own carry-in code (channel-major shards)"""
    rng = np.random.default_rng(8)
    cells = [(262, 42, 1), (1, 2, 3), (1023, 16383, 63), (500, 500, 50)]
    slots, types, chan, codes = [], [], [], []
    for c, cell in enumerate(cells):
        code = O.scramb_get_init(*cell)
        n = int(rng.integers(5, 40))
        ty = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2], n).astype(np.uint8)
        slots.append(T.synth_slots(ty, seed=100 + c, scramb_init=code, ber=0.02))
        types.append(ty)
        chan.append(np.full(n, c, np.uint32))
        codes.append(code)
    slots, types, chan = np.concatenate(slots), np.concatenate(types), np.concatenate(chan)
    rec, p, codes_out = run_plan(T, eng, slots, types, chan, np.array(codes, np.uint32))
    assert (codes_out == np.array(codes, np.uint32)).all()
    for c, code in enumerate(codes):
        m = chan == c
        assert (p["code"][m] == code).all()
        check_against_oracle(T, rec[m], types[m], slots[m], code)


def test_edge_cases(T, eng):
    import torch
    # empty batch
    plan = T.Plan(eng, 4, 1)
    plan.load(np.zeros(0, np.uint64), np.zeros(0, np.uint8))
    d = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d.data_ptr())
    # capacity / argument errors are reported, not ignored
    with pytest.raises(T.TgpuError):
        plan.load(np.zeros(5, np.uint64), np.zeros(5, np.uint8))
    with pytest.raises(T.TgpuError):
        plan.load(np.zeros(2, np.uint64), np.zeros(2, np.uint8), np.array([1, 0], np.uint32), np.zeros(2, np.uint32))
    plan.close()
    # unaligned, ragged offsets + a skipped slot + plan reuse + a single slot
    ty = np.array([O.TRAIN_NORM_1, 2, O.TRAIN_NORM_2, O.TRAIN_SYNC], np.uint8)   # 2 = NORM_3: ignored like the reference
    good = ty != 2
    slots = np.zeros((4, 510), np.uint8)
    slots[good] = T.synth_slots(ty[good], seed=1, scramb_init=0)
    offs = np.array([3, 700, 1213, 2001], np.uint64)
    buf = np.zeros(2600, np.uint8)
    for o, s in zip(offs, slots):
        buf[int(o):int(o) + 510] = s
    rec, p, _ = run_plan(T, eng, buf, ty, offsets=offs)
    assert p["type"][1] != O.TRAIN_NORM_1
    check_against_oracle(T, rec[[0, 2]], ty[[0, 2]], slots[[0, 2]], 0)
    rec1, p1, _ = run_plan(T, eng, slots[:1], ty[:1])
    assert (rec1[0, 32:320] == rec[0, 32:320]).all()
    # non-binary stream bytes are flagged (the reference's contract is 0/1 bytes)
    bad = slots[:1].copy()
    bad[0, 100] = 7
    _, pb, _ = run_plan(T, eng, bad, ty[:1])
    assert pb["flags"][0] & 1 and not p1["flags"][0] & 1


# ---------------------------------------------------------------------------
# channel API: tetra_burst_sync_in() + callbacks vs the oracle receiver
# ---------------------------------------------------------------------------
KEYS = ("burst_seq", "burst_type", "type", "blk_num", "lchan", "crc_ok", "traffic", "crc", "scramb", "time", "type1", "time_str")


def assert_same_records(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for k in KEYS:
            if w["traffic"] and k in ("crc", "crc_ok", "type1", "lchan"):
                continue
            assert g[k] == w[k], (k, g["burst_seq"], g["type"])
        if w["traffic"]:
            assert g["type4"] == w["type4"]


@pytest.mark.parametrize("batch", [1, 7, 64])
@pytest.mark.parametrize("ber", [0.0, 0.03])
def test_channel_stream_parity(T, eng, batch, ber):
    stream, _ = synth.frame_stream(seed=21, nframes=6, ber=ber)
    want, wev = O.run_rx(stream)
    ch = T.Channel(eng, batch_slots=batch)
    ch.feed(stream)
    ch.flush()
    assert_same_records(ch.records, want)
    assert ch.events == wev
    ch.close()


@pytest.mark.parametrize("batch", [1, 3, 4])
@pytest.mark.parametrize("ber", [0.0, 0.03])
def test_channel_through_workgroups_that_stay(T, eng, batch, ber, topt):
    """TGPU_OPT_RING: flushes of up to four bursts go to k_burst_ring -- workgroups that poll a request line in mapped host
    memory instead of being launched per flush.  Same records and events as the oracle's receiver; a pause longer than the
    workgroups' idle time in the middle of the stream (they leave, the next flush brings them back), a loss of lock, an ignored
    burst type through the tetra_burst_rx_cb() seam, two channels at once, and a channel that is closed while its
    workgroups are still polling"""
    import time
    if batch != 3:          # (batch 3 runs on the library's default: the ring is on without anybody asking since round 6)
        topt("RING", 1)
    assert T.get_option(T.OPT_RING) == 1
    stream, _ = synth.frame_stream(seed=31, nframes=7, ber=ber)
    s = stream.copy()
    s[100 + 510 + 510 * 9 + 244 + 5] ^= 1
    want, wev = O.run_rx(s)
    ch = T.Channel(eng, batch_slots=batch)
    other = T.Channel(eng, batch_slots=2)
    half = (len(s) // 2) & ~63
    ch.feed(s[:half])
    other.feed(s[:half])
    time.sleep(0.06)
    ch.feed(s[half:])
    other.feed(s[half:])
    ch.flush()
    other.flush()
    assert_same_records(ch.records, want)
    assert ch.events == wev
    assert_same_records(other.records, want)
    res = T.sync_walk(s)
    ch2 = T.Channel(eng, batch_slots=batch)
    sl = res["slots"]
    for i, (off, typ, seq, tn) in enumerate(sl):
        ch2.burst_rx(s[off:off + 510], 2 if i == 3 else typ, tn)      # (2: TETRA_TRAIN_NORM_3, ignored)
    ch2.flush()
    assert_same_records(ch2.records, [r for r in want if r["burst_seq"] != sl[3][2]])
    if batch == 1 and ber == 0.0:       # more channels than may hold workgroups at a time: the ones beyond flush by launch
        many = [T.Channel(eng, batch_slots=1) for _ in range(34)]
        short, _ = synth.frame_stream(seed=32, nframes=2)
        wshort, _ = O.run_rx(short)
        for c in (many[0], many[-1]):
            c.feed(short)
            c.flush()
            assert_same_records(c.records, wshort)
        for c in many:
            c.close()
    topt("RING", 0)
    plain = T.Channel(eng, batch_slots=batch)
    plain.feed(s)
    plain.flush()
    assert_same_records(plain.records, want)
    for c in (ch, other, ch2, plain):
        c.close()


def test_channel_relock_and_spurious_training_sequence(T, eng):
    stream, slots = synth.frame_stream(seed=22, nframes=5)
    s = stream.copy()
    pos0 = 100 + 510
    s[pos0 + 510 * 5 + 244 + 3] ^= 1          # corrupt a NORM training sequence -> loss of lock
    s[pos0 + 510 * 20 + 50:pos0 + 510 * 20 + 72] = np.array(
        [1, 1, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0], np.uint8)  # n-sequence inside a payload
    want, wev = O.run_rx(s)
    ch = T.Channel(eng, batch_slots=16)
    ch.feed(s)
    ch.flush()
    assert_same_records(ch.records, want)
    assert ch.events == wev
    assert any(e[0] == 5 for e in wev) and any(e[0] in (3, 4) for e in wev)
    ch.close()


@pytest.mark.parametrize("batch", [1, 5, 64])
def test_channel_burst_rx_seam(T, eng, batch):
    """the tetra_burst_rx_cb() seam: bursts found by a foreign synchroniser (here: the host walk, i.e. what
    the reference's tetra_burst_sync.c would hand over, with its TDMA step count) decode and deliver exactly
    like bursts found by the library's own tetra_burst_sync_in()"""
    stream, _ = synth.frame_stream(seed=23, nframes=5, ber=0.01)
    s = stream.copy()
    s[100 + 510 + 510 * 7 + 244 + 5] ^= 1           # one loss of lock in between: steps without a burst
    want, _ = O.run_rx(s)
    res = T.sync_walk(s)
    ch = T.Channel(eng, batch_slots=batch)
    for off, typ, seq, tn in res["slots"]:
        ch.burst_rx(s[off:off + 510], typ, tn)
    ch.flush()
    assert_same_records(ch.records, want)
    assert len(want) > 20
    with pytest.raises(T.TgpuError):
        ch.burst_rx(s[:100], 0, 1)
    # a NORM_3 / extension burst is ignored, its time step is not
    ch2 = T.Channel(eng, batch_slots=4)
    sl = res["slots"]
    for i, (off, typ, seq, tn) in enumerate(sl):
        if i == 3:
            ch2.burst_rx(s[off:off + 510], 2, tn)    # TETRA_TRAIN_NORM_3
        else:
            ch2.burst_rx(s[off:off + 510], typ, tn)
    ch2.flush()
    assert sl[3][1] == O.TRAIN_NORM_1
    assert_same_records(ch2.records, [r for r in want if r["burst_seq"] != sl[3][2]])
    ch.close()
    ch2.close()


def test_channel_traffic_feedback(T, eng):
    """feedback loops 2/3: the callback (standing in for the upper MAC) flags traffic from the AACH
    and 'block 2 stolen' from block 1; blocks are then dumped / decoded exactly as in the reference"""
    rng = np.random.default_rng(31)
    cell = synth.Cell()
    aach_traffic = np.array([0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0], np.uint8)
    parts = [rng.integers(0, 2, 77).astype(np.uint8), synth.make_sb(rng, cell, 1, 1, 1), synth.make_sb(rng, cell, 2, 1, 1)]
    for i in range(12):
        a = aach_traffic if i % 3 else None
        parts.append(synth.make_norm2(rng, cell.code, a) if i % 2 else synth.make_norm1(rng, cell.code, a))
    parts.append(np.zeros(700, np.uint8))
    stream = np.concatenate(parts)

    def upper_common(set_traffic, set_stolen, d):
        if d["type"] == O.T_BBK:
            set_traffic(5 if d["type1"][1] == 1 else 0)
            set_stolen(False)
        elif d["type"] == O.T_NDB and d["blk_num"] == 1 and d["burst_seq"] % 4 == 0:
            set_stolen(True)
        return -1

    def o_upper(rx, d, offset):
        def st(v): rx.is_traffic = v
        def ss(v): rx.blk2_stolen = int(v)
        return upper_common(st, ss, d)

    want, _ = O.run_rx(stream, upper=o_upper)
    assert any(r["traffic"] for r in want) and any(not r["traffic"] and r["type"] == O.T_NDB for r in want)

    def g_upper(chan, d, offset):
        return upper_common(chan.set_traffic, chan.set_blk2_stolen, d)

    for batch in (1, 5):
        ch = T.Channel(eng, batch_slots=batch, on_unitdata=g_upper)
        ch.feed(stream)
        ch.flush()
        assert_same_records(ch.records, want)
        ch.close()


def test_channel_multi_pdu_loop(T, eng):
    """the callback's return value advances the offset like msg->head (tetra_lower_mac.c:326-352)"""
    stream, _ = synth.frame_stream(seed=23, nframes=2)
    calls_o, calls_g = [], []

    def o_upper(rx, d, offset):
        calls_o.append((d["type"], offset))
        return 40 if d["type"] in (O.T_SCH_F, O.T_NDB) else -1

    def g_upper(ch, d, offset):
        calls_g.append((d["type"], offset))
        return 40 if d["type"] in (O.T_SCH_F, O.T_NDB) else -1

    O.run_rx(stream, upper=o_upper)
    ch = T.Channel(eng, batch_slots=9, on_unitdata=g_upper)
    ch.feed(stream)
    ch.flush()
    ch.close()
    assert calls_g == calls_o and any(off > 0 for _, off in calls_g)


# ---------------------------------------------------------------------------
# BASELINE size: size-independent properties
# ---------------------------------------------------------------------------
def test_config2_full_size_roundtrip(T, eng):
    """1M NDB bursts (BASELINE config 2): encode -> decode round trip, every CRC OK, and a 2 % BER
    second pass whose noise-free blocks must still equal the payload"""
    import torch
    n = 1_000_000
    rng = np.random.default_rng(1)
    types = np.where(rng.random(n) < 0.5, O.TRAIN_NORM_1, O.TRAIN_NORM_2).astype(np.uint8)
    slots, t1 = T.synth_slots(types, seed=1, scramb_init=0, want_type1=True)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_rec = torch.empty(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types)
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = T.parse_records(d_rec.cpu().numpy().reshape(n, T.REC_BYTES))
    n1, n2 = types == O.TRAIN_NORM_1, types == O.TRAIN_NORM_2
    assert (p["crc_ok"][:, 0] == 1).all() and (p["crc_ok"][n2, 1] == 1).all()
    assert (p["crc"][:, 0] == 0x1D0F).all()
    assert (p["bits1"][n1] == t1[n1, 14:282]).all()
    assert (p["bits1"][n2][:, :124] == t1[n2, 14:138]).all()
    assert (p["bits2"][n2] == t1[n2, 138:262]).all()
    assert (p["bbk"] == 0).all()
    # spot-check a slice against the oracle as well
    sl = slice(500_000, 500_400)
    rec_sl = np.ascontiguousarray(d_rec.view(n, T.REC_BYTES)[sl].cpu().numpy())
    check_against_oracle(T, rec_sl, types[sl], slots[sl], 0)
    # second pass, the sub-run that exercises the tie rule: the same million bursts with 2 % bit
    # errors in the coded fields.  Blocks that pass their CRC must equal the payload; a 20 000-slot slice of it (40 000
    # trellises with noise) is compared with the oracle on every bit, crc word and flag
    del p
    noisy = T.synth_slots(types, seed=1, scramb_init=0, ber=0.02)
    d_stream.copy_(torch.from_numpy(noisy.reshape(-1)))
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = T.parse_records(d_rec.cpu().numpy().reshape(n, T.REC_BYTES))
    ok1, ok2 = p["crc_ok"][:, 0] == 1, p["crc_ok"][:, 1] == 1
    assert 0.2 < ok1[n1].mean() < 0.999 and 0.5 < ok1[n2].mean() < 0.9999      # noise bites, most blocks survive
    assert (p["bits1"][n1 & ok1] == t1[n1 & ok1, 14:282]).all()
    assert (p["bits1"][n2 & ok1][:, :124] == t1[n2 & ok1, 14:138]).all()
    assert (p["bits2"][n2 & ok2] == t1[n2 & ok2, 138:262]).all()
    sl = slice(300_000, 320_000)
    rec_sl = np.ascontiguousarray(d_rec.view(n, T.REC_BYTES)[sl].cpu().numpy())
    check_against_oracle(T, rec_sl, types[sl], noisy[sl], 0)
    plan.close()


def test_front_kernel_packing(T, eng, topt):
    """k_front's packed code words == the layout function both sides are built from (tg_layout.h),
    for all burst types, odd offsets and the last slot of a buffer (no read past byte 509)"""
    topt("BURST_MAX", int("0"))        # (small batches would bypass k_front)
    import torch
    import emul
    rng = np.random.default_rng(12)
    types = np.array([O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_SYNC] * 50 + [O.TRAIN_SYNC], np.uint8)
    n = len(types)
    slots = rng.integers(0, 2, (n, 510)).astype(np.uint8)
    offs = np.cumsum(np.concatenate([[1], 510 + rng.integers(0, 9, n - 1)])).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + 510, np.uint8)     # ends exactly with the last slot
    for o, s in zip(offs, slots):
        buf[int(o):int(o) + 510] = s
    d_stream = torch.from_numpy(buf).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(offs, types)
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    got = plan.read_packed()
    for i in range(n):
        want = emul.pack_slot(int(types[i]), slots[i])
        assert (got[i, :19] == want[:19]).all(), i
        assert got[i, 19] & 0xFF == types[i] and (got[i, 19] >> 16) == (214 if types[i] == O.TRAIN_SYNC else 244)
    plan.close()


def test_wire_records_equal_full_records(T, eng):
    """the 40-byte transport form written by the trellis kernels expands to exactly the 320-byte record, and is
    byte for byte what the host packer makes of that record (every byte of a decoded slot's wire record is written)"""
    import torch
    code = O.scramb_get_init(262, 42, 1)
    types = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2] * 40, np.uint8)
    slots = T.synth_slots(types, seed=77, scramb_init=code, ber=0.03)
    n = len(types)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_wire = torch.full((n * T.WIRE_BYTES,), 0xA5, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([code], np.uint32))
    plan.set_wire(d_wire.data_ptr())
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    p = T.parse_records(rec)
    wire = d_wire.cpu().numpy().reshape(n, T.WIRE_BYTES)
    back = T.wire_unpack(wire, slot_ids=np.arange(n), codes=p["code"])
    assert (back == rec).all()
    assert (wire == T.wire_pack(rec)).all()
    assert 0 < int(p["crc_ok"].sum()) < 2 * n          # good and failed blocks both in there
    plan.close()


# ---------------------------------------------------------------------------
# BASELINE config 3: burst-sync correlation front end on the GPU
# ---------------------------------------------------------------------------
def _mutated_stream(seed, nframes=6):
    rng = np.random.default_rng(seed)
    stream, slots = synth.frame_stream(seed=seed, nframes=nframes, lead_in=int(rng.integers(0, 200)), ber=0.01)
    y = np.array([1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1], np.uint8)
    n = np.array([1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0], np.uint8)
    p0 = np.flatnonzero((np.lib.stride_tricks.sliding_window_view(stream, 38) == y).all(axis=1))[0] + 296
    s = stream.copy()
    for i in rng.choice(len(slots), 5, replace=False):
        base = p0 + 510 * int(i)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            s[base + (214 if slots[i][0] == O.TRAIN_SYNC else 244) + 4] ^= 1
        elif kind == 1:
            s[base + 60:base + 82] = n
        elif kind == 2:
            s[base + 300:base + 338] = y
        else:
            s[base + int(rng.integers(0, 21)):][:22] = n
    return s


def test_stream_classification_kernel(T, eng):
    """k_front_stream's classification words == the definition (first y/n/p hit at offset >= 21 of the
    window the reference would search), checked with a numpy statement of it"""
    import torch
    from test_stream_sync_cpu import emul_cls, emul_ysum
    for seed in (3, 4):
        s = _mutated_stream(seed)
        res = T.sync_walk(s)
        anchor = res["slots"][0][0]
        n = (len(s) - anchor) // 510
        d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
        got, ys = T.sync_classify(eng, d.data_ptr(), len(s), 64, anchor, n, with_ysum=True)
        want = emul_cls(s, anchor, 64)
        assert (got & 0x05FFFFFF).tolist() == (want & 0x05FFFFFF).tolist()
        assert ys.tolist() == emul_ysum(s, anchor).tolist()
        assert (ys != 0xFFFF).sum() >= 3
        assert T.sync_classify(eng, d.data_ptr(), len(s), 64, anchor, n).tolist() == got.tolist()
        # feeds of 128 / 256 bytes: the search window reaches up to 765 bytes from the slot's start, all of it inside the
        # kernel's view for such feeds (TG_VIEW_OF; no TG_CLS_CLIPPED); hits beyond the slot and its successor's first bytes included
        from test_stream_sync_cpu import SEQ_N
        for chunk in (128, 256):
            s2 = s.copy()
            planted = []
            for i in range(5, n - 2):         # slots whose own sequence is damaged and whose window (chunk-aligned end) reaches far
                bs = anchor + 510 * i
                w = min(-(-(bs + 510) // chunk) * chunk, len(s2)) - bs
                if (got[i] & 0xFFFFFE) == (244 << 8) and w >= 604 and len(planted) < 3 and (not planted or i > planted[-1] + 3):
                    s2[bs + 244 + 3] ^= 1
                    s2[bs + 580:bs + 602] = SEQ_N
                    planted.append(i)
            assert planted
            bare = planted[-1] + 5          # a slot that loses its sequence and gets nothing in exchange: "nothing in the window",
            while (got[bare] & 0xFFFFFE) != (244 << 8) or (got[bare + 1] & 0xFF) == 0xFF:     # with the next slot's sequence as the view's first
                bare += 1
            s2[anchor + 510 * bare + 244 + 3] ^= 1
            d2 = torch.from_numpy(np.concatenate([s2, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
            g2 = T.sync_classify(eng, d2.data_ptr(), len(s2), chunk, anchor, n)
            want = emul_cls(s2, anchor, chunk)
            assert (g2 & 0x7DFFFFFF).tolist() == (want & 0x7DFFFFFF).tolist(), chunk
            assert not (g2 >> 24 & 4).any()
            assert all((g2[i] >> 8 & 0xFFFF) == 580 for i in planted), "a hit beyond a 64-byte feed's window"
            nxt = int(got[bare + 1])
            assert (g2[bare] & 0xFF) == 0xFF and (g2[bare] >> 8 & 0xFFFF) == 510 + (nxt >> 8 & 0xFFFF) and \
                (g2[bare] >> 28 & 7) == (nxt & 0xFF) + 1, hex(int(g2[bare]))


def _hostile_stream(T, seed, nslots, lead_in, shift=True):
    """a long mixed stream with everything the search has to get right: damaged training sequences, spurious
    n / p / y sequences anywhere in a slot (below offset 21, past offset 472, across slot boundaries), bytes
    other than 0 / 1, a missing and an extra byte (the grid then runs beside the bursts)"""
    rng = np.random.default_rng(seed)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, nslots // 8 + 1)[:nslots]
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=seed, scramb_init=0x41802A07, ber=0.01)
    s = np.concatenate([rng.integers(0, 2, lead_in).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
    y = np.array([1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1], np.uint8)
    n = np.array([1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0], np.uint8)
    p = np.array([0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0], np.uint8)
    base0 = lead_in + 510
    for i in rng.choice(nslots - 2, nslots // 12, replace=False):
        base = base0 + 510 * int(i)
        kind = int(rng.integers(0, 9))
        seq = (y, n, p)[int(rng.integers(0, 3))]
        if kind == 0:
            s[base + (214 if types[i] == 3 else 244) + int(rng.integers(0, 22))] ^= 1
        elif kind == 1:
            s[base + int(rng.integers(0, 21)):][:len(seq)] = seq          # below the look-ahead bound
        elif kind == 2:
            s[base + int(rng.integers(21, 214)):][:len(seq)] = seq        # before the real one
        elif kind == 3:
            s[base + int(rng.integers(440, 515)):][:len(seq)] = seq       # end of the slot / across the boundary
        elif kind == 4:
            s[base + int(rng.integers(0, 510))] = int(rng.choice([2, 3, 0x80, 0xff]))
        elif kind == 5:
            s[base + (214 if types[i] == 3 else 244) + 3] = 2             # training sequence with a non-binary byte
        elif kind == 6:
            s[base + int(rng.integers(290, 470)):][:38] = y               # second SYNC sequence in the slot
        elif kind == 7:
            o = int(rng.integers(22, 200))
            s[base + o:][:22] = n
            s[base + o + 30:][:38] = y
        else:
            s[base + 214:base + 252] = y                                   # SYNC sequence where a NORM burst has none
    if shift:
        s = np.delete(s, base0 + 510 * (nslots // 2) + 77)                    # one byte lost ...
        s = np.insert(s, base0 + 510 * (3 * nslots // 4) + 300, 1)           # ... and one gained later
    return s


@pytest.mark.parametrize("seed,lead_in,chunk", [(1, 100, 64), (2, 7, 64), (3, 333, 100), (4, 1000, 510), (5, 41, 1)])
def test_stream_front_packed_bits_equals_per_position(T, eng, seed, lead_in, chunk, topt):
    """k_front_stream (packed bits, bit-parallel search, + k_front_stream_fix) == k_front_stream_v1 (the per-position
    form on every slot): classification words, SYNC summaries and packed slots, bit for bit, on hostile streams of
    every 16-byte alignment class"""
    import torch
    nsl = 3000
    s = _hostile_stream(T, seed, nsl, lead_in, shift=bool(seed & 1))      # (even seeds stay on one grid: packed slots compared)
    anchor = lead_in + 510 + seed            # deliberately any alignment, on or off the burst grid
    n = (len(s) - anchor) // 510
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    out = {}
    for mode in ("1", "0"):
        topt("STREAM_EXACT", int(mode))
        cls, ys = T.sync_classify(eng, d.data_ptr(), len(s), chunk, anchor, n, with_ysum=True)
        plan = T.Plan(eng, n + 8, 1)
        g = T.GridSync(eng, plan, s, d.data_ptr(), chunk)
        res = g.finish(burst_events=False)
        packed = plan.read_packed() if res["ngrid"] and not res["noffgrid"] else None
        out[mode] = (cls, ys, packed, res)
        plan.close()
    a, b = out["1"], out["0"]
    assert (a[0] == b[0]).all(), np.flatnonzero(a[0] != b[0])[:10]
    assert (a[1] == b[1]).all(), np.flatnonzero(a[1] != b[1])[:10]
    assert (a[0] & 0xff != 0xff).sum() > n // 2                 # most slots carry a burst ...
    assert ((a[0] >> 24) & 1).sum() == 0            # (round 3: sequences below offset 21 are evaluated, not flagged)
    assert ((a[0] >> 25) & 1).sum() > 0             # ... and the hard cases are in there
    assert (a[2] is None) == (b[2] is None) and (a[2] is None) == bool(seed & 1)
    if a[2] is not None:
        assert (a[2] == b[2]).all(), np.flatnonzero((a[2] != b[2]).any(axis=1))[:10]
    assert a[3]["events"] == b[3]["events"] and a[3]["nslots"] == b[3]["nslots"]
    # and the synchroniser built on it == the oracle's tetra_burst_sync_in() on the same bytes: events (lock, loss of
    # lock, misplaced sequences, every burst with its window) and the bursts handed on
    if chunk >= 21:
        _, wev = O.run_rx(s, chunk=chunk)
        res = T.sync_stream(eng, s, d.data_ptr(), chunk)
        assert res["events"] == wev


@pytest.mark.parametrize("seed", [1, 2, 5])
def test_config3_stream_end_to_end(T, eng, seed):
    """stream -> GPU sync front end -> plan decode -> in-order delivery == oracle tetra-rx equivalent"""
    import torch
    s = _mutated_stream(seed, nframes=8)
    want, wev = O.run_rx(s)
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    res = T.sync_stream(eng, s, d.data_ptr())
    assert res["events"] == wev
    slots = res["slots"]
    n = len(slots)
    ch = T.Channel(eng, batch_slots=1)
    plan = T.Plan(eng, max(n, 1), 1)
    ch0_code = ch.scramb_init()
    plan.load(np.array([x[0] for x in slots], np.uint64), np.array([x[1] for x in slots], np.uint8), None,
              np.array([ch0_code], np.uint32))
    d_rec = torch.zeros(max(n, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec_a = d_rec.cpu().numpy().copy()
    ch.deliver(slots, s, rec_a)
    assert_same_records(ch.records, want)
    # the same batch loaded straight from the slot table, without the per-burst events
    res2 = T.sync_stream(eng, s, d.data_ptr(), burst_events=False)
    assert res2["slots"] == slots and [e for e in wev if e[0] != 2] == res2["events"]
    d_rec.zero_()
    plan.load_slots(res2, ch0_code)
    plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_rec.cpu().numpy() == rec_a).all()
    plan.close()
    ch.close()


# ---------------------------------------------------------------------------
# BASELINE config 5: float phase stream -> bits / soft values -> soft-decision decode
# ---------------------------------------------------------------------------
def test_float_to_bits_kernel(T, eng):
    """device slicer == float_to_bits.c (golden vectors from the real binary) incl. 0, +-2, NaN, inf, and the
    sequential pseudo-AFC variant; soft values == the oracle's definition"""
    import json
    import os
    import torch
    refv = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.json")))
    sig = np.frombuffer(bytes.fromhex(refv["float_to_bits"]["input_f32_hex"]), np.float32).copy()
    d_in = torch.from_numpy(sig).cuda()
    n = len(sig)
    for args, want in refv["float_to_bits"]["runs"]:
        d_bits = torch.zeros(2 * n, dtype=torch.uint8, device="cuda")
        if "-a" in args:
            fv = float(args[args.index("-f") + 1]) if "-f" in args else 0.0001
            fg = float(args[args.index("-F") + 1]) if "-F" in args else 0.0
            eng.float_to_bits_afc(d_in.data_ptr(), n, d_bits.data_ptr(), fv, fg)
        else:
            eng.float_to_bits(d_in.data_ptr(), n, d_bits.data_ptr())
            torch.cuda.synchronize()
        assert O.bitstr(d_bits.cpu().numpy()) == want, args
    # a longer random stream (vector path + tail) and the soft values
    rng = np.random.default_rng(6)
    phi = (rng.choice([-3, -1, 1, 3], 100003) + rng.normal(0, 0.7, 100003)).astype(np.float32)
    phi[::1000] = 0.0
    phi[1::1000] = 2.0
    phi[2::1000] = -2.0
    phi[3::1000] = np.nan
    d_in = torch.from_numpy(phi).cuda()
    d_bits = torch.zeros(2 * len(phi), dtype=torch.uint8, device="cuda")
    d_soft = torch.zeros(2 * len(phi), dtype=torch.int8, device="cuda")
    eng.float_to_bits(d_in.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr())
    torch.cuda.synchronize()
    assert (d_bits.cpu().numpy() == O.float_to_bits(phi)).all()
    assert (d_soft.cpu().numpy() == O.float_to_soft(phi)).all()
    st = eng.float_to_bits_afc(d_in.data_ptr(), 5000, d_bits.data_ptr(), 0.01, 0.1)
    assert (d_bits.cpu().numpy()[:10000] == O.float_to_bits(phi[:5000], True, 0.01, 0.1)).all()


@pytest.mark.parametrize("sigma", [0.0, 0.5, 0.9, 1.6])
def test_config5_soft_decision_parity(T, eng, sigma):
    """float phases -> soft values (device) -> soft-decision decode == the oracle's soft chain, bit-exact
    (integer metrics); sigma = 0 additionally equals the hard path"""
    import torch
    rng = np.random.default_rng(int(sigma * 10) + 3)
    code = O.scramb_get_init(262, 42, 1)
    types = np.array([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_1] * 60, np.uint8)
    n = len(types)
    slots = T.synth_slots(types, seed=31, scramb_init=code)
    bits = slots.reshape(-1)
    phi = (O.bits_to_phase(bits) + rng.normal(0, sigma, len(bits) // 2)).astype(np.float32)
    d_phi = torch.from_numpy(phi).cuda()
    d_bits = torch.zeros(len(bits) + 64, dtype=torch.uint8, device="cuda")
    d_soft = torch.zeros(len(bits) + 64, dtype=torch.int8, device="cuda")
    eng.float_to_bits(d_phi.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr())
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([code], np.uint32))
    plan.execute_soft(d_soft.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    p = T.parse_records(rec)
    soft = d_soft.cpu().numpy()[:len(bits)].reshape(n, 510)
    assert (soft.reshape(-1) == O.float_to_soft(phi)).all()
    nok = 0
    for i in range(n):
        s = soft[i]
        t = types[i]
        if t == O.TRAIN_SYNC:
            blocks = [(O.T_SB1, s[94:214], 3, p["bits1"][i][:60], 0), (O.T_SB2, s[282:498], code, p["bits2"][i], 1)]
            bbk = s[252:282]
        elif t == O.TRAIN_NORM_2:
            blocks = [(O.T_NDB, s[14:230], code, p["bits1"][i][:124], 0), (O.T_NDB, s[282:498], code, p["bits2"][i], 1)]
            bbk = np.concatenate([s[230:244], s[266:282]])
        else:
            blocks = [(O.T_SCH_F, np.concatenate([s[14:230], s[282:498]]), code, p["bits1"][i], 0)]
            bbk = np.concatenate([s[230:244], s[266:282]])
        for bt, sv, c, got, which in blocks:
            w1, wcrc, wok, _ = O.decode_block_soft(bt, sv, c)
            assert (got == w1).all(), (i, bt)
            assert p["crc"][i, which] == wcrc and p["crc_ok"][i, which] == int(wok)
            nok += wok
        wb = O.decode_block_soft(O.T_BBK, bbk, code)[0]
        assert (p["bbk"][i] == wb).all()
    if sigma == 0.0:
        d_rec2 = torch.zeros_like(d_rec)
        plan.execute(d_bits.data_ptr(), d_rec2.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert (d_rec2.cpu().numpy().reshape(n, -1)[:, 2:] == rec[:, 2:]).all()
        assert nok == 60 * 2 + 120 + 60 * 2
    plan.close()


# ---------------------------------------------------------------------------
# BASELINE config 1 (plumbing) and config 4 (64 channels in 8 shards)
# ---------------------------------------------------------------------------
def test_config1_plumbing_stream(T, eng):
    """64 zero bytes + SB + SB + NDB(SCH/F) + SB + 700 zero bytes: the first SB only gives lock, the second
    decodes to the reference's golden SYNC PDU"""
    rng = np.random.default_rng(1)
    cell = synth.Cell(262, 42, 0)
    sb = lambda: synth.make_sb(rng, cell, 1, 1, 1)
    stream = np.concatenate([np.zeros(64, np.uint8), sb(), sb(), synth.make_norm1(rng, cell.code), sb(),
                             np.zeros(700, np.uint8)])
    ch = T.Channel(eng, batch_slots=4)
    ch.feed(stream)
    ch.flush()
    want, wev = O.run_rx(stream)
    assert_same_records(ch.records, want)
    sb1 = [r for r in ch.records if r["type"] == O.T_SB1]
    assert len(sb1) == 2 and all(r["crc_ok"] for r in sb1)
    assert O.bitstr(np.frombuffer(sb1[0]["type1"], np.uint8)) == \
        "000000000000000010000010000000001000001100000000010101000000"
    assert sb1[0]["time"] == (1, 1, 1) and ch.records[1]["scramb"] == O.scramb_get_init(262, 42, 0)
    ch.close()


def test_config4_64_channels_in_8_shards(T, eng):
    """64 independent channel streams (own cell each), sharded 8 per rank; every shard is one multi-channel
    plan; the wire records of all shards, concatenated in rank order (what the RCCL gather delivers to rank 0),
    expand to exactly the blocks the oracle decodes channel by channel"""
    import torch
    from osmo_tetra_amd import dist as tdist
    nchan, world = 64, 8
    streams = []
    for c in range(nchan):
        cell = synth.Cell(200 + c, 1000 + 3 * c, c % 64)
        s, _ = synth.frame_stream(seed=500 + c, nframes=2, cell=cell, ber=0.01, lead_in=100 + c)
        streams.append(s)
    gathered = []
    meta = []
    for rank in range(world):
        lo, hi = tdist.shard_channels(nchan, rank, world)
        parts, offs, types, chans = [], [], [], []
        base = 0
        for k, c in enumerate(range(lo, hi)):
            s = streams[c]
            d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
            res = T.sync_stream(eng, s, d.data_ptr())
            sa = res["slot_arr"]
            offs.append(sa["off"] + base)
            types.append(sa["type"])
            chans.append(np.full(len(sa), k, np.uint32))
            meta.append((c, sa))
            parts.append(s)
            base += len(s)
        big = np.concatenate(parts + [np.zeros(T.STREAM_SLACK, np.uint8)])
        offs, types, chans = np.concatenate(offs), np.concatenate(types), np.concatenate(chans)
        n = len(types)
        d_stream = torch.from_numpy(big).cuda()
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        d_wire = torch.zeros(n * T.WIRE_BYTES, dtype=torch.uint8, device="cuda")
        plan = T.Plan(eng, n, hi - lo)
        plan.load(offs, types, chans, np.zeros(hi - lo, np.uint32))
        plan.set_wire(d_wire.data_ptr())
        plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gathered.append(d_wire.cpu().numpy().reshape(n, T.WIRE_BYTES))
        plan.close()
    wire = np.concatenate(gathered)
    pos = 0
    for c, sa in meta:
        w = wire[pos:pos + len(sa)]
        pos += len(sa)
        want, _ = O.run_rx(streams[c])
        rec = T.wire_unpack(w)
        got = []
        for i in range(len(sa)):
            for b in T.record_blocks(rec[i]):
                got.append((int(sa["burst_seq"][i]), b["type"], b["blk_num"], b["crc_ok"], b["crc"], b["type1"]))
        exp = [(r["burst_seq"], r["type"], r["blk_num"], r["crc_ok"], r["crc"], r["type1"]) for r in want]
        assert got == exp, c
    assert pos == len(wire)


def test_slot_offsets_beyond_4_gib(T, eng):
    """maximum sizes: slot offsets are 56-bit; slots parked above the 4 GiB mark of one device buffer decode
    like the same slots at offset 0 (64-bit address arithmetic in the front end)"""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 6 * 2**30:
        pytest.skip("needs 6 GiB of free HBM")
    n = 64
    rng = np.random.default_rng(77)
    ty = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_SYNC], n).astype(np.uint8)
    slots = T.synth_slots(ty, seed=5, scramb_init=0, ber=0.02)
    big = torch.zeros(5 * 2**30, dtype=torch.uint8, device="cuda")
    base = 4 * 2**30 + 12345
    offs = base + np.arange(n, dtype=np.uint64) * 1021         # ragged, unaligned
    flat = torch.from_numpy(slots)
    for i in range(n):
        big[int(offs[i]):int(offs[i]) + 510] = flat[i].cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(offs, ty)
    plan.execute(big.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec_hi = d_rec.cpu().numpy().reshape(n, T.REC_BYTES).copy()
    plan.close()
    del big
    rec_lo, _, _ = run_plan(T, eng, slots, ty)
    assert (rec_hi == rec_lo).all()
    with pytest.raises(T.TgpuError):
        T.Plan(eng, 1, 1).load(np.array([1 << 56], np.uint64), ty[:1])


def test_sync_stream_that_never_locks(T, eng):
    """a stream without any SYNC training sequence: no GPU classification is launched, no slots, the events
    are the oracle's (none)"""
    import torch
    rng = np.random.default_rng(3)
    s = rng.integers(0, 2, 20000).astype(np.uint8)
    want, wev = O.run_rx(s)
    assert not want
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    res = T.sync_stream(eng, s, d.data_ptr())
    assert res["slots"] == [] and res["events"] == wev
    plan = T.Plan(eng, 4, 1)
    plan.load_slots(res, 0)
    d_rec = torch.zeros(4 * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr())
    torch.cuda.synchronize()
    plan.close()


@pytest.mark.parametrize("seed", [1, 2, 5, 9])
def test_config3_grid_plan_matches_slot_table_plan(T, eng, seed):
    """stream mode without a host slot table (tgpu_sync_stream_grid: bitmap from the walk, lists built on the
    device, packed slots reused from the classification pass) == the slot-table path, record for record"""
    import torch
    s = _mutated_stream(seed, nframes=8)
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    hs = torch.cuda.current_stream().cuda_stream
    res = T.sync_stream(eng, s, d.data_ptr())
    slots = res["slots"]
    n = len(slots)
    plan = T.Plan(eng, max(n, 1), 1)
    plan.load_slots(res, 0)
    d_rec = torch.zeros(max(n, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    want = d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)
    codes_want = plan.final_codes().tolist()
    plan.close()

    ngrid_max = len(s) // 510 + 1
    gplan = T.Plan(eng, ngrid_max, 1)
    g = T.sync_stream_grid(eng, gplan, s, d.data_ptr())
    assert g["events"] == res["events"] and g["anchor"] == res["anchor"]
    on = [(o - g["anchor"]) // 510 for (o, t, q, tn) in slots if (o - g["anchor"]) % 510 == 0 and o >= g["anchor"]]
    assert g["noffgrid"] == n - len(on)
    if g["noffgrid"]:
        pytest.skip("stream re-locks off the grid: the caller falls back to the slot table")
    assert T.grid_indices(g).tolist() == on and g["nslots"] == n
    d_rec2 = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    gplan.execute(d.data_ptr(), d_rec2.data_ptr(), hs)
    torch.cuda.synchronize()
    got = d_rec2.cpu().numpy().reshape(-1, T.REC_BYTES)[on]
    pw, pg = T.parse_records(want), T.parse_records(got)
    for k in pw:
        if k == "slot":
            continue
        assert (np.asarray(pw[k]) == np.asarray(pg[k])).all(), k
    # the scrambling code the channel carries on after the batch
    assert gplan.final_codes().tolist() == codes_want
    gplan.close()


# ---------------------------------------------------------------------------
# block mode: the tp_sap_udata_ind() unit, incl. SCH/HU
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("ber", [0.0, 0.03, 0.08])
def test_block_mode_all_block_types(T, eng, ber):
    """type-5 blocks on their own (SB1, SB2, NDB, BBK, SCH/HU, SCH/F; several scrambling codes, ragged
    unaligned offsets, noise -> trellis ties) == the oracle's type-5 -> type-1 chain, bit for bit"""
    import torch
    rng = np.random.default_rng(int(ber * 100) + 40)
    n = 1500
    kinds = [O.T_SB1, O.T_SB2, O.T_NDB, O.T_BBK, O.T_SCH_HU, O.T_SCH_F]
    types = rng.choice(kinds, n).astype(np.uint8)
    codes_pool = np.array([0, 3, 0x41802A07, 0x12345677, 0xFFFFFFFF], np.uint32)
    codes = codes_pool[rng.integers(0, len(codes_pool), n)]
    offs, bufs, pos = [], [], 7
    t5s = []
    for i in range(n):
        t = int(types[i])
        K, n2, n1, a = O.BLK[t]
        enc_code = 3 if t == O.T_SB1 else int(codes[i])
        if t == O.T_BBK:
            t5 = O.encode_bbk(rng.integers(0, 2, 14).astype(np.uint8), enc_code)
        else:
            t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), enc_code)
        t5 = t5 ^ (rng.random(K) < ber).astype(np.uint8)
        t5s.append(t5)
        offs.append(pos)
        bufs.append((pos, t5))
        pos += K + int(rng.integers(0, 5))
    buf = np.zeros(pos, np.uint8)                  # no slack after the last block: reads must stay inside
    for p0, t5 in bufs:
        buf[p0:p0 + len(t5)] = t5
    d = torch.from_numpy(buf).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 8)
    plan.load_blocks(np.array(offs, np.uint64), types, codes)
    plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    p = T.parse_records(rec)
    nok = 0
    for i in range(n):
        t = int(types[i])
        K, n2, n1, a = O.BLK[t]
        code = 3 if t == O.T_SB1 else int(codes[i])
        assert p["type"][i] == t and p["slot"][i] == i and p["code"][i] == code
        if t == O.T_BBK:
            want = (t5s[i] ^ O.scramb_seq(code, 30))[:14]
            assert (p["bbk"][i] == want).all() and p["crc_ok"][i, 0] == 1
            continue
        w1, wcrc, wok, _ = O.decode_block(t, t5s[i], code)
        assert (p["bits1"][i][:n1] == w1).all(), (i, t)
        assert p["crc"][i, 0] == wcrc and p["crc_ok"][i, 0] == int(wok)
        nok += int(wok)
    assert nok > (0.9 * n * 5 / 6 if ber == 0 else 10)
    # too many distinct codes for the plan, bad type
    with pytest.raises(T.TgpuError):
        T.Plan(eng, 4, 1).load_blocks(np.zeros(2, np.uint64), np.array([2, 2], np.uint8), np.array([1, 2], np.uint32))
    with pytest.raises(T.TgpuError):
        plan.load_blocks(np.zeros(1, np.uint64), np.array([9], np.uint8), np.zeros(1, np.uint32))
    plan.close()


def test_block_mode_extreme_inputs(T, eng):
    """the trellis kernels on inputs that drive the path metrics to their extremes -- all zeros, all ones, alternating and
    period-3 patterns, pure noise (BER 0.5), a clean code word with one burst of errors -- for every block kind and several
    scrambling codes == the oracle.  (vit_core.h keeps all metrics at or above a floor of 2 because the difference form of
    a two-bit step adds n - 2m >= -2 to a predecessor's metric in unsigned packed arithmetic; eight bits hold the metric.)"""
    import torch
    rng = np.random.default_rng(77)
    kinds = [O.T_SB1, O.T_SB2, O.T_NDB, O.T_SCH_HU, O.T_SCH_F]
    codes_pool = [0, 3, 0x41802A07, 0xFFFFFFFF]
    blocks = []
    for t in kinds:
        K, n2, n1, a = O.BLK[t]
        pats = [np.zeros(K, np.uint8), np.ones(K, np.uint8), (np.arange(K) % 2).astype(np.uint8), (np.arange(K) % 3 == 0).astype(np.uint8),
                (np.arange(K) % 3 != 1).astype(np.uint8)]
        pats += [rng.integers(0, 2, K).astype(np.uint8) for _ in range(40)]
        for _ in range(20):
            code = 3 if t == O.T_SB1 else int(rng.choice(codes_pool))
            t5 = O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), code)
            p0 = int(rng.integers(0, K - 24))
            t5[p0:p0 + 24] ^= 1
            pats.append(t5)
        for q in pats:
            blocks.append((t, 3 if t == O.T_SB1 else int(rng.choice(codes_pool)), q))
    n = len(blocks)
    types = np.array([b[0] for b in blocks], np.uint8)
    codes = np.array([b[1] for b in blocks], np.uint32)
    offs, pos = [], 3
    for t, _, q in blocks:
        offs.append(pos)
        pos += len(q) + 1
    buf = np.zeros(pos, np.uint8)
    for o, (_, _, q) in zip(offs, blocks):
        buf[o:o + len(q)] = q
    d = torch.from_numpy(buf).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 8)
    plan.load_blocks(np.array(offs, np.uint64), types, codes)
    plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = T.parse_records(d_rec.cpu().numpy().reshape(n, T.REC_BYTES))
    for i, (t, code, q) in enumerate(blocks):
        K, n2, n1, a = O.BLK[t]
        w1, wcrc, wok, _ = O.decode_block(t, q, code)
        assert (p["bits1"][i][:n1] == w1).all(), (i, t)
        assert p["crc"][i, 0] == wcrc and p["crc_ok"][i, 0] == int(wok), (i, t)
    plan.close()


def test_rm3014_decode_flag(T, eng):
    """optional AACH decoding: with the flag on, BBK bit errors are corrected exactly like the oracle's exhaustive
    minimum-distance decoder (slot mode and block mode); with the flag off the reference's behaviour stays"""
    import torch
    rng = np.random.default_rng(91)
    n = 600
    ty = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2], n).astype(np.uint8)   # (a SYNC burst would switch the code)
    slots = T.synth_slots(ty, seed=9, scramb_init=0)
    bbk_pos = {}
    norm_pos = list(range(230, 244)) + list(range(266, 282))
    nflip = rng.integers(0, 6, n)
    for i in range(n):
        pos = bbk_pos.get(int(ty[i]), norm_pos)
        for b in rng.choice(30, int(nflip[i]), replace=False):
            slots[i, pos[int(b)]] ^= 1
    rec_off, p_off, _ = run_plan(T, eng, slots, ty)
    check_against_oracle(T, rec_off, ty, slots, 0)           # flag off: unchanged reference behaviour
    assert (rec_off[:, 28] == 0).all()
    # flag on
    buf = np.zeros(n * 510 + 64, np.uint8)
    for i in range(n):
        buf[i * 510:i * 510 + 510] = slots[i]
    d = torch.from_numpy(buf).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.set_rm_decode(True)
    plan.load(np.arange(n, dtype=np.uint64) * 510, ty)
    plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec_on = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    p_on = T.parse_records(rec_on)
    fixed = 0
    for i in range(n):
        pos = bbk_pos.get(int(ty[i]), norm_pos)
        rx = 0
        for b in range(30):                                  # scramb_init 0 -> mask 0: stream bits are type-4 bits
            rx |= int(slots[i, pos[b]]) << (29 - b)
        want, werr = O.rm3014_decode_ml(rx)
        got = 0
        for b in range(14):
            got |= int(p_on["bbk"][i][b]) << (13 - b)
        assert got == want and rec_on[i, 28] == werr, (i, nflip[i])
        if nflip[i] <= 3:
            assert werr == nflip[i]
            fixed += 1
    assert fixed > n // 3
    # everything but the BBK bytes and the error count is the same record
    same = np.ones(T.REC_BYTES, bool); same[28] = False; same[32:46] = False
    assert (rec_on[:, same] == rec_off[:, same]).all()
    plan.close()
    # block mode
    bb = np.stack([slots[i, bbk_pos.get(int(ty[i]), norm_pos)] for i in range(n)])
    dblk = torch.from_numpy(bb.reshape(-1)).cuda()
    d_rec2 = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    bplan = T.Plan(eng, n, 1)
    bplan.set_rm_decode(True)
    bplan.load_blocks(np.arange(n, dtype=np.uint64) * 30, np.full(n, O.T_BBK, np.uint8), np.zeros(n, np.uint32))
    bplan.execute(dblk.data_ptr(), d_rec2.data_ptr())
    torch.cuda.synchronize()
    rec_b = d_rec2.cpu().numpy().reshape(n, T.REC_BYTES)
    assert (rec_b[:, 32:46] == rec_on[:, 32:46]).all() and (rec_b[:, 28] == rec_on[:, 28]).all()
    bplan.close()


def test_walk_bitmap_steady_state_equals_the_general_path(T, eng):
    """the walk's closed-form steady state on k_cls_plain's bitmap (no per-burst events) against its own general path
    (per-burst events on, every slot on its own) on 40 damaged streams of 200..4000 slots: lock-loss / re-lock events,
    delivered grid slots, counters and final state; chunk sizes 32 / 64 / 128; the oracle receiver on the shorter ones"""
    import torch
    from test_stream_sync_cpu import SEQ_Y, SEQ_N
    rng = np.random.default_rng(20260928)
    hs = torch.cuda.current_stream().cuda_stream
    nloss = 0
    for trial in range(40):
        n = int(rng.integers(200, 4000))
        s, _ = _mix_stream(T, n, 5000 + trial, ber=0.0)
        s = s.copy()
        for _ in range(int(rng.integers(0, 12))):              # extra damage on top of the 1 % damaged training sequences
            kind = int(rng.integers(0, 6))
            p = int(rng.integers(0, len(s) - 300))
            if kind == 0:
                s[p] ^= 1
            elif kind == 1:
                s = np.concatenate([s[:p], rng.integers(0, 2, int(rng.integers(1, 40))).astype(np.uint8), s[p:]])
            elif kind == 2:
                s = np.concatenate([s[:p], s[p + int(rng.integers(1, 40)):]])
            elif kind == 3:
                s[p:p + 38] = SEQ_Y
            elif kind == 4:
                s[p:p + 22] = SEQ_N
            else:
                s[p:p + int(rng.integers(1, 2000))] = 0
        s = np.ascontiguousarray(s)
        chunk = int(rng.choice([32, 64, 64, 128]))
        d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
        pa, pb = T.Plan(eng, len(s) // 510 + 8, 1), T.Plan(eng, len(s) // 510 + 8, 1)
        a = T.sync_stream_grid(eng, pa, s, d.data_ptr(), chunk, hs, burst_events=False)
        b = T.sync_stream_grid(eng, pb, s, d.data_ptr(), chunk, hs, burst_events=True)
        evb = [e for e in b["events"] if e[0] != 2]
        assert a["events"] == evb, trial
        for k in ("nslots", "ngrid", "noffgrid", "anchor", "final_state", "burst_seq", "tail_tn_adds"):
            assert a[k] == b[k], (trial, k)
        if a["ngrid"] and not a["noffgrid"]:
            assert (np.asarray(a["grid_bits"]) == np.asarray(b["grid_bits"])).all(), trial
        if n < 1200:
            _, wev = O.run_rx(s, chunk=chunk)
            assert b["events"] == wev, trial
        nloss += sum(1 for e in evb if e[0] in (3, 5))
        pa.close()
        pb.close()
    assert nloss > 300


def test_fuzz_damaged_streams_end_to_end(T, eng):
    """random streams with random damage (bit flips, inserted / deleted bytes, spurious training sequences, zeroed
    stretches, payload noise): channel API (records, events, TDMA time, codes) and the stream-mode plan paths
    (slot table and grid) all equal the oracle's tetra-rx equivalent"""
    import torch
    from test_stream_sync_cpu import SEQ_Y, SEQ_N
    rng = np.random.default_rng(4242)
    hs = torch.cuda.current_stream().cuda_stream
    ngridruns = 0
    for trial in range(14):
        stream, _ = synth.frame_stream(seed=int(rng.integers(1, 1 << 30)), nframes=int(rng.integers(2, 6)),
                                       lead_in=int(rng.integers(0, 600)), pad=int(rng.integers(600, 900)),
                                       ber=float(rng.choice([0.0, 0.0, 0.02])))
        s = stream.copy()
        for _ in range(int(rng.integers(0, 5))):
            kind = int(rng.integers(0, 6))
            p = int(rng.integers(0, len(s) - 60))
            if kind == 0:
                s[p] ^= 1
            elif kind == 1:
                s = np.concatenate([s[:p], rng.integers(0, 2, int(rng.integers(1, 40))).astype(np.uint8), s[p:]])
            elif kind == 2:
                s = np.concatenate([s[:p], s[p + int(rng.integers(1, 40)):]])
            elif kind == 3:
                s[p:p + 38] = SEQ_Y
            elif kind == 4:
                s[p:p + 22] = SEQ_N
            else:
                s[p:p + int(rng.integers(1, 200))] = 0
        s = np.ascontiguousarray(s)
        want, wev = O.run_rx(s)
        ch = T.Channel(eng, batch_slots=int(rng.choice([1, 3, 64])))
        ch.feed(s)
        ch.flush()
        assert_same_records(ch.records, want)
        assert ch.events == wev
        ch.close()
        # stream mode: slot table, then grid (when the stream stays on one grid)
        d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
        res = T.sync_stream(eng, s, d.data_ptr())
        assert res["events"] == wev
        n = len(res["slots"])
        if not n:
            continue
        ch2 = T.Channel(eng, batch_slots=1)
        plan = T.Plan(eng, n, 1)
        plan.load_slots(res, ch2.scramb_init())
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
        torch.cuda.synchronize()
        rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
        ch2.deliver(res["slots"], s, rec)
        assert_same_records(ch2.records, want)
        ch2.close()
        plan.close()
        gplan = T.Plan(eng, len(s) // 510 + 1, 1)
        g = T.sync_stream_grid(eng, gplan, s, d.data_ptr())
        assert g["events"] == wev
        if g["noffgrid"] == 0 and g["ngrid"]:
            ngridruns += 1
            on = T.grid_indices(g)
            assert len(on) == n
            d_rec2 = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            gplan.execute(d.data_ptr(), d_rec2.data_ptr(), hs)
            torch.cuda.synchronize()
            got = d_rec2.cpu().numpy().reshape(-1, T.REC_BYTES)[on]
            pw, pg = T.parse_records(rec), T.parse_records(got)
            for k in pw:
                if k != "slot":
                    assert (np.asarray(pw[k]) == np.asarray(pg[k])).all(), (trial, k)
        gplan.close()
    assert ngridruns >= 3


def test_clean_block_fastpath_is_invisible(T, eng):
    """tgpu_plan_set_fastpath: blocks that are code words skip the trellis, the others do not -- the records are the
    same bytes as without the flag, for clean input, noisy input and a mixture (some slots clean, some with a single
    flipped bit in the payload, in the lead-in bits or in the last half block, some heavily damaged), with a SYNC burst
    switching the scrambling code in between"""
    import torch
    rng = np.random.default_rng(123)
    n = 3000
    ty = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_NORM_2, O.TRAIN_SYNC], n).astype(np.uint8)
    ty[0] = O.TRAIN_SYNC
    slots = T.synth_slots(ty, seed=77, scramb_init=0x41802A07)   # the code the synthetic SYNC PDUs (262 / 42 / 1) announce
    kind = rng.integers(0, 5, n)
    for i in range(n):
        if kind[i] == 1:                      # one flipped payload bit somewhere
            slots[i, int(rng.choice(np.r_[14:230, 282:498]))] ^= 1
        elif kind[i] == 2:                    # heavy noise
            m = rng.random(510) < 0.05
            m[214:266] = False
            slots[i] ^= m.astype(np.uint8)
        elif kind[i] == 3:                    # first / last bits of the blocks
            slots[i, int(rng.choice([14, 15, 16, 229, 282, 283, 496, 497]))] ^= 1
    buf = np.zeros(n * 510 + 64, np.uint8)
    buf[:n * 510] = slots.reshape(-1)
    d = torch.from_numpy(buf).cuda()
    hs = torch.cuda.current_stream().cuda_stream
    recs = []
    for fast in (False, True, True):
        plan = T.Plan(eng, n, 1)
        plan.set_fastpath(fast)
        plan.load(np.arange(n, dtype=np.uint64) * 510, ty)
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        for _ in range(2):                    # twice: the device-side counters must reset
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
        torch.cuda.synchronize()
        recs.append(d_rec.cpu().numpy().reshape(n, T.REC_BYTES).copy())
        plan.close()
    assert (recs[0] == recs[1]).all() and (recs[1] == recs[2]).all()
    p = T.parse_records(recs[1])
    assert 0.3 * n < int(p["crc_ok"][:, 0].sum()) < n


@pytest.mark.parametrize("fast", [False, True])
def test_batch_size_sweep(T, eng, fast, topt):
    """ragged batch sizes around the kernels' tiling (wave = 64 items, workgroups of 256 / 1024, grid-stride loops,
    pipelined front end with its tail): every size decodes like the oracle (the lane-per-trellis kernels at every
    size: the workgroup-per-burst path of small batches is switched off here, it has its own test)"""
    topt("BURST_MAX", int("0"))
    rng = np.random.default_rng(5)
    nmax = 9000
    ty_all = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2], nmax).astype(np.uint8)
    slots_all = T.synth_slots(ty_all, seed=3, scramb_init=0, ber=0.01)
    for n in (1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1023, 1025, 4097, 8193, 9000):
        ty, slots = ty_all[:n], slots_all[:n]
        import torch
        buf = np.zeros(n * 510 + 2, np.uint8)          # 2 bytes of slack only: no read may run far past a slot
        buf[:n * 510] = slots.reshape(-1)
        d = torch.from_numpy(buf).cuda()
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        plan = T.Plan(eng, n, 1)
        plan.set_fastpath(fast)
        plan.load(np.arange(n, dtype=np.uint64) * 510, ty)
        plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
        plan.close()
        check_against_oracle(T, rec, ty, slots, 0)


@pytest.mark.parametrize("ber", [0.0, 0.04, 0.5])
def test_lane_per_slot_kernel_equals_the_lane_per_block_kernels(T, eng, ber, topt):
    """round 6, TGPU_OPT_SLOT: k_slot_t (one lane per SLOT: both blocks of a burst and a SYNC burst's SB1 on one 36-block
    schedule, slot_core.h) against k_vit<216> + k_vit<432> (one lane per BLOCK) on the same loads -- mixed types in every
    arrangement inside a wave (runs of one type, alternating, random), three channels with carry-in codes, SYNC bursts that
    change the code mid-batch, a failed SB1, ignored burst types, clean / noisy / pure-noise payloads, ragged sizes: every byte
    of every record and of every wire record equal, and equal to the oracle's decode"""
    import torch
    topt("BURST_MAX", 0)
    hs = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(int(ber * 100) + 41)
    c1, c2, c3 = O.scramb_get_init(262, 42, 1), O.scramb_get_init(901, 77, 9), O.scramb_get_init(1, 2, 3)
    for n in (1, 63, 64, 65, 200, 4099):
        ty = np.concatenate([np.full(n // 3, O.TRAIN_NORM_1), np.full(n // 3, O.TRAIN_NORM_2),
                             rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_SYNC], n - 2 * (n // 3), p=[0.4, 0.4, 0.2])]).astype(np.uint8)
        if n > 70:
            ty[5::17] = O.TRAIN_NORM_3          # ignored by the path (no record)
            ty[64:128:2], ty[65:128:2] = O.TRAIN_NORM_1, O.TRAIN_NORM_2
        chan = np.sort(rng.integers(0, 3, n)).astype(np.uint32)
        carry = np.array([c1, c2, 0], np.uint32)
        # every channel's cell: its SYNC bursts carry that cell's SYNC PDU, so the code in force follows the oracle's rule
        cell_of = [(262, 42, 1), (901, 77, 9), (1, 2, 3)]
        slots = np.zeros((n, 510), np.uint8)
        for c in range(3):
            m = chan == c
            if m.any():
                mcc, mnc, cc = cell_of[c]
                slots[m] = T.synth_slots(ty[m], seed=100 + c + n, scramb_init=[c1, c2, c3][c], ber=min(ber, 0.2), mcc=mcc, mnc=mnc, cc=cc)
        if ber >= 0.5:
            slots[n // 2:] = rng.integers(0, 2, (n - n // 2, 510)).astype(np.uint8)     # pure noise: ties everywhere
        out = {}
        for mode in (0, 1):
            topt("SLOT", mode)
            d = torch.from_numpy(np.concatenate([slots.reshape(-1), np.zeros(64, np.uint8)])).cuda()
            d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            d_wire = torch.full((n * T.WIRE_BYTES,), 0xff, dtype=torch.uint8, device="cuda")
            plan = T.Plan(eng, n, 3)
            plan.set_wire(d_wire.data_ptr())
            plan.load(np.arange(n, dtype=np.uint64) * 510, ty, chan, carry)
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
            torch.cuda.synchronize()
            out[mode] = (d_rec.cpu().numpy().reshape(n, T.REC_BYTES), d_wire.cpu().numpy().reshape(n, T.WIRE_BYTES), plan.final_codes().tolist())
            plan.close()
        assert (out[0][0] == out[1][0]).all(), n
        assert (out[0][1] == out[1][1]).all(), n
        assert out[0][2] == out[1][2]
        # ... and against the oracle, channel by channel: the code in force at a slot = the latest good SB1 at or before it, else the carry-in
        rec = out[1][0]
        p = T.parse_records(rec)
        dec = np.isin(ty, [O.TRAIN_NORM_1, O.TRAIN_NORM_2, O.TRAIN_SYNC])
        assert (p["type"][dec] == ty[dec]).all()
        assert (rec[~dec][:, 0] == 0xff).all() and (rec[~dec][:, 1:] == 0).all()      # an ignored burst type: the front end's "nothing" mark, no decode
        for i in np.flatnonzero(dec)[:: max(1, n // 300)]:
            ok, want, wcrc = O.bench_decode_slots(slots[i:i + 1], ty[i:i + 1], int(p["code"][i]), want_out=True, want_crc=True)
            t = ty[i]
            assert (p["bbk"][i] == want[0, :14]).all()
            if t == O.TRAIN_NORM_1:
                assert (p["bits1"][i] == want[0, 14:282]).all() and p["crc"][i, 0] == wcrc[0, 0]
            else:
                n1 = 60 if t == O.TRAIN_SYNC else 124
                assert (p["bits1"][i][:n1] == want[0, 14:14 + n1]).all() and (p["bits2"][i] == want[0, 138:262]).all()
                assert p["crc"][i, 0] == wcrc[0, 0] and p["crc"][i, 1] == wcrc[0, 1]


def test_grid_plan_multi_block_scan(T, eng):
    """a longer stream (several 1024-slot blocks in the device-side list building, lock losses in between):
    the grid plan's records == the slot-table plan's"""
    import torch
    rng = np.random.default_rng(17)
    stream, slots = synth.frame_stream(seed=77, nframes=700, lead_in=123, pad=800)
    s = stream.copy()
    p0 = 123 + 510
    for i in rng.choice(len(slots), 40, replace=False):
        off = 214 if slots[int(i)][0] == O.TRAIN_SYNC else 244
        s[p0 + 510 * int(i) + off + 3] ^= 1
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    hs = torch.cuda.current_stream().cuda_stream
    res = T.sync_stream(eng, s, d.data_ptr(), burst_events=False)
    n = len(res["slots"])
    assert n > 5000
    plan = T.Plan(eng, n, 1)
    plan.load_slots(res, 0)
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    want = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    plan.close()
    gplan = T.Plan(eng, len(s) // 510 + 1, 1)
    g = T.sync_stream_grid(eng, gplan, s, d.data_ptr(), burst_events=False)
    assert g["noffgrid"] == 0 and g["nslots"] == n and g["events"] == res["events"]
    on = T.grid_indices(g)
    assert on.tolist() == [(o - g["anchor"]) // 510 for (o, t, q, tn) in res["slots"]]
    d_rec2 = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    gplan.execute(d.data_ptr(), d_rec2.data_ptr(), hs)
    torch.cuda.synchronize()
    got = d_rec2.cpu().numpy().reshape(-1, T.REC_BYTES)[on]
    pw, pg = T.parse_records(want), T.parse_records(got)
    for k in pw:
        if k != "slot":
            assert (np.asarray(pw[k]) == np.asarray(pg[k])).all(), k
    assert int(pw["crc_ok"][:, 0].sum()) > 0.9 * n
    gplan.close()


def test_grid_sync_two_halves_in_flight(T, eng):
    """tgpu_sync_stream_grid_begin/_finish with two streams in flight on their own plans and HIP streams (the
    software pipeline bench.py --workload config3 runs) == the one-call form, outcome and records"""
    import torch
    streams = [_mutated_stream(seed, nframes=6) for seed in (3, 4)]
    devs = [torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda() for s in streams]
    hs = torch.cuda.current_stream().cuda_stream
    want = []
    for s, d in zip(streams, devs):
        p = T.Plan(eng, len(s) // 510 + 1, 1)
        g = T.sync_stream_grid(eng, p, s, d.data_ptr())
        rec = None
        if not g["noffgrid"] and g["ngrid"]:
            r = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            p.execute(d.data_ptr(), r.data_ptr(), hs)
            torch.cuda.synchronize()
            rec = r.cpu().numpy()
        want.append((g, rec))
        p.close()
    side = [torch.cuda.Stream() for _ in streams]
    plans = [T.Plan(eng, len(s) // 510 + 1, 1) for s in streams]
    torch.cuda.synchronize()
    halves = [T.GridSync(eng, p, s, d.data_ptr(), 64, st.cuda_stream) for p, s, d, st in zip(plans, streams, devs, side)]
    for h, p, d, st, (g, rec) in zip(halves, plans, devs, side, want):
        out = h.finish()
        for k in ("events", "anchor", "ngrid", "noffgrid", "nslots"):
            assert out[k] == g[k], k
        if rec is not None:
            assert T.grid_indices(out).tolist() == T.grid_indices(g).tolist()
            r = torch.zeros(out["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            p.execute(d.data_ptr(), r.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            assert (r.cpu().numpy() == rec).all()
        p.close()


# ---------------------------------------------------------------------------
# the remaining puncturers + the speech trellis: k_conv
# ---------------------------------------------------------------------------
def _conv_batch(shape, n, seed):
    """n noisy type-3 blocks of one shape: mixed BER, erased (0xff) and non-binary bytes"""
    L, K, mother, pu = shape
    rng = np.random.default_rng(seed)
    t2 = rng.integers(0, 2, (n, L)).astype(np.uint8)
    t2[::3, -4:] = 0
    t3 = np.stack([O.conv_encode_block(pu, mother, t2[i], K) for i in range(n)])
    ber = rng.choice([0.0, 0.02, 0.06, 0.15, 0.5], n)[:, None]
    t3 ^= (rng.random((n, K)) < ber).astype(np.uint8)
    er = rng.random((n, K)) < np.where(rng.random(n) < 0.2, 0.15, 0.0)[:, None]
    t3[er] = 0xFF
    nb = (rng.random((n, K)) < 0.01) & (t3 == 1)
    t3[nb] = rng.integers(2, 255, int(nb.sum())).astype(np.uint8)
    return t2, t3


@pytest.mark.parametrize("shape", O.PUNCT_SHAPES)
def test_generic_trellis_all_puncturers(T, eng, shape):
    """tgpu_conv_execute == depuncture + Viterbi of the oracle (both restated libosmocore algorithms agree, checked
    on the CPU side) for every (type2, type3, mother code, puncturer) row of tetra_conv_enc.c:257-267; ragged
    batch sizes, a byte-misaligned input/output base, empty batch"""
    import torch
    L, K, mother, pu = shape
    cv = T.ConvDecoder(eng, pu, mother, K, L)
    hs = torch.cuda.current_stream().cuda_stream
    for n, off in ((1, 0), (63, 0), (64, 0), (200, 0), (131, 1)):
        t2, t3 = _conv_batch(shape, n, seed=n * 31 + pu)
        d_in = torch.zeros(n * K + 8, dtype=torch.uint8, device="cuda")
        d_in[off:off + n * K] = torch.from_numpy(t3.reshape(-1)).cuda()
        d_out = torch.full((n * L + 8,), 7, dtype=torch.uint8, device="cuda")
        cv.execute(d_in.data_ptr() + off, n, d_out.data_ptr() + off, hs)
        torch.cuda.synchronize()
        raw = d_out.cpu().numpy()
        got = raw[off:off + n * L].reshape(n, L)
        assert (raw[:off] == 7).all() and (raw[off + n * L:] == 7).all()          # nothing written outside
        for i in range(n):
            want = O.conv_decode_block(pu, mother, t3[i], L, 0)
            assert (got[i] == want).all(), (shape, n, i)
    cv.execute(0, 0, 0, hs)                                                        # empty batch: no launch
    cv.close()


def test_generic_trellis_on_the_control_channel_shapes(T, eng):
    """SCH/F, NDB and SB1 blocks (the shapes the specialised k_vit kernels decode) through k_conv after a host-side
    de-interleave: the type-2 bits of the oracle's block chain, i.e. of block mode"""
    import torch
    rng = np.random.default_rng(12)
    hs = torch.cuda.current_stream().cuda_stream
    for t, pshape in ((O.T_SCH_F, (288, 432, 4, 0)), (O.T_NDB, (144, 216, 4, 0)), (O.T_SB1, (80, 120, 4, 0))):
        K, n2, n1, a = O.BLK[t]
        n = 96
        t5 = np.stack([O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), 0) for _ in range(n)])
        t5 ^= (rng.random((n, K)) < 0.06).astype(np.uint8)
        t3 = np.stack([O.deinterleave(K, a, x) for x in t5])
        cv = T.ConvDecoder(eng, 0, 4, K, n2)
        d_in = torch.from_numpy(t3.reshape(-1)).cuda()
        d_out = torch.zeros(n * n2, dtype=torch.uint8, device="cuda")
        cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().reshape(n, n2)
        for i in range(n):
            assert (got[i] == O.decode_block(t, t5[i], 0)[3]).all()
        cv.close()


def test_generic_trellis_full_size_roundtrip(T, eng):
    """size-independent property at a large batch: encode -> puncture -> k_conv gives the data back (noise-free),
    and with 3 % channel errors every decoded block is a valid trellis path at least as close to the received
    bits as the transmitted one (checked through re-encoding on the host for a sample)"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    for shape in ((292, 432, 4, 2), (72, 162, 3, 5)):
        L, K, mother, pu = shape
        rng = np.random.default_rng(5)
        base = 512
        t2 = rng.integers(0, 2, (base, L)).astype(np.uint8)
        t2[:, -4:] = 0
        t3 = np.stack([O.conv_encode_block(pu, mother, x, K) for x in t2])
        reps = 400                                                                 # 204 800 blocks
        d_in = torch.from_numpy(t3.reshape(-1)).cuda().repeat(reps)
        n = base * reps
        d_out = torch.zeros(n * L, dtype=torch.uint8, device="cuda")
        cv = T.ConvDecoder(eng, pu, mother, K, L)
        cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
        torch.cuda.synchronize()
        got = d_out.view(reps, base, L)
        assert bool((got == torch.from_numpy(t2).cuda()[None]).all())
        flips = (torch.rand(n * K, device="cuda") < 0.03).to(torch.uint8)
        d_noisy = d_in ^ flips
        cv.execute(d_noisy.data_ptr(), n, d_out.data_ptr(), hs)
        torch.cuda.synchronize()
        rx = d_noisy.view(n, K)[:: n // 64].cpu().numpy()
        dec = d_out.view(n, L)[:: n // 64].cpu().numpy()
        for i in range(len(rx)):
            assert (dec[i] == O.conv_decode_block(pu, mother, rx[i], L, 0)).all()
            d_dec = int((O.conv_encode_block(pu, mother, dec[i], K) != rx[i]).sum())
            d_tx = int((t3[(i * (n // 64)) % base] != rx[i]).sum())
            assert d_dec <= d_tx
        cv.close()


@pytest.mark.parametrize("L,K,mother,pu", [(8, 12, 4, 0), (8, 24, 4, 1), (8, 12, 3, 4)])
def test_generic_trellis_is_maximum_likelihood_with_the_stated_tie_rule_exhaustively(T, eng, L, K, mother, pu):
    """every received word of a short block through k_conv (tgpu_conv_execute), erasures included: the brute-force
    minimum-distance sequence under the stated tie rule (tests/ml_exhaustive.py), which is also the oracle's answer
    (test_oracle_props.py does the same on the CPU).  One launch per erasure pattern, up to 4096 blocks each."""
    import torch
    import ml_exhaustive as ML
    # (the third case is the rate-1/3 speech code under its 8/12 puncturer, tetra_conv_enc.c:164-171; round 5 had the 1/3
    # puncturer of the OTHER mother code there -- a shape neither side accepts, so the case always skipped and read as coverage)
    cv = T.ConvDecoder(eng, pu, mother, K, L)
    assert O.conv_decode_block(pu, mother, np.zeros(K, np.uint8), L, 0) is not None
    hs = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(L * 100 + K + mother)
    xs, cb = ML.codebook(L, K, mother, pu)
    words = ((np.arange(1 << K)[:, None] >> np.arange(K)[None, :]) & 1).astype(np.uint8) if K <= 12 else \
        rng.integers(0, 2, (4096, K)).astype(np.uint8)
    tied = total = 0
    for pat in ML.erasure_patterns(K, rng):
        rx = words.copy()
        rx[:, pat] = 0xff
        rx = np.unique(rx, axis=0)
        want, dmin, nties = ML.ml_decode(xs, cb, rx)
        n = len(rx)
        d_in = torch.from_numpy(rx.reshape(-1)).cuda()
        d_out = torch.full((n * L,), 9, dtype=torch.uint8, device="cuda")
        cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().reshape(n, L)
        bad = np.flatnonzero((got != want).any(1))
        assert not len(bad), (pat.nonzero()[0].tolist(), rx[bad[0]].tolist(), got[bad[0]].tolist(), want[bad[0]].tolist(), int(nties[bad[0]]))
        tied += int((nties > 1).sum())
        total += n
    assert tied > total // 8
    cv.close()


def test_generic_trellis_rejects_bad_shapes(T, eng):
    for args in ((7, 4, 120, 80), (0, 5, 120, 80), (0, 4, 432, 80), (0, 4, 15, 10), (0, 4, 1200, 800)):
        with pytest.raises(T.TgpuError):
            T.ConvDecoder(eng, *args)


@pytest.mark.parametrize("shape", [(504, 756, 4, 0), (344, 1022, 4, 1), (252, 378, 3, 4), (12, 18, 4, 0)])
def test_generic_trellis_extreme_shapes(T, eng, shape):
    """the largest block the decoder accepts (63 history blocks, 8 register chunks), the longest type-3 row with a
    row length that is not a multiple of 4 (raw staging copy, > 64 KB of LDS), a long speech-code block and the
    smallest block with a partial history block -- against the oracle"""
    import torch
    L, K, mother, pu = shape
    cv = T.ConvDecoder(eng, pu, mother, K, L)
    hs = torch.cuda.current_stream().cuda_stream
    n = 150
    t2, t3 = _conv_batch(shape, n, seed=L + K)
    d_in = torch.from_numpy(t3.reshape(-1)).cuda()
    d_out = torch.zeros(n * L, dtype=torch.uint8, device="cuda")
    cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(n, L)
    for i in range(n):
        assert (got[i] == O.conv_decode_block(pu, mother, t3[i], L, 0)).all(), (shape, i)
    cv.close()


def test_plan_and_conv_execute_are_graph_capturable(T, eng):
    """tgpu_plan_execute / tgpu_conv_execute only launch (header contract): captured into a HIP graph on a side
    stream and replayed, they produce the records / bits of a direct call"""
    import torch
    n = 3000
    rng = np.random.default_rng(77)
    types = rng.choice([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2], n).astype(np.uint8)
    code = O.scramb_get_init(262, 42, 1)
    slots = T.synth_slots(types, seed=3, scramb_init=code, ber=0.03)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * T.SLOT_BYTES, types, None, np.array([code], np.uint32))
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    shape = (292, 432, 4, 2)
    t2, t3 = _conv_batch(shape, 500, seed=5)
    cv = T.ConvDecoder(eng, shape[3], shape[2], shape[1], shape[0])
    d_t3 = torch.from_numpy(t3.reshape(-1)).cuda()
    d_t2 = torch.zeros(500 * shape[0], dtype=torch.uint8, device="cuda")
    hs = torch.cuda.current_stream().cuda_stream
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), hs)
    cv.execute(d_t3.data_ptr(), 500, d_t2.data_ptr(), hs)
    torch.cuda.synchronize()
    want_rec, want_t2 = d_rec.clone(), d_t2.clone()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        cs = torch.cuda.current_stream().cuda_stream
        plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), cs)
        cv.execute(d_t3.data_ptr(), 500, d_t2.data_ptr(), cs)
    for _ in range(2):
        d_rec.zero_()
        d_t2.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert bool((d_rec == want_rec).all()) and bool((d_t2 == want_t2).all())
    cv.close()
    plan.close()


def test_front_end_pipeline_tails(T, eng, topt):
    """k_front with very few waves (TGPU_FRONT_BLOCKS, a test knob of the launcher), so that every wave runs its
    prologue, the unconditional main loop and the checked tail over sequences of every length: groups of four
    slots, a short last group, fewer slots than waves -- every batch size from 1 to 150 and a few larger ones"""
    topt("BURST_MAX", int("0"))        # (small batches would bypass k_front)
    import torch
    import emul
    rng = np.random.default_rng(15)
    nmax = 700
    ty_all = rng.choice([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2], nmax).astype(np.uint8)
    slots_all = T.synth_slots(ty_all, seed=9, scramb_init=0)
    hs = torch.cuda.current_stream().cuda_stream
    want = [emul.pack_slot(int(ty_all[i]), slots_all[i])[:19] for i in range(nmax)]
    for blocks in ("1", "2"):
        topt("FRONT_BLOCKS", int(blocks))
        for n in list(range(1, 151)) + [255, 256, 257, 511, 700]:
            d = torch.from_numpy(slots_all[:n].reshape(-1).copy()).cuda()
            plan = T.Plan(eng, n, 1)
            plan.load(np.arange(n, dtype=np.uint64) * 510, ty_all[:n])
            d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
            torch.cuda.synchronize()
            got = plan.read_packed()
            plan.close()
            for i in range(n):
                assert (got[i, :19] == want[i]).all(), (blocks, n, i)


def test_generic_trellis_random_shapes(T, eng):
    """random valid (puncturer, mother code, type-2 length, type-3 length) shapes beyond the reference's nine rows:
    every history-chunk count, partial tail blocks, type-3 lengths of every residue mod 4 -- against the oracle"""
    import torch
    rng = np.random.default_rng(2024)
    hs = torch.cuda.current_stream().cuda_stream
    done = 0
    while done < 40:
        pu = int(rng.integers(0, 7))
        mother = 3 if pu >= 4 else 4
        L = int(rng.integers(1, 63)) * 8 + int(rng.choice([0, 0, 4, 5, 6, 7]))
        if L > 504:
            continue
        kmax = 1
        while kmax < 1022 and O.conv_decode_block(pu, mother, np.zeros(kmax + 1, np.uint8), L, 0) is not None:
            kmax += 1                                   # longest type-3 length whose positions fit the mother buffer
        K = int(rng.integers(max(1, kmax - 6), kmax + 1))
        if O.conv_decode_block(pu, mother, np.zeros(K, np.uint8), L, 0) is None:
            continue
        n = int(rng.integers(1, 130))
        t2 = rng.integers(0, 2, (n, L)).astype(np.uint8)
        mlen = L * mother
        t3 = np.zeros((n, K), np.uint8)
        for i in range(n):
            m = O.conv_encode_tch(t2[i]) if mother == 3 else O.conv_encode(t2[i])
            t3[i] = O.puncture(pu, m, K)
        t3 ^= (rng.random((n, K)) < 0.05).astype(np.uint8)
        t3[rng.random((n, K)) < 0.02] = 0xFF
        cv = T.ConvDecoder(eng, pu, mother, K, L)
        d_in = torch.from_numpy(t3.reshape(-1)).cuda()
        d_out = torch.zeros(n * L, dtype=torch.uint8, device="cuda")
        cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().reshape(n, L)
        for i in range(n):
            assert (got[i] == O.conv_decode_block(pu, mother, t3[i], L, 0)).all(), (pu, mother, L, K, n, i)
        cv.close()
        done += 1


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_acelp_reordering_on_device(T, eng, seed):
    """tgpu_reorder_* with the maps of tgpu_acelp_build_map() == the oracle's restatement of
    lower_mac/tch_reordering.c:94-140 on batches of blocks, both directions, tables that are permutations and
    tables damaged like the reference's own; with the real object's tables where oracle/_ref is there"""
    import torch
    from test_host_logic import _acelp_tables
    cases = [_acelp_tables(seed)]
    if seed == 0 and O.ref_acelp_tables() is not None:
        cases.append((O.ref_acelp_tables(), 137))
    rng = np.random.default_rng(50 + seed)
    for cls, nbits in cases:
        nb = int(rng.integers(1000, 5000))
        x = rng.integers(0, 2, (nb, 2 * nbits)).astype(np.uint8)
        d_in = torch.from_numpy(x).cuda()
        for to_codec in (True, False):
            r = T.Reorder(eng, T.acelp_build_map(cls, to_codec))
            d_out = torch.full((nb, 2 * nbits), 7, dtype=torch.uint8, device="cuda")
            r.execute(d_in.data_ptr(), nb, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            f = O.acelp_type2_to_codec if to_codec else O.acelp_codec_to_acelp
            for i in rng.choice(nb, 40, replace=False):
                assert got[i].tolist() == f(x[i], cls, fill=7).tolist()
            r.close()


def _reference_stdout_lines(records):
    """what tetra-rx prints for these tp_sap_udata_ind() calls (lower_mac/tetra_lower_mac.c:258-266): 'CRC COMP' line
    per CRC-protected block, and the type-1 bits of the good ones with the block's name and TDMA time"""
    names = {O.T_SB1: "SB1", O.T_SB2: "SB2", O.T_NDB: "NDB", O.T_SCH_HU: "SCH/HU", O.T_SCH_F: "SCH/F"}
    out = []
    for r in records:
        if r["type"] == O.T_BBK or r["traffic"]:
            continue
        if r["crc_ok"]:
            tn, fn, mn = r["time_str"]
            out.append("CRC COMP: 0x%04x OK" % r["crc"])
            out.append("%s %02u/%02u/%u/%03u type1: %s" % (names[r["type"]], mn, fn, tn, 0, "".join(str(b) for b in r["type1"])))
        else:
            out.append("CRC COMP: 0x%04x WRONG" % r["crc"])
    return out


@pytest.mark.parametrize("seam,with_ref_obj", [("sync_in", False), ("rx_cb", False), ("rx_cb", True)])
def test_c_consumer_links_and_prints_reference_observables(T, eng, seam, with_ref_obj, tmp_path):
    """a plain C translation unit against include/tetra_gpu.h, linked with -ltetra_gpu (tools/tetra_rx_gpu.c), fed a
    capture: its stdout == the lines tetra-rx would print for the oracle's decode of the same capture ('CRC COMP: ...
    OK|WRONG', '<NAME> <time> type1: <bits>'), through tetra_burst_sync_in() and through the reference-signature
    tetra_burst_rx_cb() / tp_sap_udata_ind() seam -- the latter also with the reference's own phy/tetra_burst.o
    linked in front of the library (its tetra_burst_rx_cb() then calls this library's tp_sap_udata_ind())"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler on this box")
    obj = os.path.join(root, "oracle", "_ref", "tetra_burst.o")
    if with_ref_obj and not os.path.exists(obj):
        pytest.skip("oracle/_ref/tetra_burst.o not built (needs /root/reference)")
    exe = str(tmp_path / "tetra_rx_gpu")
    libdir = os.path.join(root, "osmo-tetra_amd")
    cmd = [gcc, "-O2", "-Wall", os.path.join(root, "tools", "tetra_rx_gpu.c")] + ([obj] if with_ref_obj else []) + \
        ["-I" + os.path.join(root, "include"), "-L" + libdir, "-ltetra_gpu", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.check_call(cmd)
    stream, _ = synth.frame_stream(seed=31, nframes=10, ber=0.035)
    s = stream.copy()
    s[100 + 510 + 510 * 9 + 244 + 7] ^= 1            # a loss of lock and a re-lock in between
    cap = tmp_path / "capture.bits"
    s.tofile(cap)
    want, wev = O.run_rx(s)
    r = subprocess.run([exe, str(cap), "--seam", seam, "--batch", "7"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith(("CRC COMP", "SB", "NDB", "SCH"))]
    assert lines == _reference_stdout_lines(want)
    nok = sum(1 for x in want if x["crc_ok"] and x["type"] != O.T_BBK)
    assert nok >= 30 and any(not x["crc_ok"] for x in want)
    assert "%d CRC OK" % nok in r.stderr
    if with_ref_obj:      # the reference's burst builders print at load... no: its tetra_burst_rx_cb ran -- nothing else to see
        assert "found SYNC training sequence in bit #" in r.stdout
    assert r.stderr.count("####") == sum(1 for e in wev if e[0] in (3, 4, 5))


def _mix_stream(T, n, seed, code_cell=(262, 42, 1), ber=0.02):
    """BASELINE config 3 at size n: lead-in, lock-only SB, frames [SB,N1,N2,N1,N2,N1,N2,N1], 1 % of the slots with a
    damaged training sequence, payload noise"""
    rng = np.random.default_rng(seed)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    mcc, mnc, cc = code_cell
    code = O.scramb_get_init(mcc, mnc, cc)
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=seed, scramb_init=code, mcc=mcc, mnc=mnc, cc=cc, ber=ber)
    y = slots[0, 214:252].tolist()
    for i in np.flatnonzero(rng.random(n) < 0.01) + 1:
        off = 214 if slots[i, 214:252].tolist() == y else 244
        slots[i, off + int(rng.integers(0, 22))] ^= 1
    return np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)]), code


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("ber,per", [(0.0, 3000), (0.03, 9000)])
def test_front_end_and_trellis_in_one_launch_equal_the_separate_kernels(T, eng, ber, per, mode, topt):
    """round 6, TGPU_OPT_SLOT 2: k_slot -- a wave packs and classifies 64 neighbouring grid slots, keeps them in LDS and decodes them
    there, on a HINTED scrambling code (the carry-in, else what the plan's last batch of the channel ended with); after the walk
    every delivered slot whose code in force differs, or that the exact pass settled, goes through k_slot_t.  Against the same
    batches with the option at 1 (front end, then k_slot_t on the exact lists): events, bitmaps, counts, final codes, and every
    byte of every DELIVERED slot's record and wire record -- for a plan's first batch (no hint: not fused), its second (hints from
    the first), carry-in codes given (fused at once), WRONG hints (another cell's codes: everything is decoded twice), a channel
    whose cell changes in mid-recording, channels that start inside a wave's 64 slots (grids are padded to 32), 1 % damaged
    training sequences, payload noise.  mode 3 (TGPU_OPT_SLOT 3): the same hints and the same look-back check, the trellises by
    k_slot_e on a stream of the plan's own beside the walk (and in line, with the plan's side streams off)"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (505, 1, 60)]
    lens = [per, per // 2 + 37, 96, per, per // 3]          # (96 slots: the next channel starts 32 slots into a wave's task)
    streams, codes = [], []
    for c, cell in enumerate(cells):
        st, code = _mix_stream(T, lens[c], 7000 + c, cell, ber=ber)
        if c == 3:      # a recording whose cell changes half way: the second half carries another cell's SYNC PDUs and code
            st2, _ = _mix_stream(T, lens[c], 7100, (208, 10, 5), ber=ber)
            cut = 100 + 510 * (1 + lens[c] // 2)
            st = np.concatenate([st[:cut], st2[cut:]])
        streams.append(st)
        codes.append(code)
    offs, o = [], 0
    for st in streams:
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    d = torch.from_numpy(buf).cuda()
    ntot = sum((len(st) // 510 + 32) for st in streams)

    def batch(plan, carry=None, sts=None):
        d_rec = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        d_wire = torch.full((ntot * T.WIRE_BYTES,), 0xff, dtype=torch.uint8, device="cuda")
        plan.set_wire(d_wire.data_ptr())
        msd = T.MultiSyncDev(eng, plan, sts or streams, d.data_ptr(), offs, d_rec.data_ptr(), 64, hs, codes=carry)
        fused = msd.fused
        outs = msd.collect()
        assert not msd.fellback
        torch.cuda.synchronize()
        return (fused, outs, d_rec.cpu().numpy().reshape(-1, T.REC_BYTES), d_wire.cpu().numpy().reshape(-1, T.WIRE_BYTES), plan.final_codes().tolist())

    def same(a, b):
        assert len(a[1]) == len(b[1])
        n = 0
        for c, (x, y) in enumerate(zip(a[1], b[1])):
            assert x["events"] == y["events"], c
            for k in ("nslots", "ngrid", "noffgrid", "anchor", "final_state", "burst_seq", "tail_tn_adds", "grid_base"):
                assert x[k] == y[k], (c, k)
            assert (np.asarray(x["grid_bits"]) == np.asarray(y["grid_bits"])).all()
            idx = x["grid_base"] + T.grid_indices(x)
            assert (a[2][idx] == b[2][idx]).all(), c
            assert (a[3][idx] == b[3][idx]).all(), c
            n += len(idx)
        assert a[4] == b[4]
        return n

    topt("SLOT", 1)
    pa = T.Plan(eng, ntot, len(cells))
    ref0 = batch(pa)                                    # carry-in codes 0
    ref1 = batch(pa, carry=codes)                       # carry-in codes = the cells'
    assert not ref0[0] and not ref1[0]
    pa.close()
    topt("SLOT", mode)
    pb = T.Plan(eng, ntot, len(cells))
    b1 = batch(pb)
    assert not b1[0]                                    # nothing to decode on yet
    assert same(ref0, b1) > 0.9 * sum(lens)
    b2 = batch(pb)
    assert b2[0]                                        # hints: the codes the first batch ended with
    same(ref0, b2)
    b3 = batch(pb, carry=codes)
    assert b3[0] == (1 if mode == 2 else 2)
    same(ref1, b3)
    pb.set_side_stream(False)                           # everything of a batch on the caller's stream (several batches in flight)
    b4 = batch(pb)
    assert b4[0]
    same(ref0, b4)
    pb.close()
    pc = T.Plan(eng, ntot, len(cells))
    c1 = batch(pc, carry=codes)                         # a plan's first batch with carry-in codes: fused at once
    assert c1[0]
    same(ref1, c1)
    wrong = [codes[(c + 1) % len(codes)] for c in range(len(codes))]
    c2 = batch(pc, carry=wrong)                         # every channel decoded on another cell's code first
    refw = None
    topt("SLOT", 1)
    pd = T.Plan(eng, ntot, len(cells))
    refw = batch(pd, carry=wrong)
    pd.close()
    assert c2[0] and not refw[0]
    same(refw, c2)
    pc.close()
    # ... and the reference itself against the oracle: every delivered burst of the two-cell channel under the code in force
    out = ref0[1][3]
    idx = T.grid_indices(out)
    r = ref0[2][out["grid_base"] + idx]
    st = streams[3]
    slots = st[out["anchor"]:out["anchor"] + 510 * (int(idx[-1]) + 1)].reshape(-1, 510)[idx]
    p = T.parse_records(r)
    assert len(set(p["code"].tolist())) == 2            # the two cells' codes (the first delivered burst is a SYNC burst: its own SB1 sets the code)
    for i in range(0, len(idx), max(1, len(idx) // 400)):
        ty = p["type"][i:i + 1].astype(np.uint8)
        ok, want, wcrc = O.bench_decode_slots(slots[i:i + 1], ty, int(p["code"][i]), use_acc=1, want_out=True, want_crc=True)
        assert (p["bbk"][i] == want[0, :14]).all() and p["crc"][i, 0] == wcrc[0, 0]


def test_config3_full_size_stream_against_the_oracle(T, eng):
    """BASELINE config 3 at 100 000 slots through the GPU front end + grid plan: the synchroniser's events == the
    oracle receiver's (every lock loss and re-lock of ~1000 damaged slots), every delivered burst's type-1 bits,
    BBK and CRC words == the oracle's decode, the number of tp_sap_udata_ind() deliveries matches"""
    import torch
    n = 100_000
    s, code = _mix_stream(T, n, 77)
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    hs = torch.cuda.current_stream().cuda_stream
    plan = T.Plan(eng, n + 8, 1)
    g = T.sync_stream_grid(eng, plan, s, d.data_ptr(), 64, hs, burst_events=False)
    assert g["noffgrid"] == 0 and 0.9 * n < g["nslots"] < n
    d_rec = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    idx = T.grid_indices(g)
    rec = d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)[idx]
    # the oracle's receiver on the same bytes: events and delivery count
    nrec = [0]
    want_ev = []
    import ctypes as C
    rx = O.Rx()
    ucb = O.UPPER_CB(lambda rxp, recp, off, priv: (nrec.__setitem__(0, nrec[0] + (off in (0, 0xFFFFFFFF))), -1)[1])
    ecb = O.EVENT_CB(lambda ev, bitnum, arg, priv: want_ev.append((ev, bitnum, arg)) if ev != 2 else None)
    O.lib().orc_rx_init(C.byref(rx), ucb, ecb, None)
    rx.use_acc = 1
    O.lib().orc_rx_feed(C.byref(rx), O._p(s), len(s), 64)
    assert g["events"] == want_ev and len(want_ev) > 1500
    p = T.parse_records(rec)
    ty = p["type"].astype(np.uint8)
    assert nrec[0] == int(3 * ((ty == 3).sum() + (ty == 1).sum()) + 2 * (ty == 0).sum()) and rx.burst_seq >= len(idx)
    slots = np.stack([s[g["anchor"] + 510 * int(k):g["anchor"] + 510 * int(k) + 510] for k in idx])
    ok, p2 = check_against_oracle(T, rec, ty, slots, code)
    assert 0.3 * nrec[0] < ok < nrec[0]
    plan.close()


def test_config5_full_size_soft_decode_against_the_oracle(T, eng):
    """BASELINE config 5 at 100 000 slots: float phases (sigma 0.7) -> device float_to_bits (hard bits == the
    reference slicer's, soft values == the oracle's) -> soft-decision decode == the oracle's soft chain on every block"""
    import torch
    n = 100_000
    rng = np.random.default_rng(55)
    code = O.scramb_get_init(262, 42, 1)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    slots = T.synth_slots(types, seed=41, scramb_init=code)
    bits = slots.reshape(-1)
    phi = (O.bits_to_phase(bits) + rng.normal(0, 0.7, len(bits) // 2)).astype(np.float32)
    d_phi = torch.from_numpy(phi).cuda()
    d_bits = torch.zeros(len(bits) + 64, dtype=torch.uint8, device="cuda")
    d_soft = torch.zeros(len(bits) + 64, dtype=torch.int8, device="cuda")
    eng.float_to_bits(d_phi.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr())
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([code], np.uint32))
    plan.execute_soft(d_soft.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_bits.cpu().numpy()[:len(bits)] == O.float_to_bits(phi)).all()
    soft = d_soft.cpu().numpy()[:len(bits)]
    assert (soft == O.float_to_soft(phi)).all()
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    p = T.parse_records(rec)
    ok, want, wcrc = O.bench_decode_slots_soft(soft.reshape(n, 510), types, code)
    n1, n2, sb = types == 0, types == 1, types == 3
    assert (p["bbk"] == want[:, :14]).all()
    assert (p["bits1"][n1] == want[n1, 14:282]).all()
    assert (p["bits1"][n2][:, :124] == want[n2, 14:138]).all() and (p["bits2"][n2] == want[n2, 138:262]).all()
    assert (p["bits1"][sb][:, :60] == want[sb, 14:74]).all() and (p["bits2"][sb] == want[sb, 138:262]).all()
    assert (p["crc"][:, 0] == wcrc[:, 0]).all() and (p["crc"][n2 | sb, 1] == wcrc[n2 | sb, 1]).all()
    nblk = int(2 * (n2 | sb).sum() + n1.sum())
    assert int(p["crc_ok"][:, 0].sum() + p["crc_ok"][n2 | sb, 1].sum()) == ok and 0.3 * nblk < ok < nblk
    # the fused path (slicer inside the gather kernel, no bit / soft stream in memory): the same records
    d_rec2 = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute_float(d_phi.data_ptr(), len(phi), d_rec2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_rec, d_rec2)
    plan.close()


def test_float_input_two_channels_codes_learnt_mid_batch(T, eng):
    """tgpu_plan_execute_float on a two-channel batch whose scrambling codes are only learnt from SB1 inside the batch
    (carry-in code 3 / a wrong one): records == float_to_bits + execute_soft, blocks behind a good SYNC slot pass
    their CRC with the learnt code, final codes per channel"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9)]
    n = 4000
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    chan = (np.arange(n) >= n // 2).astype(np.uint32)
    parts = []
    for c, cell in enumerate(cells):
        m = int((chan == c).sum())
        parts.append(T.synth_slots(types[chan == c], seed=70 + c, scramb_init=O.scramb_get_init(*cell), mcc=cell[0], mnc=cell[1], cc=cell[2]))
    slots = np.concatenate(parts)
    rng = np.random.default_rng(8)
    phi = (O.bits_to_phase(slots.reshape(-1)) + rng.normal(0, 0.45, slots.size // 2)).astype(np.float32)
    d_phi = torch.from_numpy(phi).cuda()
    d_bits = torch.zeros(2 * len(phi) + 64, dtype=torch.uint8, device="cuda")
    d_soft = torch.zeros(2 * len(phi) + 64, dtype=torch.int8, device="cuda")
    eng.float_to_bits(d_phi.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr())
    plan = T.Plan(eng, n, 2)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, chan, np.array([3, 0x1234567], np.uint32))
    a = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    b = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute_soft(d_soft.data_ptr(), a.data_ptr(), hs)
    plan.execute_float(d_phi.data_ptr(), len(phi), b.data_ptr(), hs)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    p = T.parse_records(b.cpu().numpy().reshape(n, T.REC_BYTES))
    want = np.array([O.scramb_get_init(*cells[int(c)]) for c in chan], np.uint32)
    assert (p["code"][types != 3] == want[types != 3]).all()          # slot 0 of each channel is a SYNC slot
    assert p["crc_ok"][types == 0, 0].mean() > 0.9
    assert plan.final_codes().tolist() == [O.scramb_get_init(*cells[0]), O.scramb_get_init(*cells[1])]
    plan.close()


def test_float_input_odd_offsets_and_stream_end(T, eng):
    """tgpu_plan_execute_float: slots at odd stream positions (a burst that starts on the second value of a symbol),
    ignored burst types in between, NaN / infinite / huge phases, and a last slot that ends exactly with the input
    (the 256th symbol of its window does not exist) -- records == float_to_bits + execute_soft on the same stream"""
    import torch
    rng = np.random.default_rng(91)
    hs = torch.cuda.current_stream().cuda_stream
    for trial in range(6):
        n = int(rng.integers(1, 70))
        types = rng.choice(np.array([0, 1, 3, 2, 4], np.uint8), n, p=[0.3, 0.3, 0.2, 0.1, 0.1]).astype(np.uint8)
        code = int(rng.integers(0, 1 << 32))
        slots = T.synth_slots(np.where((types == 2) | (types > 3), 0, types).astype(np.uint8), seed=100 + trial, scramb_init=code)
        gaps = rng.integers(0, 40, n)                      # arbitrary (odd and even) gaps in front of every slot
        if trial == 0:
            gaps[:] = 0
        offs, parts, o = [], [], 0
        for i in range(n):
            parts.append(rng.integers(0, 2, gaps[i]).astype(np.uint8)); o += int(gaps[i])
            offs.append(o); parts.append(slots[i]); o += 510
        bits = np.concatenate(parts)
        if len(bits) & 1:
            bits = np.concatenate([bits, np.zeros(1, np.uint8)])
        phi = (O.bits_to_phase(bits) + rng.normal(0, 0.5, len(bits) // 2)).astype(np.float32)
        k = rng.integers(0, len(phi), 12)
        phi[k[:4]] = np.nan; phi[k[4:8]] = np.inf; phi[k[8:10]] = -np.inf; phi[k[10:]] = 3e38
        d_phi = torch.from_numpy(phi).cuda()
        d_bits = torch.zeros(2 * len(phi) + 64, dtype=torch.uint8, device="cuda")
        d_soft = torch.zeros(2 * len(phi) + 64, dtype=torch.int8, device="cuda")
        eng.float_to_bits(d_phi.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr())
        plan = T.Plan(eng, n, 1)
        plan.load(np.array(offs, np.uint64), types, None, np.array([code], np.uint32))
        a = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        b = torch.full((n * T.REC_BYTES,), 0, dtype=torch.uint8, device="cuda")
        plan.execute_soft(d_soft.data_ptr(), a.data_ptr(), hs)
        plan.execute_float(d_phi.data_ptr(), len(phi), b.data_ptr(), hs)
        torch.cuda.synchronize()
        assert torch.equal(a, b), trial
        plan.close()


def test_config4_channels_in_one_grid_batch(T, eng):
    """BASELINE config 4's per-GPU share as ONE batch (tgpu_sync_multi_*): eight recorded channels of different cells,
    lengths, lead-ins and damage in one device buffer -> one classification launch, one walk per channel (host
    threads), one multi-channel plan.  Every channel's events and delivered bursts == the oracle receiver's on that
    channel alone, records == the single-channel grid path's, final codes per channel"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (262, 42, 2), (505, 1, 60), (208, 10, 5), (222, 99, 7)]
    streams, offs, o = [], [], 0
    rng = np.random.default_rng(404)
    for c, cell in enumerate(cells):
        nsl = int(rng.integers(40, 1500)) if c != 3 else 2          # (one channel with next to nothing in it)
        st, _ = _mix_stream(T, nsl, 900 + c, cell, ber=0.02)
        if c == 5:
            st = st[:len(st) - 700 - 200]                            # a channel that ends in the middle of a burst
        streams.append(st)
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    d = torch.from_numpy(buf).cuda()
    ntot = sum((len(st) // 510 + 32) for st in streams)
    plan = T.Plan(eng, ntot, len(cells))
    ms = T.MultiSync(eng, plan, streams, d.data_ptr(), offs, 64, hs)
    outs = ms.finish(burst_events=True, nthreads=3)
    d_rec = torch.zeros(max(ms.ngrid, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)
    codes = plan.final_codes()
    ndeliv = 0
    for c, (st, out) in enumerate(zip(streams, outs)):
        _, wev = O.run_rx(st)
        assert out["events"] == wev, c
        assert out["noffgrid"] == 0
        # the same channel alone through the single-channel grid path
        p1 = T.Plan(eng, len(st) // 510 + 8, 1)
        dd = torch.from_numpy(np.concatenate([st, np.zeros(T.STREAM_SLACK + 2304, np.uint8)])).cuda()
        g = T.sync_stream_grid(eng, p1, st, dd.data_ptr(), 64, hs)
        assert g["events"] == wev and g["nslots"] == out["nslots"] and g["anchor"] == out["anchor"]
        if g["ngrid"]:
            r1 = torch.zeros(g["ngrid"] * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            p1.execute(dd.data_ptr(), r1.data_ptr(), hs)
            torch.cuda.synchronize()
            idx = T.grid_indices(g)
            assert idx.tolist() == T.grid_indices(out).tolist()
            a = r1.cpu().numpy().reshape(-1, T.REC_BYTES)[idx]
            b = rec[out["grid_base"] + idx]
            pa, pb = T.parse_records(a), T.parse_records(b)
            for k in pa:
                if k != "slot":
                    assert (np.asarray(pa[k]) == np.asarray(pb[k])).all(), (c, k)
            assert (pb["slot"] == out["grid_base"] + idx).all()
            assert int(codes[c]) == int(p1.final_codes()[0])
            ndeliv += len(idx)
        p1.close()
    assert ndeliv > 3000
    plan.close()


def _multi_batch(T, streams):
    import torch
    offs, o = [], 0
    for st in streams:
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    return torch.from_numpy(buf).cuda(), offs, sum((len(st) // 510 + 32) for st in streams)


def _same_batch_outcome(T, a, b, rec_a, rec_b, what):
    for c, (x, y) in enumerate(zip(a, b)):
        assert x["events"] == y["events"], (what, c)
        for k in ("nslots", "ngrid", "noffgrid", "anchor", "final_state", "burst_seq", "tail_tn_adds", "grid_base"):
            assert x[k] == y[k], (what, c, k, x[k], y[k])
        if x["ngrid"] and not x["noffgrid"]:
            assert (np.asarray(x["grid_bits"]) == np.asarray(y["grid_bits"])).all(), (what, c)
            idx = x["grid_base"] + T.grid_indices(x)
            assert (rec_a[idx] == rec_b[idx]).all(), (what, c)


@pytest.mark.parametrize("chunk,mono,wide", [(64, 0, 0), (32, 0, 0), (64, 1, 0), (64, 0, 1), (128, 0, 0), (256, 0, 0)])
def test_device_walk_batch_equals_host_walk_batch(T, eng, chunk, mono, wide, topt):
    """tgpu_sync_multi_launch / _collect (the synchroniser walks on the device: k_walk) against tgpu_sync_multi_begin /
    _finish (host walks) on the same multi-channel batch: eight channels of different cells, lengths, lead-ins and damage
    -- damaged training sequences in runs, right behind SYNC bursts (the one-call backlog) and in the last slots, spurious
    sequences below offset 21 (the reference's skewed look-ahead rule, now evaluated by the kernels), a channel with next
    to nothing in it, one that ends inside a burst.  Per channel: events, counts, final state, delivered bitmap, every
    delivered record byte, final codes.  No fallback on these -- neither in the plan's first batch (node arrays laid out
    for the full cap) nor in its second (laid out for twice the nodes the first one showed); then once more with the
    fallback forced (same results)"""
    import torch
    from test_stream_sync_cpu import SEQ_N, SEQ_P
    topt("WALK_MONO", mono)         # (the walk as one launch per form instead of three: same outcome)
    topt("WALK_WIDE", wide)         # (the per-channel launches as 1024 threads / 128 KB of LDS: same outcome)
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (262, 42, 2), (505, 1, 60), (208, 10, 5), (222, 99, 7)]
    rng = np.random.default_rng(4040 + chunk)
    streams = []
    for c, cell in enumerate(cells):
        nsl = int(rng.integers(200, 3000)) if c != 3 else 2
        st, _ = _mix_stream(T, nsl, 1900 + c, cell, ber=0.02)
        st = st.copy()
        lead = 100 + 510            # (_mix_stream: 100 lead-in bits, then the lock-only SYNC burst)
        tr = [lead + 510 * i + (214 if i % 8 == 0 else 244) for i in range(nsl)]
        for q, i in enumerate(tr):
            if rng.random() < 0.03:
                st[i + int(rng.integers(0, 22))] ^= 1
        if nsl > 100:
            for j in (17, 18, 19, 41, 43, nsl - 2, nsl - 1, 64, 72, 73):      # runs, slots next to SYNC bursts (multiples of 8), the end
                st[tr[j] + 2] ^= 1
            for j in (30, 55, 77):                                                # sequences below offset 21 of a slot
                seq = (SEQ_N, SEQ_P)[j % 2]           # (a spurious SYNC sequence pulls the walk off the grid: next test)
                o = lead + 510 * j + int(rng.integers(0, 21))
                st[o:o + len(seq)] = seq
        if c == 5:
            st = st[:len(st) - 700 - 200]
        streams.append(np.ascontiguousarray(st))
    d, offs, ntot = _multi_batch(T, streams)
    pa, pb = T.Plan(eng, ntot, len(cells)), T.Plan(eng, ntot, len(cells))
    ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, chunk, hs)
    ref = ms.finish(burst_events=False, nthreads=3)
    ra = torch.zeros(max(ms.ngrid, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    pa.execute(d.data_ptr(), ra.data_ptr(), hs)
    torch.cuda.synchronize()
    rec_a = ra.cpu().numpy().reshape(-1, T.REC_BYTES)
    assert all(x["noffgrid"] == 0 for x in ref) and sum(len(x["events"]) for x in ref) > 300
    for what, forced in (("device", False), ("device, second batch of the plan", False), ("forced", True)):
        if forced:
            topt("WALK_HOST", int("1"))
        rb = torch.zeros(max(ms.ngrid, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), chunk, hs)
        got = msd.collect()
        assert msd.fellback == forced and msd.ngrid == ms.ngrid, (what, msd.why)
        _same_batch_outcome(T, ref, got, rec_a, rb.cpu().numpy().reshape(-1, T.REC_BYTES), what)
        assert pb.final_codes().tolist() == pa.final_codes().tolist()
    pa.close()
    pb.close()


def test_fuzz_device_walk_batches_with_feeds_of_32_to_256_bytes(T, eng):
    """random batches of randomly damaged channels (bit flips in training sequences, spurious sequences, zeroed stretches,
    inserted / deleted bytes, payload noise), replayed with feeds of 32 / 64 / 128 / 256 bytes: tgpu_sync_multi_launch /
    _collect == tgpu_sync_multi_begin / _finish per channel (events, counts, final state, bitmap, every delivered record),
    whether the device walk settles the batch or hands it over; the reasons it gives are the documented ones, and most
    batches of the on-grid kind stay on the device"""
    import torch
    from test_stream_sync_cpu import SEQ_Y, SEQ_N
    rng = np.random.default_rng(90210)
    hs = torch.cuda.current_stream().cuda_stream
    stayed = handed = 0
    for trial in range(12):
        chunk = int(rng.choice([32, 64, 128, 256]))
        gentle = trial % 3 != 2          # two of three batches: damage that leaves the slot grid where it is
        streams = []
        for c in range(int(rng.integers(2, 6))):
            st, _ = synth.frame_stream(seed=int(rng.integers(1, 1 << 30)), nframes=int(rng.integers(3, 40)),
                                       lead_in=int(rng.integers(0, 600)), pad=int(rng.integers(700, 900)),
                                       ber=float(rng.choice([0.0, 0.02])))
            s = st.copy()
            tr = [i for i in range(0, len(s) - 60) if (s[i:i + 22] == SEQ_N).all() or (s[i:i + 38] == SEQ_Y).all()]
            for i in tr:
                if rng.random() < 0.1:
                    s[i + int(rng.integers(0, 22))] ^= 1
            for _ in range(int(rng.integers(0, 4))):
                kind = int(rng.integers(0, 3 if gentle else 5))
                p = int(rng.integers(600, len(s) - 800))
                if kind == 0:
                    s[p:p + 22] = SEQ_N
                elif kind == 1:
                    s[p:p + int(rng.integers(1, 1500))] = 0
                elif kind == 2:
                    s[p] ^= 1
                elif kind == 3:
                    s = np.concatenate([s[:p], rng.integers(0, 2, int(rng.integers(1, 40))).astype(np.uint8), s[p:]])
                else:
                    s[p:p + 38] = SEQ_Y
            streams.append(np.ascontiguousarray(s))
        d, offs, ntot = _multi_batch(T, streams)
        pa, pb = T.Plan(eng, ntot, len(streams)), T.Plan(eng, ntot, len(streams))
        ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, chunk, hs)
        ref = ms.finish(burst_events=False, nthreads=2)
        ra = torch.zeros(max(ms.ngrid, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        pa.execute(d.data_ptr(), ra.data_ptr(), hs)
        rb = torch.zeros(max(ms.ngrid, 1) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), chunk, hs)
        got = msd.collect()
        torch.cuda.synchronize()
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES),
                            "fuzz batch %d, feeds of %d" % (trial, chunk))
        assert all(w in (0, 1, 2, 3, 4, 5, 6, 7, 8) for w in msd.why), msd.why
        if gentle:
            stayed += not msd.fellback
            handed += bool(msd.fellback)
        pa.close()
        pb.close()
    assert stayed >= 4, (stayed, handed)


def test_device_walk_node_arrays_follow_the_channels(T, eng):
    """the LDS form's node arrays are laid out per launch for twice the nodes the plan's batches have shown: a quiet batch
    first (a few dozen nodes: the smallest cap), then one whose channel has over a thousand damaged slots -- that batch comes
    back through the host walks (.fellback, same outcome), the cap follows, and the same batch once more stays on the device"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cell = (262, 42, 1)
    quiet, _ = _mix_stream(T, 2000, 61, cell, ber=0.0)
    noisy, _ = _mix_stream(T, 5000, 62, cell, ber=0.0)
    noisy = noisy.copy()
    rng = np.random.default_rng(99)
    lead = 100 + 510
    nbad = 0
    for i in range(5000):
        if i % 8 and rng.random() < 0.3:
            noisy[lead + 510 * i + 244 + 4] ^= 1
            nbad += 1
    assert nbad > 1100
    d1, offs1, n1 = _multi_batch(T, [quiet, quiet])
    d2, offs2, n2 = _multi_batch(T, [quiet, noisy])
    pa, pb = T.Plan(eng, max(n1, n2), 2), T.Plan(eng, max(n1, n2), 2)
    for streams, d, offs, fb in (([quiet, quiet], d1, offs1, False), ([quiet, noisy], d2, offs2, True), ([quiet, noisy], d2, offs2, False)):
        ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, 64, hs)
        ref = ms.finish(burst_events=False, nthreads=2)
        ra = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        pa.execute(d.data_ptr(), ra.data_ptr(), hs)
        rb = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), 64, hs)
        got = msd.collect()
        torch.cuda.synchronize()
        assert msd.fellback == fb, (fb, msd.fellback)
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES), "cap follows")
    pa.close()
    pb.close()


def test_device_walk_hands_over_what_only_the_bytes_settle(T, eng):
    """streams whose walk the device form cannot settle -- a byte other than 0 / 1 next to a damaged slot, an inserted
    byte (the synchroniser re-locks beside the grid), a SYNC sequence in the first 21 bytes of a search buffer -- come
    back through the host walks: .fellback is set, the outcome is the host path's (and, off the grid, noffgrid says
    so as before)"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cell = (262, 42, 1)
    base, _ = _mix_stream(T, 600, 77, cell, ber=0.0)
    lead = 100 + 510
    cases = []
    a = base.copy()
    a[lead + 510 * 20 + 244 + 3] ^= 1
    a[lead + 510 * 20 + 100] = 7
    cases.append(a)
    b = np.concatenate([base[:lead + 510 * 33 + 17], [1], base[lead + 510 * 33 + 17:]]).astype(np.uint8)
    cases.append(b)
    from test_stream_sync_cpu import SEQ_Y
    c3 = base.copy()              # a spurious SYNC sequence behind a lost lock: a lock beside the grid, nothing delivered there
    c3[lead + 510 * 72 + 214 + 2] ^= 1
    c3[lead + 510 * 77 + 6:][:38] = SEQ_Y
    cases.append(c3)
    for st in cases:
        d, offs, ntot = _multi_batch(T, [st, base])
        pa, pb = T.Plan(eng, ntot, 2), T.Plan(eng, ntot, 2)
        ms = T.MultiSync(eng, pa, [st, base], d.data_ptr(), offs, 64, hs)
        ref = ms.finish(burst_events=False, nthreads=2)
        ra = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        pa.execute(d.data_ptr(), ra.data_ptr(), hs)
        rb = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, [st, base], d.data_ptr(), offs, rb.data_ptr(), 64, hs)
        got = msd.collect()
        torch.cuda.synchronize()
        assert msd.fellback
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES), "handed over")
        pa.close()
        pb.close()


def test_burst_kernel_long_runs_and_many_channels(T, eng, topt):
    """k_burst's look-back for the scrambling code beyond its 256-slot LDS window (one SYNC slot, then 899 NORM slots
    of the same channel: the code must still come from slot 0), a failed SB1 far back that must be skipped, and 70
    channels with carry-in codes (channels >= 64 take the code from memory): records and final codes == the batch
    kernels'"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cell = (901, 77, 9)
    code = O.scramb_get_init(*cell)
    cases = []
    # (a) one good SYNC slot, a long run behind it; carry-in code is something else
    n = 900
    ty = np.where(np.arange(n) % 2 == 0, O.TRAIN_NORM_1, O.TRAIN_NORM_2).astype(np.uint8)
    ty[0] = O.TRAIN_SYNC
    sl = T.synth_slots(ty, seed=77, scramb_init=code, mcc=cell[0], mnc=cell[1], cc=cell[2], ber=0.01)
    cases.append((ty, sl, np.zeros(n, np.uint32), np.array([0x12345], np.uint32)))
    # (b) the same with a second SYNC slot at 300 whose SB1 is destroyed: slots behind it keep slot 0's code
    sl2 = sl.copy()
    ty2 = ty.copy()
    ty2[300] = O.TRAIN_SYNC
    sl2[300] = T.synth_slots(np.array([O.TRAIN_SYNC], np.uint8), seed=5, scramb_init=code, mcc=1, mnc=2, cc=3)[0]
    sl2[300, 94:214] ^= (np.random.default_rng(3).random(120) < 0.5).astype(np.uint8)
    cases.append((ty2, sl2, np.zeros(n, np.uint32), np.array([0x12345], np.uint32)))
    # (c) 70 channels, two slots each, no SYNC slot: every channel decodes with its own carry-in code
    nch = 70
    carry = (np.arange(nch, dtype=np.uint32) * 2654435761 + 3).astype(np.uint32)
    ty3 = np.tile(np.array([O.TRAIN_NORM_1, O.TRAIN_NORM_2], np.uint8), nch)
    ch3 = np.repeat(np.arange(nch, dtype=np.uint32), 2)
    sl3 = np.concatenate([T.synth_slots(ty3[2 * c:2 * c + 2], seed=900 + c, scramb_init=int(carry[c])) for c in range(nch)])
    cases.append((ty3, sl3, ch3, carry))
    for ty_, sl_, ch_, carry_ in cases:
        n_ = len(ty_)
        d = torch.from_numpy(sl_.reshape(-1)).cuda()
        out = {}
        for mode, mx in (("burst", "100000"), ("batch", "0")):
            topt("BURST_MAX", int(mx))
            d_rec = torch.zeros(n_ * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            plan = T.Plan(eng, n_, len(carry_))
            plan.load(np.arange(n_, dtype=np.uint64) * 510, ty_, ch_, carry_)
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
            torch.cuda.synchronize()
            out[mode] = (d_rec.cpu().numpy().reshape(n_, T.REC_BYTES), plan.final_codes().tolist())
            plan.close()
        assert (out["burst"][0] == out["batch"][0]).all() and out["burst"][1] == out["batch"][1]
        p = T.parse_records(out["burst"][0])
        if len(carry_) == 1:
            assert (p["code"][1:] == code).all() and p["crc_ok"][1:, 0].mean() > 0.5
        else:
            assert (p["code"] == carry_[ch_]).all() and p["crc_ok"][:, 0].all()


def test_wire_only_mode(T, eng, topt):
    """tgpu_plan_set_wire_only: the wire records are the ones of a normal run byte for byte (all burst types, noise, code
    learnt from SB1 mid-batch, the RM option), the 320-byte records are left alone (but for the type byte of ignored
    slots), the scrambling codes in force afterwards are the same"""
    import torch
    topt("BURST_MAX", int("0"))
    hs = torch.cuda.current_stream().cuda_stream
    n = 5000
    rng = np.random.default_rng(12)
    types = np.tile(np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8), n // 8 + 1)[:n]
    types[rng.integers(0, n, 20)] = 2                      # ignored type
    cell = (262, 42, 1)
    code = O.scramb_get_init(*cell)
    slots = T.synth_slots(np.where(types == 2, 0, types).astype(np.uint8), seed=9, scramb_init=code, mcc=cell[0], mnc=cell[1], cc=cell[2], ber=0.03)
    d = torch.from_numpy(slots.reshape(-1)).cuda()
    for rm in (False, True):
        res = {}
        for mode in ("full", "wire_only"):
            plan = T.Plan(eng, n, 1)
            plan.set_rm_decode(rm)
            plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([3], np.uint32))
            d_rec = torch.full((n * T.REC_BYTES,), 0xAB, dtype=torch.uint8, device="cuda")
            d_wire = torch.full((n * T.WIRE_BYTES,), 0xFF, dtype=torch.uint8, device="cuda")
            plan.set_wire(d_wire.data_ptr())
            plan.set_wire_only(mode == "wire_only")
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
            torch.cuda.synchronize()
            res[mode] = (d_rec.cpu().numpy().reshape(n, T.REC_BYTES), d_wire.cpu().numpy().reshape(n, T.WIRE_BYTES),
                         plan.final_codes().tolist())
            plan.close()
        assert (res["full"][1] == res["wire_only"][1]).all() and res["full"][2] == res["wire_only"][2] == [code]
        untouched = res["wire_only"][0].copy()
        assert (untouched[types != 2] == 0xAB).all()
        assert (untouched[types == 2][:, 1:] == 0xAB).all() and (untouched[types == 2][:, 0] == 0xFF).all()
        # and the wire records unpack to the full run's records
        for i in (0, 1, 2, 3, 9, n - 1):
            back = T.wire_unpack(res["wire_only"][1][[i]], [i], [code])[0]
            full = res["full"][0][i]
            pa, pb = T.parse_records(back[None]), T.parse_records(full[None])
            n1, n2 = {0: (268, 0), 1: (124, 124), 3: (60, 124)}[int(types[i])]      # (the full run's buffer was prefilled)
            assert (pa["bits1"][0][:n1] == pb["bits1"][0][:n1]).all() and (pa["bits2"][0][:n2] == pb["bits2"][0][:n2]).all()
            for k in ("type", "bbk"):
                assert (np.asarray(pa[k]) == np.asarray(pb[k])).all(), (i, k)
            nb = 2 if n2 else 1
            assert (pa["crc"][0][:nb] == pb["crc"][0][:nb]).all() and (pa["crc_ok"][0][:nb] == pb["crc_ok"][0][:nb]).all()


def test_comm_gather_single_rank(T, eng):
    """tgpu_comm_*: the C-ABI gather over RCCL with the one rank a 1-GPU box has -- id, communicator, the grouped
    send / receive to the root (here: to itself) on a side stream, twice with different sizes, wire records of a
    decoded batch arriving byte for byte; bad arguments are refused"""
    import torch
    uid = T.comm_unique_id()
    assert uid.shape == (T.COMM_ID_BYTES,) and uid.any()
    comm = T.Comm(eng, uid, 0, 1)
    n = 3000
    types = np.tile(np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8), n // 8 + 1)[:n]
    slots = T.synth_slots(types, seed=5, scramb_init=O.scramb_get_init(262, 42, 1))
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_wire = torch.full((n * T.WIRE_BYTES,), 0xff, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([3], np.uint32))
    plan.set_wire(d_wire.data_ptr())
    side = torch.cuda.Stream()
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    done = torch.cuda.Event()
    done.record()
    sink = torch.zeros(n * T.WIRE_BYTES, dtype=torch.uint8, device="cuda")
    side.wait_event(done)
    comm.gather(d_wire.data_ptr(), n * T.WIRE_BYTES, sink.data_ptr(), 0, side.cuda_stream)
    side.synchronize()
    assert torch.equal(sink, d_wire)
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    w = sink.cpu().numpy().reshape(n, T.WIRE_BYTES)
    for i in (0, 1, 2, 7, n - 1):
        assert (w[i] == T.wire_pack(rec[i])).all()
    small = torch.arange(100, dtype=torch.uint8, device="cuda")
    out = torch.zeros(100, dtype=torch.uint8, device="cuda")
    comm.gather(small.data_ptr(), 100, out.data_ptr(), 0, side.cuda_stream)
    side.synchronize()
    assert torch.equal(small, out)
    # a size per rank (the compact transport form's exchange): exact bytes at the root's offset, nothing beyond them
    sink2 = torch.full((4096,), 0x5A, dtype=torch.uint8, device="cuda")
    comm.gatherv(small.data_ptr(), [77], sink2.data_ptr(), [256], 0, side.cuda_stream)
    side.synchronize()
    got = sink2.cpu().numpy()
    assert (got[256:256 + 77] == np.arange(77)).all() and (got[:256] == 0x5A).all() and (got[256 + 77:] == 0x5A).all()
    comm.gatherv(0, [0], sink2.data_ptr(), [0], 0, side.cuda_stream)             # a rank with nothing to send
    side.synchronize()
    with pytest.raises(T.TgpuError):
        comm.gatherv(small.data_ptr(), [77], 0, [0], 0, side.cuda_stream)         # the root needs a sink
    with pytest.raises(T.TgpuError):
        comm.gather(small.data_ptr(), 100, out.data_ptr(), 1, side.cuda_stream)      # no such root
    with pytest.raises(T.TgpuError):
        comm.gather(small.data_ptr(), 100, 0, 0, side.cuda_stream)                   # the root needs a sink
    with pytest.raises(T.TgpuError):
        T.Comm(eng, uid, 1, 1)
    comm.close()
    plan.close()


def test_metric_workload_full_size_against_the_oracle(T, eng):
    """The default bench line's step at its own size: 8 recorded channels x 125 000 slots (own cell each, 1 % damaged
    training sequences, 1 % payload bit errors) as ONE batch through the multi-channel synchroniser (64-byte feeds) and
    plan.  Per channel: the synchroniser's events and the number of delivered bursts == the oracle receiver's on that
    channel's bytes; EVERY delivered burst's type, type-1 bits, BBK, CRC words and flags == the oracle's decode of the
    same 510 bytes; the channel's final scrambling code"""
    import ctypes as C
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (262, 42, 2), (505, 1, 60), (208, 10, 5), (222, 99, 7)]
    per = 125_000
    streams, codes, offs, o = [], [], [], 0
    for c, cell in enumerate(cells):
        st, code = _mix_stream(T, per, 3100 + c, cell, ber=0.01)
        streams.append(st)
        codes.append(code)
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    d = torch.from_numpy(buf).cuda()
    ntot = sum((len(st) // 510 + 32) for st in streams)
    plan = T.Plan(eng, ntot, len(cells))
    ms = T.MultiSync(eng, plan, streams, d.data_ptr(), offs, 64, hs)
    outs = ms.finish(burst_events=False, nthreads=8)
    d_rec = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    rec = d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)
    fin = plan.final_codes()
    total = 0
    for c, (st, out) in enumerate(zip(streams, outs)):
        want_ev = []
        rx = O.Rx()
        ecb = O.EVENT_CB(lambda ev, bitnum, arg, priv: want_ev.append((ev, bitnum, arg)) if ev != 2 else None)
        O.lib().orc_rx_init(C.byref(rx), O.UPPER_CB(), ecb, None)
        rx.use_acc = 1
        O.lib().orc_rx_feed(C.byref(rx), O._p(st), len(st), 64)
        assert out["events"] == want_ev and len(want_ev) > 1500, c
        # every slot the locked receiver looked at is either delivered or reported (events 3, 4, 5)
        dropped = sum(1 for e in want_ev if e[0] in (3, 4, 5))
        assert out["noffgrid"] == 0 and out["nslots"] == rx.burst_seq - dropped, c
        idx = T.grid_indices(out)
        assert len(idx) == out["nslots"] and 0.93 * per < len(idx) < per
        r = rec[out["grid_base"] + idx]
        slots = st[out["anchor"]:out["anchor"] + 510 * (int(idx[-1]) + 1)].reshape(-1, 510)[idx]
        ty = T.parse_records(r)["type"].astype(np.uint8)
        ok, p = check_against_oracle(T, r, ty, slots, codes[c], use_acc=1)
        nblk = int(2 * (ty != 0).sum() + (ty == 0).sum())
        assert 0.2 * nblk < ok <= nblk
        assert (p["code"][ty != 3] == codes[c]).all() and int(fin[c]) == codes[c]
        total += len(idx)
    assert total > 930_000
    # the same batch with the walks on the device (k_walk; what bench.py times): no fallback, every channel's events,
    # counts and delivered bitmap equal the host walks' (just checked against the oracle), every record byte equal
    plan2 = T.Plan(eng, ntot, len(cells))
    d_rec2 = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    msd = T.MultiSyncDev(eng, plan2, streams, d.data_ptr(), offs, d_rec2.data_ptr(), 64, hs)
    outs2 = msd.collect()
    assert not msd.fellback and msd.ngrid == ms.ngrid
    rec2 = d_rec2.cpu().numpy().reshape(-1, T.REC_BYTES)
    for c, (a, b) in enumerate(zip(outs, outs2)):
        assert a["events"] == b["events"], c
        for k in ("nslots", "ngrid", "noffgrid", "anchor", "final_state", "burst_seq", "tail_tn_adds", "grid_base"):
            assert a[k] == b[k], (c, k)
        assert (np.asarray(a["grid_bits"]) == np.asarray(b["grid_bits"])).all()
        idx = a["grid_base"] + T.grid_indices(a)
        assert (rec[idx] == rec2[idx]).all(), c
    assert plan2.final_codes().tolist() == fin.tolist()
    plan2.close()
    plan.close()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 16, 33, 200, 1500])
def test_burst_kernel_equals_batch_kernels(T, eng, n, topt):
    """k_burst (one workgroup per burst, trellis states across lanes; the path of small batches) writes the very
    records the lane-per-trellis batch kernels write -- every byte -- and both are the oracle's decode: mixed burst
    types with SYNC slots that switch the code mid-batch (a valid cell, a failed SB1, a second cell), two channels
    with carry-in codes, payload noise up to garbage, an ignored burst type, misaligned slot offsets"""
    import torch
    rng = np.random.default_rng(100 + n)
    cells = [(262, 42, 1), (901, 77, 9)]
    codes = [O.scramb_get_init(*c) for c in cells]
    for trial in range(6 if n < 100 else 2):
        ty = rng.choice([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2], n, p=[0.3, 0.35, 0.35]).astype(np.uint8)
        cut = int(rng.integers(0, n + 1))                      # channel 0: slots < cut, channel 1: the rest
        chan = (np.arange(n) >= cut).astype(np.uint32)
        slots = np.zeros((n, 510), np.uint8)
        cur = [codes[0], 0]                                    # carry-in codes: channel 0 knows its cell, channel 1 does not
        carry = np.array(cur, np.uint32)
        for i in range(n):
            c = int(chan[i])
            cell = cells[int(rng.integers(0, 2))]
            ber = float(rng.choice([0.0, 0.02, 0.08, 0.5]))
            if ty[i] == O.TRAIN_SYNC:
                code_i = O.scramb_get_init(*cell)
                sl = T.synth_slots(ty[i:i + 1], seed=int(rng.integers(1, 1 << 30)), scramb_init=code_i, mcc=cell[0], mnc=cell[1], cc=cell[2], ber=ber)
            else:
                sl = T.synth_slots(ty[i:i + 1], seed=int(rng.integers(1, 1 << 30)), scramb_init=int(cur[c]), ber=ber)
            slots[i] = sl[0]
            if ty[i] == O.TRAIN_SYNC and O.decode_block(O.T_SB1, sl[0][94:214], 3)[2]:
                cur[c] = O.scramb_get_init(*cell)
        if n >= 3 and trial == 0:
            ty[n // 2] = 2                                      # TETRA_TRAIN_NORM_3: ignored
        stride = 510 + int(rng.integers(0, 7))
        off0 = int(rng.integers(0, 5))
        buf = np.zeros(off0 + n * stride + 8, np.uint8)
        offs = off0 + np.arange(n, dtype=np.uint64) * stride
        for i in range(n):
            buf[int(offs[i]):int(offs[i]) + 510] = slots[i]
        if trial == 1:
            buf[int(offs[0]) + 300] = 7                         # a non-binary byte: flag + "anything else is a 1"
        d = torch.from_numpy(buf).cuda()
        out = {}
        for mode, mx in (("burst", "100000"), ("batch", "0")):
            topt("BURST_MAX", int(mx))
            d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
            plan = T.Plan(eng, n, 2)
            plan.load(offs, ty, chan, carry)
            plan.execute(d.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            out[mode] = (d_rec.cpu().numpy().reshape(n, T.REC_BYTES), plan.final_codes().tolist())
            plan.close()
        a, b = out["burst"], out["batch"]
        assert (a[0] == b[0]).all(), (n, trial, np.argwhere(a[0] != b[0])[:8].tolist())
        assert a[1] == b[1]
        # ... and the oracle's, slot by slot with the code the oracle has in force
        keep = ty != 2
        exp = np.zeros(n, np.uint32)
        cur = [int(carry[0]), int(carry[1])]
        sl_in = np.stack([buf[int(offs[i]):int(offs[i]) + 510] for i in range(n)])
        for i in range(n):
            c = int(chan[i])
            if ty[i] == O.TRAIN_SYNC:
                t1, _, ok, _ = O.decode_block(O.T_SB1, (sl_in[i][94:214] != 0).astype(np.uint8), 3)
                if ok:
                    f = lambda x, k: int("".join(str(int(v)) for v in t1[x:x + k]), 2)
                    cur[c] = O.scramb_get_init(f(31, 10), f(41, 14), f(4, 6))
            exp[i] = cur[c]
        p = T.parse_records(a[0])
        assert (p["code"][keep] == exp[keep]).all()
        for cval in sorted(set(exp[keep].tolist())):
            m = keep & (exp == cval)
            check_against_oracle(T, a[0][m], ty[m], (sl_in[m] != 0).astype(np.uint8), int(cval))


def test_burst_path_then_batch_path_on_one_load(T, eng, topt):
    """one load, several executes on different paths: a static batch (no SYNC slot) goes through k_burst,
    which uses the mask index / mask table as its own scratch, and is then executed again on the lane-per-trellis
    kernels (per-stage profiling, a wire buffer, the RM option, TGPU_BURST_MAX lowered): the static mask table must be
    rebuilt -- every execute gives the first one's records.  And a channel whose LAST slot is of an ignored burst type
    still reports its code after a k_burst execute (the batch kernels' forward fill gives every slot one)"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    n = 300
    rng = np.random.default_rng(31)
    ty = rng.choice([O.TRAIN_NORM_1, O.TRAIN_NORM_2], n).astype(np.uint8)
    chan = (np.arange(n) >= 120).astype(np.uint32)
    carry = np.array([O.scramb_get_init(262, 42, 1), O.scramb_get_init(901, 77, 9)], np.uint32)
    sl = np.concatenate([T.synth_slots(ty[:120], seed=1, scramb_init=int(carry[0]), ber=0.02),
                         T.synth_slots(ty[120:], seed=2, scramb_init=int(carry[1]), ber=0.02)])
    d = torch.from_numpy(sl.reshape(-1)).cuda()
    plan = T.Plan(eng, n, 2)
    plan.load(np.arange(n, dtype=np.uint64) * 510, ty, chan, carry)
    recs = []
    prof = T.Prof(1)
    d_wire = torch.full((n * T.WIRE_BYTES,), 0xFF, dtype=torch.uint8, device="cuda")
    for step in range(5):
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        if step == 1:
            plan.execute_prof(d.data_ptr(), d_rec.data_ptr(), hs, prof, 0)       # never the burst path
        elif step == 3:
            plan.set_wire(d_wire.data_ptr())                                        # neither
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
            plan.set_wire(0)
        else:
            plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)                        # k_burst (n <= 1024)
        torch.cuda.synchronize()
        recs.append(d_rec.cpu().numpy())
        assert plan.final_codes().tolist() == carry.tolist(), step
    for step in range(1, 5):
        assert (recs[step] == recs[0]).all(), step
    p = T.parse_records(recs[0].reshape(n, T.REC_BYTES))
    assert (p["code"] == carry[chan]).all() and p["crc_ok"][:, 0].mean() > 0.5
    plan.close()
    # an ignored burst type in a channel's last slot, SYNC slots in the batch: the code in force comes from them
    ty2 = np.array([3, 0, 1, 0, 2, 3, 1, 2], np.uint8)
    chan2 = np.array([0, 0, 0, 0, 0, 1, 1, 1], np.uint32)
    cells = [(262, 42, 1), (901, 77, 9)]
    sl2 = np.concatenate([T.synth_slots(np.where(ty2[:5] == 2, 0, ty2[:5]).astype(np.uint8), seed=3, scramb_init=O.scramb_get_init(*cells[0]),
                                        mcc=262, mnc=42, cc=1),
                          T.synth_slots(np.where(ty2[5:] == 2, 0, ty2[5:]).astype(np.uint8), seed=4, scramb_init=O.scramb_get_init(*cells[1]),
                                        mcc=901, mnc=77, cc=9)])
    d2 = torch.from_numpy(sl2.reshape(-1)).cuda()
    res = {}
    for mode, mx in (("burst", "100000"), ("batch", "0")):
        topt("BURST_MAX", int(mx))
        pl = T.Plan(eng, 8, 2)
        pl.load(np.arange(8, dtype=np.uint64) * 510, ty2, chan2, np.array([5, 9], np.uint32))
        r = torch.zeros(8 * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        pl.execute(d2.data_ptr(), r.data_ptr(), hs)
        torch.cuda.synchronize()
        res[mode] = (r.cpu().numpy(), pl.final_codes().tolist())
        pl.close()
    assert res["burst"][1] == res["batch"][1] == [O.scramb_get_init(*cells[0]), O.scramb_get_init(*cells[1])]
    assert (res["burst"][0] == res["batch"][0]).all()


def test_gsmtap_messages_of_a_batch_on_device(T, eng):
    """tgpu_gsmtap_batch (k_gsmtap): the GSMTAP message of every CRC-OK block of a decoded batch -- all three burst types,
    payload noise (failed CRCs give no message), SYNC bursts that move the clock, the BNCH frame, traffic bursts with and
    without a stolen second block -- against the oracle's restatement of tetra_gsmtap_makemsg() (tetra_gsmtap.c:31-63)
    driven by the reference's call (tetra_upper_mac.c:483-486: ts = tn - 1, ss = signal = snr = 0, the block's type-1
    bits) with the clock and logical channel of tetra_lower_mac.c:167-173, 291-319; and against the product's own host
    function"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    n = 900
    rng = np.random.default_rng(77)
    cell = (262, 42, 1)
    code = O.scramb_get_init(*cell)
    types = np.tile(np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8), n // 8 + 1)[:n]
    types[(np.arange(n) // 4) % 18 == 17] = 3       # (the generator's SYNC PDU of slot i carries fn = (i / 4) % 18 + 1: frame 18 -> BNCH)
    slots = T.synth_slots(types, seed=5000, scramb_init=code, mcc=cell[0], mnc=cell[1], cc=cell[2], ber=0.04)
    # the PHY clock burst by burst: one time step per burst (tetra_tdma.c:27-58), a good SB1 sets tn / fn / mn
    tm = [0, 0, 1, 1, 1]              # hn, sn, tn, fn, mn
    times = np.zeros(n, T.TDMA_TIME_DTYPE)
    clock = []
    for i in range(n):
        tm[2] += 1
        if tm[2] > 4:
            tm[3] += tm[2] // 4
            tm[2] %= 4
        if tm[3] > 18:
            tm[4] += tm[3] // 18
            tm[3] %= 18
        if tm[4] > 60:
            tm[4] %= 60
        times[i] = (tm[0], 0, tm[1], tm[2], tm[3], tm[4])
        clock.append(tuple(tm))
        if types[i] == 3 and O.decode_block(O.T_SB1, slots[i][94:214], 3)[2]:
            tm[2], tm[3], tm[4] = i % 4 + 1, (i // 4) % 18 + 1, (i // 72) % 60 + 1
    traffic = np.zeros(n, np.uint8)
    traffic[rng.integers(0, n, 60)] = 1
    traffic[rng.integers(0, n, 30)] = 3
    d = torch.from_numpy(slots.reshape(-1)).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([code], np.uint32))
    plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    d_times = torch.from_numpy(times.view(np.uint8)).cuda()
    d_traffic = torch.from_numpy(traffic).cuda()
    d_msgs = torch.full((3 * n * T.GSMTAP_STRIDE,), 0xEE, dtype=torch.uint8, device="cuda")
    d_lens = torch.full((3 * n,), 0xEE, dtype=torch.uint8, device="cuda")
    T.gsmtap_batch(eng, d_rec.data_ptr(), d_times.data_ptr(), n, d_msgs.data_ptr(), d_lens.data_ptr(), d_traffic.data_ptr(), hs)
    torch.cuda.synchronize()
    msgs = d_msgs.cpu().numpy().reshape(n, 3, T.GSMTAP_STRIDE)
    lens = d_lens.cpu().numpy().reshape(n, 3)
    rec = d_rec.cpu().numpy().reshape(n, T.REC_BYTES)
    LC = {"SCH_F": 1, "AACH": 8, "BSCH": 10, "BNCH": 11}
    nmsg = nbnch = 0
    for i in range(n):
        blocks = T.record_blocks(rec[i])
        t = list(clock[i])
        sb1_ok = types[i] == 3 and blocks[0]["crc_ok"]
        if sb1_ok:
            f0 = int(T.parse_records(rec[i][None])["sbf0"][0])
            t[2], t[3], t[4] = (f0 >> 8) & 0xFF, (f0 >> 16) & 0xFF, f0 >> 24
            assert (t[2], t[3], t[4]) == (i % 4 + 1, (i // 4) % 18 + 1, (i // 72) % 60 + 1)
        for k, b in enumerate(blocks):
            is_bnch = t[3] == 18 and t[2] == 4 - ((t[4] + 3) % 4)
            lchan = {T.T_SB1: LC["BSCH"], T.T_BBK: LC["AACH"], T.T_SCH_F: LC["SCH_F"]}.get(b["type"], LC["BNCH"] if (b["type"] == T.T_SB2 and is_bnch) else 0)
            dumped = traffic[i] & 1 and (b["type"] == T.T_SCH_F or (b["blk_num"] == 2 and not (traffic[i] & 2)))
            if not b["crc_ok"] or dumped:
                assert lens[i, k] == 0, (i, k)
                continue
            t1 = np.frombuffer(b["type1"], np.uint8)
            want = np.frombuffer(bytes(O.gsmtap_makemsg((t[0], t[1], t[2], t[3], t[4]), lchan, t[2] - 1, t1)), np.uint8)
            assert lens[i, k] == len(want) and (msgs[i, k, :len(want)] == want).all(), (i, k)
            own = np.frombuffer(bytes(T.gsmtap_makemsg((t[0], t[1], t[2], t[3], t[4]), lchan, t[2] - 1, t1)), np.uint8)
            assert (own == want).all()
            nmsg += 1
            nbnch += lchan == LC["BNCH"]
        for k in range(len(blocks), 3):
            assert lens[i, k] == 0
    assert nmsg > 1200 and nbnch >= 3, (nmsg, nbnch)
    plan.close()


@pytest.mark.gpu
def test_device_host_locality(T):
    """tgpu_device_host_locality(): the GPU's PCI address as sysfs spells it, a NUMA node (or -1) and that node's CPUs"""
    import os
    import re
    bdf, node, cpus = T.device_host_locality(0)
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", bdf), bdf
    assert node >= -1 and all(c >= 0 for c in cpus) and cpus == sorted(set(cpus))
    if os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bdf):
        assert node == int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())


@pytest.mark.gpu
@pytest.mark.parametrize("ber", [0.0, 0.05])
def test_stages_step_by_step_and_against_the_fused_path(T, eng, ber):
    """tgpu_stages_*: the lower MAC one step at a time on the device (what the reference prints with DEBUGP,
    tetra_lower_mac.c:175-254) == the oracle's step functions (scrambler, de-interleaver, de-puncturer, Viterbi, CRC), for
    every block type; and the fused kernels' type-1 bits / CRC words of the same blocks == the staged chain's"""
    import torch
    rng = np.random.default_rng(int(ber * 100) + 77)
    hs = torch.cuda.current_stream().cuda_stream
    codes_pool = np.array([0, 3, 0x41802A07, 0x12345677, 0xFFFFFFFF], np.uint32)
    for t in (O.T_SB1, O.T_SB2, O.T_NDB, O.T_SCH_HU, O.T_SCH_F, O.T_BBK):
        K, n2, n1, a = O.BLK[t]
        n = 300
        codes = codes_pool[rng.integers(0, len(codes_pool), n)]
        t5 = np.zeros((n, K), np.uint8)
        for i in range(n):
            enc = 3 if t == O.T_SB1 else int(codes[i])
            t5[i] = O.encode_bbk(rng.integers(0, 2, 14).astype(np.uint8), enc) if t == O.T_BBK else \
                O.encode_block(t, rng.integers(0, 2, n1).astype(np.uint8), enc)
        t5 ^= (rng.random(t5.shape) < ber).astype(np.uint8)
        st = T.Stages(eng, t)
        assert (st.K, st.type2_len, st.type1_len) == (K, n2 if t != O.T_BBK else 14, n1)
        d5 = torch.from_numpy(t5 * 255).cuda()        # (a received byte other than 0 is a 1)
        dc = torch.from_numpy(codes.view(np.int32)).cuda()
        d4 = torch.full((n, K), 7, dtype=torch.uint8, device="cuda")
        d3 = torch.full((n, K), 7, dtype=torch.uint8, device="cuda")
        ddp = torch.full((n, max(st.mother_len, 1)), 7, dtype=torch.uint8, device="cuda")
        d2 = torch.full((n, st.type2_len), 7, dtype=torch.uint8, device="cuda")
        dcrc = torch.zeros(n, dtype=torch.int16, device="cuda")
        st.execute(d5.data_ptr(), dc.data_ptr(), n, d4.data_ptr(), d3.data_ptr(), ddp.data_ptr(), d2.data_ptr(), dcrc.data_ptr(), hs)
        torch.cuda.synchronize()
        g4, g3, gdp, g2 = d4.cpu().numpy(), d3.cpu().numpy(), ddp.cpu().numpy(), d2.cpu().numpy()
        gcrc = dcrc.cpu().numpy().view(np.uint16)
        for i in range(n):
            code = 3 if t == O.T_SB1 else int(codes[i])
            w4 = O.scramb(code, t5[i])
            assert (g4[i] == w4).all(), (t, i, "type4")
            if t == O.T_BBK:
                assert (g2[i] == w4[:14]).all() and (g3[i] == 7).all()
                continue
            w3 = O.deinterleave(K, a, w4)
            assert (g3[i] == w3).all(), (t, i, "type3")
            wdp = O.depuncture(0, w3, 4 * n2)
            assert (gdp[i] == wdp).all(), (t, i, "type3dp")
            w2 = O.viterbi_hard(wdp, n2)
            assert (g2[i] == w2).all(), (t, i, "type2")
            assert gcrc[i] == O.crc16(w2[:n1 + 16]), (t, i, "crc")
        # the fused kernels on the same blocks
        plan = T.Plan(eng, n, 8)
        flat = torch.from_numpy(t5.reshape(-1)).cuda()
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        plan.load_blocks(np.arange(n, dtype=np.uint64) * K, np.full(n, t, np.uint8), codes)
        plan.execute(flat.data_ptr(), d_rec.data_ptr(), hs)
        torch.cuda.synchronize()
        p = T.parse_records(d_rec.cpu().numpy().reshape(n, T.REC_BYTES))
        if t == O.T_BBK:
            assert (p["bbk"] == g2[:, :14]).all()
        else:
            assert (p["bits1"][:, :n1] == g2[:, :n1]).all() and (p["crc"][:, 0] == gcrc).all()
            assert (p["crc_ok"][:, 0] == (gcrc == 0x1d0f)).all()
        plan.close()
        st.close()
    with pytest.raises(T.TgpuError):
        T.Stages(eng, 9)


def _oracle_check_of_a_walked_channel(T, st, out, rec, code, nrec=20000, chunk=64):
    """a channel of a device-walk batch against the ORACLE's receiver on the same bytes (not against the product's own host
    walk): every synchroniser event, the number of delivered bursts, and the records of a slice of the delivered bursts
    (nrec of them: from the start, the middle and the end) against the oracle's decode"""
    import ctypes as C
    want_ev = []
    rx = O.Rx()
    ecb = O.EVENT_CB(lambda ev, bitnum, arg, priv: want_ev.append((ev, bitnum, arg)) if ev != 2 else None)
    O.lib().orc_rx_init(C.byref(rx), O.UPPER_CB(), ecb, None)
    rx.use_acc = 1
    O.lib().orc_rx_feed(C.byref(rx), O._p(st), len(st), chunk)
    assert out["events"] == want_ev
    dropped = sum(1 for e in want_ev if e[0] in (3, 4, 5))
    assert out["noffgrid"] == 0 and out["nslots"] == rx.burst_seq - dropped
    idx = T.grid_indices(out)
    assert len(idx) == out["nslots"]
    third = nrec // 3
    pick = np.unique(np.concatenate([np.arange(min(third, len(idx))), len(idx) // 2 + np.arange(min(third, len(idx) - len(idx) // 2)),
                                     np.arange(max(0, len(idx) - third), len(idx))]))
    gi = idx[pick]
    r = rec[out["grid_base"] + gi]
    slots = np.stack([st[out["anchor"] + 510 * int(g):out["anchor"] + 510 * int(g) + 510] for g in gi])
    ty = T.parse_records(r)["type"].astype(np.uint8)
    ok, p = check_against_oracle(T, r, ty, slots, code, use_acc=1)
    assert (p["code"][ty != 3] == code).all()
    return len(want_ev), len(pick)


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [128, 256])
def test_device_walk_with_long_feeds_against_the_oracle(T, eng, chunk):
    """three bench channels of 40 000 slots (1 % damaged training sequences) replayed with feeds of 128 / 256 bytes, walks on
    the device: no hand-over, and per channel the oracle receiver's events (fed the same bytes in the same feeds), its number of
    delivered bursts, and 6 000 delivered records against the oracle's decode"""
    import torch
    import bench
    hs = torch.cuda.current_stream().cuda_stream
    made = [bench.make_mix_stream(T, 40000, 20 + c, mnc=70 + c, cc=2 + c) for c in range(3)]
    streams = [np.ascontiguousarray(m[0]) for m in made]
    d, offs, ntot = _multi_batch(T, streams)
    plan = T.Plan(eng, ntot, 3)
    rec = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    msd = T.MultiSyncDev(eng, plan, streams, d.data_ptr(), offs, rec.data_ptr(), chunk, hs)
    outs = msd.collect()
    torch.cuda.synchronize()
    assert not msd.fellback, msd.why
    r = rec.cpu().numpy().reshape(-1, T.REC_BYTES)
    for c in range(3):
        nev, npick = _oracle_check_of_a_walked_channel(T, streams[c], outs[c], r, made[c][2], nrec=6000, chunk=chunk)
        assert nev > 400 and npick >= 5000
    plan.close()


@pytest.mark.gpu
def test_device_walk_of_a_channel_beyond_one_workgroups_arrays(T, eng):
    """a recording of 300 000 slots (more than the 262 144 whose bitmap and node list k_walk keeps in LDS) next to a
    short one in the same batch: k_walk_big walks it with its arrays in the plan's scratch area -- no hand-over to the
    host walks, and events (5 600 of them: more than the block that comes down with the batch holds), counts, final state, delivered bitmap, every
    delivered record and the final codes equal the host-walk batch's"""
    import torch
    import bench
    hs = torch.cuda.current_stream().cuda_stream
    st0, _, code0 = bench.make_mix_stream(T, 300000, 2, mnc=61, cc=4)
    st1, _, code1 = bench.make_mix_stream(T, 5000, 3, mnc=62, cc=5)
    streams = [np.ascontiguousarray(st0), np.ascontiguousarray(st1)]
    d, offs, ntot = _multi_batch(T, streams)
    pa, pb = T.Plan(eng, ntot, 2), T.Plan(eng, ntot, 2)
    ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, 64, hs)
    ref = ms.finish(burst_events=False, nthreads=2)
    ra = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    pa.execute(d.data_ptr(), ra.data_ptr(), hs)
    torch.cuda.synchronize()
    assert ref[0]["ngrid"] > 262144 and len(ref[0]["events"]) > 4096     # (more events than the block that comes down with the batch)
    for rep in range(2):            # (twice: the scratch area is allocated by the first batch that needs it)
        rb = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), 64, hs)
        got = msd.collect()
        assert not msd.fellback and msd.ngrid == ms.ngrid
        rec_b = rb.cpu().numpy().reshape(-1, T.REC_BYTES)
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rec_b, "long channel")
        assert pb.final_codes().tolist() == pa.final_codes().tolist()
        if rep == 0:        # the device walk's outcome against the oracle's receiver on the same bytes, both channels
            nev, nrec = _oracle_check_of_a_walked_channel(T, streams[0], got[0], rec_b, code0)
            assert nev > 4096 and nrec >= 20000 - 2
            _oracle_check_of_a_walked_channel(T, streams[1], got[1], rec_b, code1)
    pa.close()
    pb.close()


@pytest.mark.gpu
def test_device_walk_takes_noisy_channels_into_the_long_form(T, eng):
    """a channel short enough for the LDS form of the walk but with more exceptions than that form holds (150 000 slots, 8 %
    damaged training sequences: > 8192 nodes): the first batch goes through the host walks (.fellback), the plan remembers
    the density, and the next batches walk that channel in the long form on the device -- same outcome every time"""
    import torch
    import bench
    hs = torch.cuda.current_stream().cuda_stream
    st0, _, code0 = bench.make_mix_stream(T, 150000, 6, mnc=71, cc=2, damaged=0.08)
    st1, _, code1 = bench.make_mix_stream(T, 4000, 7, mnc=72, cc=3)
    streams = [np.ascontiguousarray(st0), np.ascontiguousarray(st1)]
    d, offs, ntot = _multi_batch(T, streams)
    pa, pb = T.Plan(eng, ntot, 2), T.Plan(eng, ntot, 2)
    ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, 64, hs)
    ref = ms.finish(burst_events=False, nthreads=2)
    ra = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    pa.execute(d.data_ptr(), ra.data_ptr(), hs)
    torch.cuda.synchronize()
    fell = []
    for rep in range(3):
        rb = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), 64, hs)
        got = msd.collect()
        fell.append(bool(msd.fellback))
        rec_b = rb.cpu().numpy().reshape(-1, T.REC_BYTES)
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rec_b, "noisy channel %d" % rep)
        if rep == 1:        # the long form's outcome (k_walk_big on a channel with > 8192 exceptions) against the oracle's receiver
            nev, nrec = _oracle_check_of_a_walked_channel(T, streams[0], got[0], rec_b, code0)
            assert nev > 8192 and nrec >= 20000 - 2
            _oracle_check_of_a_walked_channel(T, streams[1], got[1], rec_b, code1)
    assert fell == [True, False, False], fell
    pa.close()
    pb.close()


@pytest.mark.gpu
def test_device_walk_batches_on_small_plans(T, eng):
    """device-walk batches on plans of a few hundred slots, several alive at once, each channel of another cell: such plans
    keep their upload arena in mapped host memory (where the walk's kernels must not run their atomics: they get a device
    arena) and their mask table must hold the batch's code hash table whatever the plan's size (entries up to
    1 + nchan + 4096).  Outcome == the host-walk batch's, for every plan, after all of them ran"""
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    sets = []
    for k in range(6):
        cells = [(262, 42 + 7 * k, 1 + k), (901, 77 + k, 9 + k)]
        streams = [_mix_stream(T, 40 + 5 * k, 8800 + 10 * k + c, cell, ber=0.0)[0] for c, cell in enumerate(cells)]
        d, offs, ntot = _multi_batch(T, streams)
        assert ntot <= 256                # (the mapped-arena size class)
        pa, pb = T.Plan(eng, ntot, 2), T.Plan(eng, ntot, 2)
        ms = T.MultiSync(eng, pa, streams, d.data_ptr(), offs, 64, hs)
        ref = ms.finish(burst_events=False, nthreads=2)
        ra = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        pa.execute(d.data_ptr(), ra.data_ptr(), hs)
        rb = torch.zeros(ms.ngrid * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        msd = T.MultiSyncDev(eng, pb, streams, d.data_ptr(), offs, rb.data_ptr(), 64, hs)
        sets.append((ref, ra, rb, msd, pa, pb, d, streams))
    for ref, ra, rb, msd, pa, pb, d, streams in sets:
        got = msd.collect()
        torch.cuda.synchronize()
        assert not msd.fellback
        _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES), "small plan")
        assert pb.final_codes().tolist() == pa.final_codes().tolist()
    for s_ in sets:
        s_[4].close()
        s_[5].close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [64, 256])
def test_packed_ingest_equals_the_byte_path(T, eng, chunk):
    """tgpu_pack_bits + tgpu_sync_multi_launch_packed (the capture crosses PCIe one bit per bit; the front end starts behind
    its own bytes -> bits step) against tgpu_sync_multi_launch on the bytes: eight channels of different cells, lengths and
    lead-ins (so that the channels' bit offsets and slot alignments differ), damaged training sequences (the per-position
    pass runs on packed input too), spurious sequences below offset 21, a channel with next to nothing in it, one that
    ends inside a burst.  Same events, counts, bitmaps, every record byte, final codes; no fallback either way"""
    import torch
    from test_stream_sync_cpu import SEQ_N, SEQ_P
    hs = torch.cuda.current_stream().cuda_stream
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (262, 42, 2), (505, 1, 60), (208, 10, 5), (222, 99, 7)]
    rng = np.random.default_rng(808)
    streams = []
    for c, cell in enumerate(cells):
        nsl = int(rng.integers(200, 3000)) if c != 3 else 2
        st, _ = _mix_stream(T, nsl, 2900 + c, cell, ber=0.02)
        st = np.concatenate([rng.integers(0, 2, int(rng.integers(0, 40))).astype(np.uint8), st])      # (another lead-in per channel)
        lead = len(st) - 700 - 510 * (nsl + 1) + 510
        if nsl > 100:
            for j in (17, 18, 41, nsl - 2, nsl - 1, 64, 72):
                st[lead + 510 * j + (214 if j % 8 == 0 else 244) + 2] ^= 1
            for j in (30, 55):
                seq = (SEQ_N, SEQ_P)[j % 2]
                o = lead + 510 * j + int(rng.integers(0, 21))
                st[o:o + len(seq)] = seq
        if c == 5:
            st = st[:len(st) - 700 - 200]
        streams.append(np.ascontiguousarray(st))
    d, offs, ntot = _multi_batch(T, streams)
    # the packed buffer: every channel packed on its own, at a 16-byte aligned place; offsets in bits
    poffs, o = [], 0
    for st in streams:
        poffs.append(o)
        o += ((len(st) + 7) // 8 + 15) & ~15
    pbuf = np.zeros(o + 1024, np.uint8)
    for st, f in zip(streams, poffs):
        p, bad = T.pack_bits(st, nthreads=3)
        assert bad == 0
        pbuf[f:f + len(p)] = p
    dp = torch.from_numpy(pbuf).cuda()
    pa, pb = T.Plan(eng, ntot, len(cells)), T.Plan(eng, ntot, len(cells))
    ra = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    rb = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    a = T.MultiSyncDev(eng, pa, streams, d.data_ptr(), offs, ra.data_ptr(), chunk, hs)
    ref = a.collect()
    b = T.MultiSyncDev(eng, pb, streams, dp.data_ptr(), [8 * f for f in poffs], rb.data_ptr(), chunk, hs, packed=True)
    got = b.collect()
    torch.cuda.synchronize()
    assert not a.fellback and not b.fellback and a.ngrid == b.ngrid
    assert sum(len(x["events"]) for x in ref) > 100
    _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES), "packed ingest")
    assert pa.final_codes().tolist() == pb.final_codes().tolist()
    pa.close()
    pb.close()
    # a stream the device walk hands to the host walks (an inserted byte: the synchroniser re-locks beside the grid): the packed
    # batch takes the same way back -- the host walks read the unpacked host bytes, the decode needs no stream any more
    base, _ = _mix_stream(T, 600, 77, (262, 42, 1), ber=0.0)
    lead = 100 + 510
    bent = np.concatenate([base[:lead + 510 * 33 + 17], [1], base[lead + 510 * 33 + 17:]]).astype(np.uint8)
    two = [bent, base]
    d2, offs2, ntot2 = _multi_batch(T, two)
    poffs2, o = [], 0
    for st in two:
        poffs2.append(o)
        o += ((len(st) + 7) // 8 + 15) & ~15
    pbuf2 = np.zeros(o + 1024, np.uint8)
    for st, f in zip(two, poffs2):
        pbuf2[f:f + (len(st) + 7) // 8] = T.pack_bits(st)[0]
    dp2 = torch.from_numpy(pbuf2).cuda()
    pa, pb = T.Plan(eng, ntot2, 2), T.Plan(eng, ntot2, 2)
    ra = torch.zeros(ntot2 * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    rb = torch.zeros(ntot2 * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    a = T.MultiSyncDev(eng, pa, two, d2.data_ptr(), offs2, ra.data_ptr(), 64, hs)
    ref = a.collect()
    b = T.MultiSyncDev(eng, pb, two, dp2.data_ptr(), [8 * f for f in poffs2], rb.data_ptr(), 64, hs, packed=True)
    got = b.collect()
    torch.cuda.synchronize()
    assert a.fellback and b.fellback
    _same_batch_outcome(T, ref, got, ra.cpu().numpy().reshape(-1, T.REC_BYTES), rb.cpu().numpy().reshape(-1, T.REC_BYTES), "packed ingest, handed over")
    pa.close()
    pb.close()
