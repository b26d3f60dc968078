"""Traffic blocks of a decoded batch on the device (csrc/tg_traffic.hip, tgpu_plan_set_traffic /
tgpu_plan_traffic) against the oracle's restatement of lower_mac/tetra_lower_mac.c:194-241 and against the host path
(the channel API's callback hands out the same type-4 bits; tgpu_traffic_block() makes the same 690-word block).

The caller's byte per slot has tgpu_gsmtap_batch()'s shape: bit 0 = traffic burst (cur_burst.is_traffic), bit 1 = its
second block was stolen (cur_burst.blk2_stolen)."""
import numpy as np
import pytest

import oraclelib as O
import synth

pytestmark = pytest.mark.gpu

F_TRAFFIC, F_BLK1_STOLEN = 0x02, 0x04


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (the HIP path has no fallback)")
    import osmo_tetra_amd as T
    T.lib()
    return T


@pytest.fixture(scope="module")
def eng(T):
    e = T.Engine(0)
    yield e
    e.close()


def expected(slots, types, codes, traffic):
    """the oracle's side: type-4 bits (descrambled type-5 bits of the dumped block), dump block, bit count per slot"""
    n = len(types)
    lens = np.zeros(n, np.uint16)
    t4 = np.zeros((n, 432), np.uint8)
    full = (types == O.TRAIN_NORM_1) & ((traffic & 1) == 1)
    half = ((types == O.TRAIN_NORM_2) | (types == O.TRAIN_SYNC)) & ((traffic & 1) == 1) & ((traffic & 2) == 0)
    seqs = {int(c): O.scramb_seq(int(c), 432) for c in np.unique(codes)}
    seq = np.stack([seqs[int(c)] for c in codes])
    t4[full, :216] = slots[full, 14:230]                   # SCH/F = BLK1 || BLK2 (phy/tetra_burst.c:367-368)
    t4[full, 216:] = slots[full, 282:498]
    t4[full] ^= seq[full]
    t4[half, :216] = slots[half, 282:498] ^ seq[half, :216]    # BLK2 / SB2 share the offset (phy/tetra_burst.c:38,46)
    lens[full], lens[half] = 432, 216
    return t4, lens, full, half


def dump_blocks(t4, lens, idx):
    return np.stack([O.traffic_block(t4[i, :lens[i]]) for i in idx]) if len(idx) else np.zeros((0, 690), np.int16)


def test_traffic_blocks_of_a_slot_batch_against_the_oracle(T, eng):
    """120 000 slots (SB / NORM_1 / NORM_2 mixed, three cells, 2 % payload errors), a third of them marked as traffic, a
    third of those with a stolen second block: type-4 bits, dump blocks and lengths of every slot equal the oracle's
    (10^5 dumped blocks byte for byte, every one of them also through the library's host function); the records are
    the untouched run's except for the flags and the dumped block's crc_ok; the stage on its own (tgpu_plan_traffic)
    after a plain execute gives the same bytes"""
    import torch
    n = 120_000
    rng = np.random.default_rng(41)
    types = rng.choice([O.TRAIN_SYNC, O.TRAIN_NORM_1, O.TRAIN_NORM_2], n, p=[0.125, 0.5, 0.375]).astype(np.uint8)
    types[0] = O.TRAIN_SYNC
    chan = np.sort(rng.integers(0, 3, n)).astype(np.uint32)         # (a batch is channel-major)
    chan_codes = np.array([O.scramb_get_init(262, 42 + c, 1 + c) for c in range(3)], np.uint32)
    slots = np.zeros((n, 510), np.uint8)
    for c in range(3):
        m = chan == c
        slots[m] = T.synth_slots(types[m], seed=100 + c, scramb_init=int(chan_codes[c]), mcc=262, mnc=42 + c, cc=1 + c, ber=0.02)
    traffic = rng.choice([0, 1, 3, 2], n, p=[0.1, 0.6, 0.25, 0.05]).astype(np.uint8)     # (2: "stolen" without "traffic" means nothing)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_tr = torch.from_numpy(traffic).cuda()
    hs = torch.cuda.current_stream().cuda_stream

    def run(with_stage):
        d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
        d_t4 = torch.full((n * 432,), 0xEE, dtype=torch.uint8, device="cuda")
        d_blk = torch.full((n * 690,), 0x7777, dtype=torch.int16, device="cuda")
        d_len = torch.full((n,), 0x5555, dtype=torch.int16, device="cuda")
        plan = T.Plan(eng, n, 3)
        plan.load(np.arange(n, dtype=np.uint64) * 510, types, chan, chan_codes)     # carry-in codes: the cells' own
        if with_stage == "set":
            plan.set_traffic(d_tr.data_ptr(), d_t4.data_ptr(), d_blk.data_ptr(), d_len.data_ptr())
        plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), hs)
        if with_stage == "call":
            plan.traffic(d_tr.data_ptr(), d_rec.data_ptr(), d_t4.data_ptr(), d_blk.data_ptr(), d_len.data_ptr(), hs)
        torch.cuda.synchronize()
        out = (d_rec.cpu().numpy().reshape(n, -1), d_t4.cpu().numpy().reshape(n, 432), d_blk.cpu().numpy().reshape(n, 690),
               d_len.cpu().numpy().view(np.uint16))
        plan.close()
        return out

    rec0, _, _, _ = run(None)
    rec, t4, blk, lens = run("set")
    p0, p = T.parse_records(rec0), T.parse_records(rec)
    codes = p0["code"]                                    # the code in force for each slot's BBK / blocks, as the batch used it
    assert (codes == chan_codes[chan]).mean() > 0.99      # (a failed SB1 keeps the carry-in code, which is the cell's here)
    want4, wlen, full, half = expected(slots, types, codes, traffic)
    assert int(full.sum() + half.sum()) >= 100_000 * 0.6 and half.sum() > 10_000 and full.sum() > 30_000
    assert (lens == wlen).all()
    dumped = np.flatnonzero(wlen)
    assert (t4[dumped] == want4[dumped]).all()
    wblk = dump_blocks(want4, wlen, dumped)
    assert (blk[dumped] == wblk).all()
    for i in dumped[:: max(1, len(dumped) // 3000)]:      # the library's host function on the device's own type-4 bits
        assert (T.traffic_block(t4[i, :lens[i]]) == blk[i]).all()
    quiet = np.flatnonzero(wlen == 0)                     # nothing of the other slots' rows is touched
    assert (t4[quiet] == 0xEE).all() and (blk[quiet] == 0x7777).all()
    # records: flags and the dumped block's crc_ok, nothing else
    n2 = (types == O.TRAIN_NORM_2) & ((traffic & 1) == 1)
    wflags = p0["flags"] | np.where(full | half, F_TRAFFIC, 0).astype(np.uint8) | np.where(n2, F_BLK1_STOLEN, 0).astype(np.uint8)
    assert (p["flags"] == wflags).all()
    wok = p0["crc_ok"].copy()
    wok[full, 0] = 0
    wok[half, 1] = 0
    assert (p["crc_ok"] == wok).all() and p0["crc_ok"][full, 0].mean() > 0.5
    same = np.ones(T.REC_BYTES, bool)
    same[1:4] = False
    assert (rec[:, same] == rec0[:, same]).all()
    # the stage on its own, after the batch
    rec_b, t4_b, blk_b, lens_b = run("call")
    assert (rec_b == rec).all() and (t4_b == t4).all() and (blk_b == blk).all() and (lens_b == lens).all()


def _mix_stream(T, nsl, seed, cell, ber):
    rng = np.random.default_rng(seed)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, nsl // 8 + 1)[:nsl]
    mcc, mnc, cc = cell
    code = O.scramb_get_init(mcc, mnc, cc)
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=seed, scramb_init=code, mcc=mcc, mnc=mnc, cc=cc, ber=ber)
    y = slots[0, 214:252].tolist()
    for i in np.flatnonzero(rng.random(nsl) < 0.01) + 1:
        off = 214 if slots[i, 214:252].tolist() == y else 244
        slots[i, off + int(rng.integers(0, 22))] ^= 1
    return np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)]), code


def test_traffic_blocks_of_a_stream_batch_with_the_walk_on_the_device(T, eng):
    """four recorded channels through tgpu_sync_multi_launch (search, walks, lists and decode on the device) with the
    traffic stage set on the plan, wire records on: delivered bursts flagged as traffic have their blocks dumped
    (== the oracle on the slot's bytes under the record's code), undelivered and unflagged grid slots have none; the
    wire records carry the flags and unpack to a record whose dumped block is not CRC-good"""
    import torch
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3)]
    streams, codes = zip(*[_mix_stream(T, 6000 + 700 * c, 7100 + c, cell, 0.01 if c & 1 else 0.0) for c, cell in enumerate(cells)])
    offs, o = [], 0
    for st in streams:
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    ntot = sum(len(st) // 510 + 32 for st in streams)
    d = torch.from_numpy(buf).cuda()
    rng = np.random.default_rng(9)
    traffic = rng.choice([0, 1, 3], ntot, p=[0.4, 0.4, 0.2]).astype(np.uint8)
    d_tr = torch.from_numpy(traffic).cuda()
    d_rec = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_wire = torch.full((ntot * T.WIRE_BYTES,), 0xff, dtype=torch.uint8, device="cuda")
    d_t4 = torch.full((ntot * 432,), 0xEE, dtype=torch.uint8, device="cuda")
    d_blk = torch.zeros(ntot * 690, dtype=torch.int16, device="cuda")
    d_len = torch.full((ntot,), 0x5555, dtype=torch.int16, device="cuda")
    plan = T.Plan(eng, ntot, len(streams))
    plan.set_wire(d_wire.data_ptr())
    plan.set_traffic(d_tr.data_ptr(), d_t4.data_ptr(), d_blk.data_ptr(), d_len.data_ptr())
    msd = T.MultiSyncDev(eng, plan, list(streams), d.data_ptr(), offs, d_rec.data_ptr(), 64, torch.cuda.current_stream().cuda_stream)
    outs = msd.collect()
    torch.cuda.synchronize()
    assert not msd.fellback
    rec = d_rec.cpu().numpy().reshape(ntot, -1)
    wire = d_wire.cpu().numpy().reshape(ntot, -1)
    t4 = d_t4.cpu().numpy().reshape(ntot, 432)
    blk = d_blk.cpu().numpy().reshape(ntot, 690)
    lens = d_len.cpu().numpy().view(np.uint16)
    delivered = np.zeros(ntot, bool)
    ndump = 0
    for c, out in enumerate(outs):
        gi = T.grid_indices(out)
        idx = out["grid_base"] + gi
        delivered[idx] = True
        p = T.parse_records(rec[idx])
        st = streams[c]
        sl = np.stack([st[out["anchor"] + 510 * int(g):out["anchor"] + 510 * int(g) + 510] for g in gi])
        want4, wlen, full, half = expected(sl, p["type"].astype(np.uint8), p["code"], traffic[idx])
        assert (lens[idx] == wlen).all()
        dm = np.flatnonzero(wlen)
        assert (t4[idx][dm] == want4[dm]).all()
        assert (blk[idx][dm] == dump_blocks(want4, wlen, dm)).all()
        ndump += len(dm)
        n2 = (p["type"] == O.TRAIN_NORM_2) & ((traffic[idx] & 1) == 1)
        assert ((p["flags"] & F_TRAFFIC) != 0).tolist() == (full | half).tolist() and ((p["flags"] & F_BLK1_STOLEN) != 0).tolist() == n2.tolist()
        assert (p["crc_ok"][full, 0] == 0).all() and (p["crc_ok"][half, 1] == 0).all()
        # the wire records carry the flags; unpacked, the dumped block is not CRC-good and everything else is the record's
        assert (wire[idx][:, 1] == p["flags"]).all()
        back = T.parse_records(T.wire_unpack(wire[idx], idx.tolist(), p["code"].tolist()))
        assert (back["flags"] == p["flags"]).all() and (back["crc_ok"][full, 0] == 0).all() and (back["crc_ok"][half, 1] == 0).all()
        assert (back["bbk"] == p["bbk"]).all()
    assert ndump > 5000
    ng = msd.ngrid                                         # (the batch's grid slots; the buffers are sized for the plan's capacity)
    assert (lens[:ng][~delivered[:ng]] == 0).all() and (t4[:ng][lens[:ng] == 0] == 0xEE).all() and (lens[ng:] == 0x5555).all()
    plan.close()


def test_traffic_stage_equals_the_channel_path(T, eng):
    """the same bursts through the channel API (host synchroniser, the callback standing in for the upper MAC: traffic from
    the AACH, 'block 2 stolen' from block 1 of every fifth burst, tetra_lower_mac.c:194-198) and through a slot batch whose
    traffic bytes are what that callback decided: the type-4 bits the callback was handed are the device stage's, block
    for block, and tgpu_traffic_block() of them is the device's dump block -- including bursts whose block 1 was stolen"""
    import torch
    rng = np.random.default_rng(31)
    cell = synth.Cell()
    aach_traffic = np.array([0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0], np.uint8)
    bursts = [synth.make_sb(rng, cell, 1, 1, 1), synth.make_sb(rng, cell, 2, 1, 1)]
    for i in range(60):
        a = aach_traffic if i % 3 else None
        bursts.append(synth.make_norm2(rng, cell.code, a) if i % 2 else synth.make_norm1(rng, cell.code, a))
    stream = np.concatenate([rng.integers(0, 2, 77).astype(np.uint8)] + bursts + [np.zeros(700, np.uint8)])
    got, state = [], {}

    def g_upper(chan, d, offset):
        if d["type"] == O.T_BBK:
            chan.set_traffic(5 if d["type1"][1] == 1 else 0)
            chan.set_blk2_stolen(False)
            state[d["burst_seq"]] = [1 if d["type1"][1] == 1 else 0, 0]
        elif d["type"] == O.T_NDB and d["blk_num"] == 1 and d["burst_seq"] % 5 == 0:
            chan.set_blk2_stolen(True)
            state[d["burst_seq"]][1] = 1
        return -1

    ch = T.Channel(eng, batch_slots=7, on_unitdata=g_upper)
    ch.feed(stream)
    ch.flush()
    recs = list(ch.records)
    ch.close()
    seqs = sorted(state)
    bursts = bursts[1:]                                    # (the first SYNC burst is the one the synchroniser locks on: not delivered)
    assert len(seqs) == len(bursts)
    traffic = np.array([state[s][0] | (state[s][1] << 1) for s in seqs], np.uint8)
    types = np.array([O.TRAIN_SYNC] + [O.TRAIN_NORM_2 if i % 2 else O.TRAIN_NORM_1 for i in range(60)], np.uint8)
    n = len(types)
    slots = np.stack([b[:510] for b in bursts])
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_tr = torch.from_numpy(traffic).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_t4 = torch.zeros(n * 432, dtype=torch.uint8, device="cuda")
    d_blk = torch.zeros(n * 690, dtype=torch.int16, device="cuda")
    d_len = torch.zeros(n, dtype=torch.int16, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([0], np.uint32))       # the cell's code is learnt from SB1
    plan.set_traffic(d_tr.data_ptr(), d_t4.data_ptr(), d_blk.data_ptr(), d_len.data_ptr())
    plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t4 = d_t4.cpu().numpy().reshape(n, 432)
    blk = d_blk.cpu().numpy().reshape(n, 690)
    lens = d_len.cpu().numpy().view(np.uint16)
    plan.close()
    dumped = [r for r in recs if r.get("traffic")]
    assert len(dumped) > 10 and any(r["type"] == O.T_NDB for r in dumped) and any(r["type"] == O.T_SCH_F for r in dumped)
    by_seq = {s: i for i, s in enumerate(seqs)}
    seen = set()
    for r in dumped:
        i = by_seq[r["burst_seq"]]
        b = np.frombuffer(r["type4"], np.uint8)
        assert lens[i] == len(b) and (t4[i, :len(b)] == b).all(), (i, r["type"])
        assert (T.traffic_block(b) == blk[i]).all()
        seen.add(i)
    assert seen == set(np.flatnonzero(lens).tolist())
    stolen = [i for i in range(n) if traffic[i] == 3 and types[i] == O.TRAIN_NORM_2]
    assert stolen and all(lens[i] == 0 for i in stolen)
