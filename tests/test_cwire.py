"""The compact transport form of a decoded batch (csrc/tg_cwire.h): what a rank hands to the gather.

CPU tests: the host packer / reader (tgpu_cwire_pack, _info, _chan, _foreach, _expand) round-trip ANY 40-byte wire
records exactly, the sizes are what the format says.  GPU tests: the device form (k_cw_*) equals the host packer byte
for byte on real decoded batches, and unpacks to the full records the batch left in HBM (which the other tests hold
against the oracle); the metric workload at full size stays under 32 MB per rank and step.

Reference: delivered bursts only reach tetra_burst_rx_cb() (phy/tetra_burst_sync.c:113-150); the upper MAC acts on
CRC-good blocks (tetra_upper_mac.c:480-488).
"""
import numpy as np
import pytest

import oraclelib as O


def _T():
    import osmo_tetra_amd as T
    T.lib()
    return T


def _random_batch(T, rng, ngrid, gbase, ncls, p_types=(.4, .3, .2, .05, .05), p_badcrc=0.1, p_flag=0.02, p_junk=0.02, p_deliv=0.9):
    types = rng.choice([0, 1, 3, 2, 0xff], ngrid, p=list(p_types)).astype(np.uint8)
    rec = np.zeros((ngrid, T.REC_BYTES), np.uint8)
    rec[:, 0] = types
    rec[:, 32:46] = rng.integers(0, 2, (ngrid, 14))
    rec[:, 48:48 + 268] = rng.integers(0, 2, (ngrid, 268))
    two = (types == 1) | (types == 3)
    rec[two, 48 + 124:176] = 0
    rec[types == 3, 48 + 60:176] = 0
    crc = np.full((ngrid, 2), 0x1d0f, np.uint16)
    for k in range(2):
        bad = rng.random(ngrid) < p_badcrc
        crc[bad, k] = rng.integers(0, 65536, int(bad.sum()))
    crc[types == 0, 1] = 0
    rec[:, 4:8] = crc.view(np.uint8).reshape(ngrid, 4)
    rec[rng.random(ngrid) < p_flag, 1] = 1
    wire = T.wire_pack(rec)
    junk = rng.random(ngrid) < p_junk                  # records that are not in the kernels' canonical form at all
    wire[junk] = rng.integers(0, 256, (int(junk.sum()), T.WIRE_BYTES))
    inchan = np.zeros(ngrid, bool)
    for gb, nc in zip(gbase, ncls):
        inchan[gb:gb + nc] = True
    dl = (rng.random(ngrid) < p_deliv) & inchan
    bits = np.zeros((ngrid + 31) // 32, np.uint32)
    idx = np.flatnonzero(dl)
    np.bitwise_or.at(bits, idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))
    return wire, bits, dl


def test_host_pack_and_reader_round_trip_any_wire_records():
    """tgpu_cwire_pack -> tgpu_cwire_info / _chan / _foreach / _expand: every delivered slot's 40 bytes come back exactly
    (good bursts, failed CRCs, flags, ignored types, random bytes), undelivered slots come back as 0xff, the bitmap and
    the per-channel counts are right; empty channels in the middle and at the end; a grid that ends inside a word"""
    T = _T()
    rng = np.random.default_rng(1)
    for ngrid, gbase, ncls in ((5000, [0, 2016, 2016, 3968], [2000, 0, 1900, 1032]), (31, [0], [31]), (4096, [0, 2048, 4096], [2048, 2048, 0]),
                               (64, [0, 32], [1, 0])):
        wire, bits, dl = _random_batch(T, rng, ngrid, gbase, ncls)
        cw = T.cwire_pack(wire, bits, ngrid, gbase, ncls)
        inf = T.cwire_info(cw)
        assert inf["ngrid"] == ngrid and inf["nchan"] == len(gbase) and inf["ndelivered"] == int(dl.sum()) and inf["total_bytes"] == len(cw)
        assert len(cw) <= T.cwire_bound(ngrid, len(gbase))
        for (g, n, d), gb, nc in zip(inf["chans"], gbase, ncls):
            assert (g, n) == (gb, nc) and d == int(dl[gb:gb + nc].sum())
        w2, b2 = T.cwire_expand(cw)
        assert (b2 == bits).all() and (w2[dl] == wire[dl]).all() and (w2[~dl] == 0xff).all()
        assert T.cwire_count(cw) == int(dl.sum())
        # a truncated or damaged buffer is refused, not misread
        with pytest.raises(T.TgpuError):
            T.cwire_info(cw[:len(cw) - 1])
        bad = cw.copy()
        bad[0] ^= 1
        with pytest.raises(T.TgpuError):
            T.cwire_count(bad)


def test_record_sizes_of_the_mix():
    """all bursts good: 36 / 33 / 25 bytes per NORM_1 / NORM_2 / SYNC record, 33.5 per burst of the SB+NDB mix; a failed
    CRC or a flag costs 41"""
    T = _T()
    rng = np.random.default_rng(2)
    n = 8000
    wire, bits, dl = _random_batch(T, rng, n, [0], [n], p_types=(.5, .375, .125, 0, 0), p_badcrc=0, p_flag=0, p_junk=0, p_deliv=1.0)
    ty = wire[:, 0]
    cw = T.cwire_pack(wire, bits, n, [0], [n])
    hdr = np.frombuffer(cw[:32].tobytes(), np.uint32)
    payload = len(cw) - int(hdr[7])
    want = 36 * int((ty == 0).sum()) + 33 * int((ty == 1).sum()) + 25 * int((ty == 3).sum())
    assert want <= payload <= want + 3 * (n // 32)              # (a pad of 0..3 bytes per bitmap word)
    assert abs(payload / n - 33.5) < 0.2
    wire2 = wire.copy()
    w9 = wire2[:, 36:40].view(np.uint32)
    w9[:, 0] ^= 1 << 13                                          # every first block's CRC word off by one bit
    cw2 = T.cwire_pack(wire2, bits, n, [0], [n])
    assert len(cw2) - int(hdr[7]) >= 41 * n
    w3, _ = T.cwire_expand(cw2)
    assert (w3 == wire2).all()


def test_reader_refuses_damaged_buffers_without_touching_memory_beyond_them():
    """the reader runs on a buffer that came over a link: delivered bits at or above ngrid, a size that is not a whole
    number of dwords behind the tables, a channel table outside the grid, random damage anywhere -- every call returns
    (an error or a consistent parse) and writes nothing outside ngrid x 40 bytes of the caller's array (guard rows)"""
    import ctypes as C
    T = _T()
    L = T.lib()
    u8p = C.POINTER(C.c_uint8)
    rng = np.random.default_rng(5)

    def expand_guarded(cw, ngrid):
        guard = 64
        arr = np.full((ngrid + 2 * guard) * T.WIRE_BYTES, 0xa5, np.uint8)
        body = arr[guard * T.WIRE_BYTES:(guard + ngrid) * T.WIRE_BYTES]
        rc = L.tgpu_cwire_expand(cw.ctypes.data_as(u8p), len(cw), body.ctypes.data_as(u8p), None)
        assert (arr[:guard * T.WIRE_BYTES] == 0xa5).all() and (arr[(guard + ngrid) * T.WIRE_BYTES:] == 0xa5).all(), "wrote outside the grid"
        return rc

    # the advisor's case: ngrid = 33, bit 63 of the bitmap set
    ngrid, gbase, ncls = 33, [0], [33]
    wire, bits, dl = _random_batch(T, rng, ngrid, gbase, ncls, p_deliv=1.0)
    cw = T.cwire_pack(wire, bits, ngrid, gbase, ncls)
    hdr = np.frombuffer(cw[:32].tobytes(), np.uint32)
    o_bits = int(hdr[5])
    bad = cw.copy()
    bad[o_bits + 7] |= 0x80
    assert expand_guarded(bad, ngrid) != 0
    with pytest.raises(T.TgpuError):
        T.cwire_info(bad)
    # a total that leaves the record area a non-multiple of 4 (with the bytes to back it)
    bad = np.concatenate([cw, np.zeros(8, np.uint8)])
    bad[12:16] = np.frombuffer(np.uint32(len(cw) + 1).tobytes(), np.uint8)
    assert expand_guarded(bad, ngrid) != 0 and L.tgpu_cwire_foreach(bad.ctypes.data_as(u8p), len(bad), None, None) < 0
    # a channel table that leaves the grid / runs backwards
    ngrid, gbase, ncls = 4096, [0, 2048], [2048, 2048]
    wire, bits, dl = _random_batch(T, rng, ngrid, gbase, ncls)
    cw = T.cwire_pack(wire, bits, ngrid, gbase, ncls)
    for word, val in ((32 + 16 + 4, 4096), (32 + 16, 0x7fffffe0), (32 + 16, 32), (32 + 8, 0xffffffff), (32 + 16 + 12, 0xfffffff0)):
        bad = cw.copy()
        bad[word:word + 4] = np.frombuffer(np.uint32(val).tobytes(), np.uint8)
        with pytest.raises(T.TgpuError):
            T.cwire_info(bad)
    # random damage: single bytes, whole dwords in the tables, truncation -- never a write outside, never a crash
    refused = 0
    for it in range(3000):
        bad = cw.copy()
        kind = it % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, len(bad)))] = rng.integers(0, 256)
        elif kind == 1:
            o = 4 * int(rng.integers(0, int(hdr_o_rec(cw)) // 4))
            bad[o:o + 4] = rng.integers(0, 256, 4)
        else:
            bad = bad[:int(rng.integers(0, len(bad)))].copy()
        rc = expand_guarded(bad, ngrid) if len(bad) >= 32 and np.frombuffer(bad[8:12].tobytes(), np.uint32)[0] == ngrid else \
            L.tgpu_cwire_foreach(bad.ctypes.data_as(u8p), len(bad), None, None)
        refused += rc != 0 if kind != 2 else rc < 0 or rc != int(dl.sum())
    assert refused > 1000


def hdr_o_rec(cw):
    return np.frombuffer(cw[28:32].tobytes(), np.uint32)[0]


# ----------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (the HIP path has no fallback)")
    return _T()


@pytest.fixture(scope="module")
def eng(T):
    e = T.Engine(0)
    yield e
    e.close()


@gpu
def test_device_compaction_equals_the_host_packer(T, eng):
    """tgpu_wire_compact (k_cw_sizes / _scan / _emit) on arbitrary wire records and bitmaps == tgpu_cwire_pack byte for
    byte: every record form, escape records, empty channels, grids that end inside a word / a block, one slot; the total
    reported on the device; a buffer that is too small is reported, not overrun"""
    import torch
    rng = np.random.default_rng(5)
    cases = [(5000, [0, 2016, 2016, 3968], [2000, 0, 1900, 1032]), (31, [0], [31]), (4096, [0, 2048, 4096], [2048, 2048, 0]),
             (1, [0], [1]), (1024, [0], [1024]), (1025, [0, 1024], [1000, 1]), (70_000, [0, 30016], [30000, 39984])]
    for ngrid, gbase, ncls in cases:
        for kw in (dict(), dict(p_badcrc=0, p_flag=0, p_junk=0, p_types=(.5, .375, .125, 0, 0)), dict(p_deliv=0.0), dict(p_deliv=1.0, p_junk=1.0)):
            wire, bits, dl = _random_batch(T, rng, ngrid, gbase, ncls, **kw)
            want = T.cwire_pack(wire, bits, ngrid, gbase, ncls)
            cap = T.cwire_bound(ngrid, len(gbase))
            d_w = torch.from_numpy(wire.reshape(-1)).cuda()
            d_b = torch.from_numpy(bits.view(np.int32)).cuda()
            d_out = torch.full((cap + 64,), 0xAB, dtype=torch.uint8, device="cuda")
            d_tot = torch.zeros(2, dtype=torch.int32, device="cuda")
            T.wire_compact(eng, d_w.data_ptr(), d_b.data_ptr(), ngrid, gbase, ncls, d_out.data_ptr(), cap, d_tot.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            tot = d_tot.cpu().numpy().view(np.uint32)
            assert int(tot[0]) == len(want) and int(tot[1]) == int(dl.sum()), (ngrid, kw)
            got = d_out.cpu().numpy()
            assert (got[:len(want)] == want).all(), (ngrid, kw, int(np.flatnonzero(got[:len(want)] != want)[0]))
            assert (got[cap:] == 0xAB).all()
            # too small a buffer: the needed size comes back, the count word says "nothing written", nothing behind cap is touched
            if int(tot[1]) > 200:
                small = (len(want) - 100) & ~15
                d_out.fill_(0xAB)
                T.wire_compact(eng, d_w.data_ptr(), d_b.data_ptr(), ngrid, gbase, ncls, d_out.data_ptr(), small, d_tot.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                tot = d_tot.cpu().numpy().view(np.uint32)
                assert int(tot[0]) == len(want) and int(tot[1]) == 0xffffffff
                assert (d_out.cpu().numpy()[small:] == 0xAB).all()


def _mix_stream(T, n, seed, cell, ber, damaged=0.01):
    rng = np.random.default_rng(seed)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    mcc, mnc, cc = cell
    code = O.scramb_get_init(mcc, mnc, cc)
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=seed, scramb_init=code, mcc=mcc, mnc=mnc, cc=cc, ber=ber)
    y = slots[0, 214:252].tolist()
    for i in np.flatnonzero(rng.random(n) < damaged) + 1:
        off = 214 if slots[i, 214:252].tolist() == y else 244
        slots[i, off + int(rng.integers(0, 22))] ^= 1
    return np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)]), code


def _batch(T, streams):
    import torch
    offs, o = [], 0
    for st in streams:
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    return torch.from_numpy(buf).cuda(), offs, sum((len(st) // 510 + 32) for st in streams)


def _run_with_cwire(T, eng, streams, wire_only=False, cap=None, guard=0):
    import torch
    hs = torch.cuda.current_stream().cuda_stream
    d, offs, ntot = _batch(T, streams)
    plan = T.Plan(eng, ntot, len(streams))
    d_rec = torch.zeros(ntot * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_wire = torch.full((ntot * T.WIRE_BYTES,), 0xff, dtype=torch.uint8, device="cuda")
    cap = cap or T.cwire_bound(ntot, len(streams))
    d_cw = torch.full((cap + guard,), 0xCD, dtype=torch.uint8, device="cuda")
    plan.set_wire(d_wire.data_ptr())
    plan.set_wire_only(wire_only)
    plan.set_cwire(d_cw.data_ptr(), cap)
    msd = T.MultiSyncDev(eng, plan, streams, d.data_ptr(), offs, d_rec.data_ptr(), 64, hs)
    outs = msd.collect()
    torch.cuda.synchronize()
    res = dict(outs=outs, ngrid=msd.ngrid, fellback=msd.fellback, nbytes=msd.cwire_bytes, needed=msd.cwire_needed,
               rec=d_rec.cpu().numpy().reshape(-1, T.REC_BYTES)[:max(msd.ngrid, 1)],
               wire=d_wire.cpu().numpy().reshape(-1, T.WIRE_BYTES)[:max(msd.ngrid, 1)], cw=d_cw.cpu().numpy(), codes=plan.final_codes())
    plan.close()
    return res


@gpu
def test_compact_wire_of_a_decoded_batch_unpacks_to_its_records(T, eng):
    """a device-walk batch (eight channels of different cells and lengths, damaged training sequences, 4 % payload errors:
    good and failed CRCs side by side) with tgpu_plan_set_cwire: the buffer equals the host packer's on the batch's wire
    records and delivered bitmaps; its reader hands back exactly the delivered bursts, whose records (tgpu_wire_unpack)
    are the decoded 320-byte records field for field, and those the oracle's decode of the same bytes"""
    from test_gpu_parity import check_against_oracle
    cells = [(262, 42, 1), (901, 77, 9), (234, 14, 33), (1, 2, 3), (262, 42, 2), (505, 1, 60), (208, 10, 5), (222, 99, 7)]
    rng = np.random.default_rng(77)
    streams, codes = [], []
    for c, cell in enumerate(cells):
        nsl = int(rng.integers(300, 2500)) if c != 3 else 2
        st, code = _mix_stream(T, nsl, 5100 + c, cell, ber=0.04 if c % 2 else 0.0)
        streams.append(st)
        codes.append(code)
    for wire_only in (False, True):
        r = _run_with_cwire(T, eng, streams, wire_only)
        assert not r["fellback"] and r["nbytes"] > 0
        outs, ngrid = r["outs"], r["ngrid"]
        bits = np.zeros(ngrid // 32, np.uint32)
        gbase, ncls = [], []
        for o in outs:
            gbase.append(o["grid_base"])
            ncls.append(o["ngrid"])
            gb = np.asarray(o["grid_bits"], np.uint32)
            bits[o["grid_base"] // 32:o["grid_base"] // 32 + len(gb)] = gb
        cw = r["cw"][:r["nbytes"]]
        want = T.cwire_pack(r["wire"], bits, ngrid, gbase, ncls)
        assert len(cw) == len(want) and (cw == want).all()
        inf = T.cwire_info(cw)
        assert [x[2] for x in inf["chans"]] == [o["nslots"] for o in outs] and inf["ndelivered"] == sum(o["nslots"] for o in outs)
        w2, b2 = T.cwire_expand(cw)
        assert (b2 == bits).all()
        if wire_only:
            continue
        nbad = 0
        for c, o in enumerate(outs):
            idx = o["grid_base"] + T.grid_indices(o)
            if not len(idx):
                continue
            back = T.wire_unpack(w2[idx], idx.tolist(), [codes[c]] * len(idx))
            pa, pb = T.parse_records(back), T.parse_records(r["rec"][idx])
            ty = pb["type"].astype(np.uint8)
            two = ty != 0
            for k in ("type", "flags", "bbk", "slot"):
                assert (np.asarray(pa[k]) == np.asarray(pb[k])).all(), (c, k)
            assert (pa["crc"][:, 0] == pb["crc"][:, 0]).all() and (pa["crc"][two, 1] == pb["crc"][two, 1]).all()
            assert (pa["crc_ok"][:, 0] == pb["crc_ok"][:, 0]).all() and (pa["crc_ok"][two, 1] == pb["crc_ok"][two, 1]).all()
            n1, n2, sb = ty == 0, ty == 1, ty == 3
            assert (pa["bits1"][n1] == pb["bits1"][n1]).all()
            assert (pa["bits1"][n2][:, :124] == pb["bits1"][n2][:, :124]).all() and (pa["bits2"][two] == pb["bits2"][two]).all()
            assert (pa["bits1"][sb][:, :60] == pb["bits1"][sb][:, :60]).all()
            nbad += int((pb["crc_ok"][:, 0] == 0).sum())
            # ... and the records are the oracle's decode of the same bytes
            gi = T.grid_indices(o)
            st = streams[c]
            slots = st[o["anchor"]:o["anchor"] + 510 * (int(gi[-1]) + 1)].reshape(-1, 510)[gi]
            known = pb["code"] == codes[c]          # (a noisy channel's first SB1 may fail: bursts before the first good one keep the carry-in code 0)
            assert known.mean() > 0.9
            check_against_oracle(T, r["rec"][idx][known], ty[known], slots[known], codes[c], use_acc=1)
        assert nbad > 50          # (escape records were in play)


@gpu
def test_a_cwire_buffer_that_is_too_small_is_left_alone_and_the_batch_stays_valid(T, eng):
    """tgpu_plan_set_cwire() takes any capacity >= 4096: one that does not even hold the tables (header, channel table,
    bitmap, block table) is not written at all, one that holds the tables but not the records carries no records; either
    way collect succeeds, the records are the full-capacity run's, cwire_bytes is 0 and cwire_needed says what it takes"""
    rng = np.random.default_rng(78)
    streams = [_mix_stream(T, 30000 + 500 * c, 5200 + c, (262, 42 + c, 1 + c), 0.0)[0] for c in range(3)]
    full = _run_with_cwire(T, eng, streams)
    assert not full["fellback"] and full["nbytes"] > 0 and full["needed"] == 0
    hdr = np.frombuffer(full["cw"][:32].tobytes(), np.uint32)
    o_rec = int(hdr[7])
    assert o_rec > 4096 + 16
    for cap in (4096, (o_rec + 15) & ~15, (o_rec + 4096) & ~15, (full["nbytes"] - 16) & ~15):
        r = _run_with_cwire(T, eng, streams, cap=cap, guard=1 << 20)
        assert not r["fellback"] and r["nbytes"] == 0 and r["needed"] >= full["nbytes"], cap
        assert (r["cw"][cap:] == 0xCD).all(), "wrote behind a %d-byte cwire buffer" % cap
        if cap < o_rec + 16:
            assert (r["cw"] == 0xCD).all()
        assert (r["rec"] == full["rec"]).all() and (r["wire"] == full["wire"]).all()
        assert [o["nslots"] for o in r["outs"]] == [o["nslots"] for o in full["outs"]]
    r = _run_with_cwire(T, eng, streams, cap=(full["nbytes"] + 15) & ~15, guard=4096)       # exactly enough
    assert r["nbytes"] == full["nbytes"] and (r["cw"][:r["nbytes"]] == full["cw"][:r["nbytes"]]).all()


@gpu
def test_compact_wire_of_the_metric_workload_stays_under_32_MB(T, eng):
    """the default bench line's step (8 channels x 125 000 slots, 1 % damaged training sequences, no payload errors):
    bytes per rank and step of the compact form <= 32 MB (40 MB as grid wire records: more than one xGMI link moves in a
    step), every delivered burst comes back out of it, equal to the batch's own wire record"""
    streams = []
    for c in range(8):
        st, _ = _mix_stream(T, 125_000, 9100 + c, (262, 42 + c, 1 + c), ber=0.0)
        streams.append(st)
    r = _run_with_cwire(T, eng, streams)
    assert not r["fellback"]
    ndel = sum(o["nslots"] for o in r["outs"])
    assert ndel > 930_000
    assert r["nbytes"] <= 32_000_000, r["nbytes"]
    assert r["nbytes"] / ndel < 33.8
    cw = r["cw"][:r["nbytes"]]
    w2, b2 = T.cwire_expand(cw)
    dl = np.zeros(r["ngrid"], bool)
    for o in r["outs"]:
        dl[o["grid_base"] + T.grid_indices(o)] = True
    assert int(dl.sum()) == ndel == T.cwire_info(cw)["ndelivered"]
    assert (w2[dl] == r["wire"][dl]).all() and (w2[~dl] == 0xff).all()
