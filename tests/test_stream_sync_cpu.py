"""The closed-form stream synchroniser (tgpu_sync_walk, host half of BASELINE config 3) against the
oracle's call-by-call state machine: same sync events, same bursts handed to the demux, same ordinals.
cls=None settles every slot with tetra_find_train_seq() on the bytes, so no GPU is needed; a numpy
emulation of the GPU classification words exercises the cls fast path as well."""
import numpy as np
import pytest

import oraclelib as O
import synth

import osmo_tetra_amd as T

SEQ_N = np.array([1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0], np.uint8)
SEQ_Y = np.array([1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1], np.uint8)
SEQ_P = np.array([0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0], np.uint8)
SEQ_Q = np.array([1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1], np.uint8)
SEQ_X = np.array([1,0,0,1,1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0,0,0,1,1], np.uint8)
HEADS = [x[:22].tolist() for x in (SEQ_Y, SEQ_N, SEQ_P, SEQ_Q, SEQ_X)]


def skewed_gate(buf, c):
    """phy/tetra_burst.c:289-297 for a position c < 21: the look-ahead window is primed with in[0..19] and then fed
    in[cur + 21], so it holds the stream with in[20] missing; a sequence at c only counts if that window equals the
    first 22 bits of one of the five training sequences"""
    e = ([0] + buf[0:20].tolist() + [int(buf[21])]) if c == 0 else (buf[c - 1:20].tolist() + buf[21:c + 22].tolist())
    return e in HEADS


@pytest.fixture(scope="module", autouse=True)
def _built():
    T.build_library()


def oracle_view(stream, chunk):
    recs, events = O.run_rx(stream, chunk=chunk)
    bursts = []
    for r in recs:
        if not bursts or bursts[-1][0] != r["burst_seq"]:
            bursts.append((r["burst_seq"], r["burst_type"]))
    pos = {}
    seq = 0
    for ev, bitnum, arg in events:
        if ev == 2:
            seq += 1
            pos[seq] = bitnum
    return bursts, events, pos


def view_of(chunk):
    """TG_VIEW_OF (csrc/tg_layout.h): how far the kernels look from a slot's start, by the replay's feeds"""
    return 640 if chunk <= 64 else 832 if chunk <= 128 else 1088


def emul_cls(stream, anchor, chunk, view=None):
    """numpy statement of k_front_stream's classification words"""
    view = view_of(chunk) if view is None else view
    L = len(stream)
    n = (L - anchor) // 510 if L >= anchor + 510 else 0
    pad = np.concatenate([stream, np.zeros(1280, np.uint8)])
    out = np.zeros(n, np.uint32)
    for i in range(n):
        bs = anchor + 510 * i
        f = min(-(-(bs + 510) // chunk) * chunk, L)
        w = f - bs
        wv = min(w, view)
        buf = pad[bs:bs + 1200].copy()
        buf[wv:] = 0
        rc, off = 0xFF, 0
        for c in range(0, wv):
            t = None
            if c + 38 <= w and (buf[c:c + 38] == SEQ_Y).all():
                t = 3
            elif c + 22 <= w and (buf[c:c + 22] == SEQ_N).all():
                t = 0
            elif c + 22 <= w and (buf[c:c + 22] == SEQ_P).all():
                t = 1
            if t is None:
                continue
            if c < 21 and not skewed_gate(buf, c):
                continue
            rc, off = t, c
            break
        flags = 4 if (rc == 0xFF and w > view) else 0
        if rc == 0xFF:      # TG_CLS_NOVIEW: nothing in the rest of the view (up to the stream's end) either
            vis = min(L - bs, view)
            full = pad[bs:bs + 1200]
            anyv = False
            for c in range(21, vis):    # the first sequence that ends inside the view: 1 + type in bits 4..6 of the flags, offset in the word
                t = 3 if (c + 38 <= vis and (full[c:c + 38] == SEQ_Y).all()) else \
                    0 if (c + 22 <= vis and (full[c:c + 22] == SEQ_N).all()) else \
                    1 if (c + 22 <= vis and (full[c:c + 22] == SEQ_P).all()) else None
                if t is not None:
                    anyv, off = True, c
                    flags |= (t + 1) << 4
                    break
            if not anyv:
                flags |= 8
        out[i] = rc | (off << 8) | (flags << 24)
    return out


def emul_ysum(stream, anchor):
    """numpy statement of k_front_stream's SYNC-sequence summaries: per grid slot, where the 38-bit y
    sequence starts inside the slot's own 510 positions (first start, 0x8000 if several, 0xffff if none)"""
    L = len(stream)
    n = (L - anchor) // 510 if L >= anchor + 510 else 0
    out = np.full(n, 0xFFFF, np.uint16)
    if L < 38:
        return out
    win = np.lib.stride_tricks.sliding_window_view(stream, 38)
    starts = np.nonzero((win == SEQ_Y).all(axis=1))[0]
    for p in starts:
        if p < anchor:
            continue
        g = (p - anchor) // 510
        if g >= n or (p - anchor) % 510 + 38 > min(L - (anchor + 510 * g), 640):
            continue
        if out[g] == 0xFFFF:
            out[g] = (p - anchor) % 510
        else:
            out[g] |= 0x8000
    return out


def check(stream, chunk=64, with_cls=True):
    bursts, oev, pos = oracle_view(stream, chunk)
    res = T.sync_walk(stream, chunk=chunk)
    assert res["events"] == oev
    assert [(s[2], s[1]) for s in res["slots"]] == bursts
    assert all(pos[s[2]] == s[0] & 0xFFFFFFFF for s in res["slots"])
    # tn_adds: time steps between delivered bursts = difference of ordinals
    prev = 0
    for off, t, seq, tn in res["slots"]:
        assert tn == seq - prev
        prev = seq
    if with_cls and res["slots"]:
        anchor = min(s[0] for s in res["slots"]) % 510 + 510 * 0
        first = res["slots"][0][0]
        anchor = first - 510 * (first // 510) if False else first % 510
        # grid anchored at the first locked slot position (any slot of the grid works as anchor)
        anchor = first
        cls = emul_cls(stream, anchor, chunk)
        res2 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls)
        assert res2["events"] == oev and res2["slots"] == res["slots"]
        res3 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls, ysum=emul_ysum(stream, anchor))
        assert res3["events"] == oev and res3["slots"] == res["slots"]
        # grid mode: the same walk marking delivered grid slots in a bitmap instead of listing them
        res4 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls, ysum=emul_ysum(stream, anchor), grid=True)
        assert res4["events"] == oev and res4["ngrid"] == len(cls)
        on = [(s[0] - anchor) // 510 for s in res["slots"] if s[0] >= anchor and (s[0] - anchor) % 510 == 0
              and (s[0] - anchor) // 510 < len(cls)]
        assert T.grid_indices(res4).tolist() == on
        assert res4["noffgrid"] == len(res["slots"]) - len(on) and res4["nslots"] == len(on)
        # without the per-burst events the walk takes its steady-state fast path (power-of-two chunks)
        quiet = [e for e in oev if e[0] != 2]
        res5 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls, ysum=emul_ysum(stream, anchor), burst_events=False)
        assert res5["events"] == quiet and res5["slots"] == res["slots"]
        res6 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls, ysum=emul_ysum(stream, anchor), burst_events=False, grid=True)
        assert res6["events"] == quiet and T.grid_indices(res6).tolist() == on and res6["noffgrid"] == res4["noffgrid"]
        # ... and with k_cls_plain's bitmap (numpy statement) its steady state reads 32 slots per word of that
        res7 = T.sync_walk(stream, chunk=chunk, anchor=anchor, cls=cls, ysum=emul_ysum(stream, anchor), burst_events=False, grid=True,
                           plain=True)
        assert res7["events"] == quiet and T.grid_indices(res7).tolist() == on and res7["noffgrid"] == res4["noffgrid"]
        assert res7["nslots"] == res6["nslots"]
        for k in ("final_state", "tail_tn_adds", "burst_seq"):
            assert res5[k] == res[k] and res6[k] == res[k] and res4[k] == res[k] and res7[k] == res[k]
    return res


def test_clean_stream_and_tail_lengths():
    stream, _ = synth.frame_stream(seed=1, nframes=3)
    for cut in (0, 1, 63, 64, 65, 509, 700, 1300):
        check(stream[:len(stream) - cut] if cut else stream)
    check(stream[:900])       # never locks
    check(stream[:1100])
    check(np.zeros(5000, np.uint8))


@pytest.mark.parametrize("chunk", [64, 1, 7, 100, 128, 509, 510])
def test_chunk_sizes(chunk):
    stream, _ = synth.frame_stream(seed=2, nframes=2, lead_in=37)
    check(stream, chunk=chunk, with_cls=(chunk <= 128))


def test_chunk_above_a_slot_is_rejected():
    with pytest.raises(T.TgpuError):
        T.sync_walk(np.zeros(4000, np.uint8), chunk=511)


def test_lock_loss_relock_and_spurious_sequences():
    rng = np.random.default_rng(5)
    for trial in range(12):
        stream, slots = synth.frame_stream(seed=10 + trial, nframes=4, lead_in=int(rng.integers(0, 300)))
        s = stream.copy()
        p0 = np.flatnonzero((np.lib.stride_tricks.sliding_window_view(s, 38) == SEQ_Y).all(axis=1))[0] + 296
        nsl = len(slots)
        for _ in range(3):
            i = int(rng.integers(0, nsl))
            kind = trial % 4
            base = p0 + 510 * i
            if kind == 0:      # corrupt the training sequence -> loss of lock (NORM) / misplaced
                off = 214 if slots[i][0] == O.TRAIN_SYNC else 244
                s[base + off + int(rng.integers(0, 22))] ^= 1
            elif kind == 1:    # spurious n sequence early in the payload -> burst dropped, lock kept
                s[base + 30:base + 52] = SEQ_N
            elif kind == 2:    # spurious SYNC sequence in a payload -> "SYNC at offset ?" -> unlock
                s[base + 40:base + 78] = SEQ_Y
            else:              # a match inside the first 21 bits of a slot (skewed look-ahead filter zone)
                s[base + int(rng.integers(0, 21)):][:22] = SEQ_N
        check(s)


def test_sync_sequence_near_buffer_start_after_loss():
    """after a loss of lock the next SYNC sequence may sit below offset 21 of the new buffer"""
    stream, slots = synth.frame_stream(seed=33, nframes=3, lead_in=50)
    p0 = np.flatnonzero((np.lib.stride_tricks.sliding_window_view(stream, 38) == SEQ_Y).all(axis=1))[0] + 296
    for d in (0, 5, 20, 21, 22):
        s = stream.copy()
        s[p0 + 510 * 2 + 244 + 3] ^= 1             # slot 2 (NORM) loses lock; buffer then starts at slot 3
        s[p0 + 510 * 3 + d:p0 + 510 * 3 + d + 38] = SEQ_Y
        check(s)


def test_long_gap_slides_the_4096_byte_buffer():
    rng = np.random.default_rng(8)
    a, _ = synth.frame_stream(seed=40, nframes=1, pad=0)
    b, _ = synth.frame_stream(seed=41, nframes=2, lead_in=0)
    gap = rng.integers(0, 2, 9000).astype(np.uint8)
    check(np.concatenate([a, gap, b]), with_cls=False)
    check(np.concatenate([a, np.zeros(5003, np.uint8), b]), with_cls=False)


def test_many_sync_sequences_per_slot_with_summary():
    """several y sequences inside one grid slot (the summary's MULTI bit) while lock is lost and regained:
    the table-driven re-lock search must agree with the byte scan"""
    rng = np.random.default_rng(77)
    for trial in range(10):
        stream, slots = synth.frame_stream(seed=60 + trial, nframes=5, lead_in=int(rng.integers(0, 200)))
        s = stream.copy()
        p0 = np.flatnonzero((np.lib.stride_tricks.sliding_window_view(s, 38) == SEQ_Y).all(axis=1))[0] + 296
        for _ in range(4):
            i = int(rng.integers(1, len(slots) - 2))
            base = p0 + 510 * i
            off = 214 if slots[i][0] == O.TRAIN_SYNC else 244
            s[base + off + 2] ^= 1                              # lose lock at slot i
            for d in sorted(rng.integers(0, 470, int(rng.integers(1, 4)))):
                s[base + 510 + int(d):base + 510 + int(d) + 38] = SEQ_Y   # and litter the next slot with y
        res = check(s)
        first = res["slots"][0][0]
        ys = emul_ysum(s, first)
        assert (ys[ys != 0xFFFF] & 0x8000).any() or trial > 0


def test_fuzz_random_mutations():
    """random streams with random damage -- bit flips, inserted / deleted bytes (the slot grid shifts), spurious
    SYNC and normal training sequences, zeroed stretches -- and random read sizes: every form of the walk
    (bytes only, classification words, + SYNC summaries, without per-burst events, grid bitmap) == oracle"""
    rng = np.random.default_rng(2024)
    for trial in range(70):
        stream, _ = synth.frame_stream(seed=int(rng.integers(1, 1 << 30)), nframes=int(rng.integers(2, 5)),
                                       lead_in=int(rng.integers(0, 600)), pad=int(rng.integers(0, 900)))
        s = stream.copy()
        for _ in range(int(rng.integers(0, 6))):
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
        chunk = int(rng.choice([64, 64, 64, 32, 128, 100, 510, 7]))
        check(s, chunk=chunk, with_cls=(chunk <= 128))


def test_long_runs_take_the_bulk_path():
    """hundreds of consecutive good slots (the 32-slot bulk step of the grid walk) with lock losses, a misplaced
    burst and a backlog after each re-lock in between"""
    rng = np.random.default_rng(99)
    for trial in range(3):
        stream, slots = synth.frame_stream(seed=500 + trial, nframes=45, lead_in=int(rng.integers(0, 300)))
        s = stream.copy()
        p0 = np.flatnonzero((np.lib.stride_tricks.sliding_window_view(s, 38) == SEQ_Y).all(axis=1))[0] + 296
        for i in rng.choice(np.arange(40, len(slots) - 40), 3, replace=False):
            off = 214 if slots[int(i)][0] == O.TRAIN_SYNC else 244
            s[p0 + 510 * int(i) + off + 4] ^= 1
        s[p0 + 510 * 100 + 30:p0 + 510 * 100 + 52] = SEQ_N      # spurious n sequence: burst dropped, lock kept
        res = check(s, chunk=64)
        assert len(res["slots"]) > 300


def test_per_call_form_equals_closed_form_and_oracle():
    """tgpu_sync_walk()'s two forms -- the reference's state machine call by call (used outside feed sizes 21..296,
    or on request) and the closed form -- give the same slot table and events on damaged streams, and both are the
    oracle's; feed sizes below 21 and above 296 go through the per-call form by themselves"""
    rng = np.random.default_rng(808)
    for trial in range(6):
        stream, slots = synth.frame_stream(seed=300 + trial, nframes=5, lead_in=int(rng.integers(0, 300)))
        s = stream.copy()
        for i in rng.choice(len(slots), 4, replace=False):
            base = int(np.flatnonzero((np.lib.stride_tricks.sliding_window_view(s, 38) == SEQ_Y).all(axis=1))[0]) + 296 + 510 * int(i)
            k = int(rng.integers(0, 3))
            if k == 0:
                s[base + 244 + 3] ^= 1
            elif k == 1:
                s[base + int(rng.integers(0, 20)):][:22] = SEQ_N
            else:
                s[base + 300:base + 338] = SEQ_Y
        for chunk in (64, 21, 296, 150):
            a = T.sync_walk(s, chunk=chunk)
            b = T.sync_walk(s, chunk=chunk, per_call=True)
            assert a["slots"] == b["slots"] and a["events"] == b["events"]
        for chunk in (5, 20, 297, 400):
            check(s, chunk=chunk, with_cls=False)


# ---- the walk in the form the device runs it (csrc/tg_walk_core.h, k_walk's phases on the host) ----
def _dev_form(s, chunk):
    import emul
    ref0 = T.sync_walk(s, chunk=chunk, burst_events=False)
    ev0 = [e for e in ref0["events"] if e[0] == 1]
    if not ev0:
        return None
    anchor = ev0[0][1] + ev0[0][2] + 296
    if anchor + 510 > len(s):
        return None
    cls, ys = emul.cls_ysum(s, anchor, chunk)
    ref = T.sync_walk(s, chunk=chunk, anchor=anchor, cls=cls, ysum=ys, burst_events=False, grid=True, plain=True)
    got, st, why = T.sync_walk_emul(s, chunk, anchor, cls, ys)
    if st:
        return "fallback", why
    assert ref["noffgrid"] == 0
    assert got["events"] == ref["events"]
    for k in ("nslots", "ngrid", "final_state", "burst_seq", "tail_tn_adds"):
        assert got[k] == ref[k], k
    assert (np.asarray(got["grid_bits"]) == np.asarray(ref["grid_bits"])).all()
    # ... and the host walk itself equals the oracle's receiver on the bytes (events, number of bursts handed over)
    recs, oev = O.run_rx(s, chunk=chunk)
    assert [e for e in oev if e[0] != 2] == ref["events"]
    return "ok", len(ref["events"])


def test_cls_statement_in_c_equals_numpy_statement():
    """tests/host_emul/cls_emul.c (what the large CPU cases use) == emul_cls / emul_ysum above, early sequences and
    the TG_CLS_NOVIEW flag included; and both give tetra_find_train_seq()'s answer slot by slot"""
    import emul
    rng = np.random.default_rng(3)
    for t in range(12):
        s, _ = synth.frame_stream(seed=t + 1, nframes=int(rng.integers(2, 8)), lead_in=int(rng.integers(0, 300)), pad=int(rng.integers(600, 900)))
        s = s.copy()
        anchor = int(rng.integers(0, 700))
        n = (len(s) - anchor) // 510
        for _ in range(6):
            seq = (SEQ_Y, SEQ_N, SEQ_P)[int(rng.integers(0, 3))]
            o = anchor + 510 * int(rng.integers(0, n)) + int(rng.integers(0, 30))
            s[o:o + len(seq)] = seq
        for _ in range(3):
            s[int(rng.integers(0, len(s)))] ^= 1
        for chunk in (32, 64, 256):
            c, d = emul.cls_ysum(s, anchor, chunk)
            assert (emul_cls(s, anchor, chunk) == c).all() and (emul_ysum(s, anchor) == d).all()
        c, _ = emul.cls_ysum(s, anchor, 64)
        pad = np.concatenate([s, np.zeros(64, np.uint8)])
        for i in range(n):
            bs = anchor + 510 * i
            w = min(-(-(bs + 510) // 64) * 64, len(s)) - bs
            rc, off = T.find_train_seq(pad[bs:bs + w + 40], w, (1 << 0) | (1 << 1) | (1 << 3))
            cw = int(c[i])
            assert (cw & 0xFF, (cw >> 8) & 0xFFFF if cw & 0xFF != 0xFF else 0) == ((rc, off) if rc >= 0 else (0xFF, 0))


def test_device_form_of_the_walk_equals_the_host_walk():
    """tgpu_sync_walk_emul (one lane per non-plain grid slot through tgw_run, reachability by pointer doubling: what
    k_walk does) == the host walk == the oracle on damaged streams that stay on their grid: damaged training
    sequences (incl. two and three in a row, right after a SYNC burst, at the stream's end), spurious sequences
    anywhere, zeroed stretches; where it reports 'fallback' the reason is one of the documented ones"""
    rng = np.random.default_rng(2468)
    nok = nfb = 0
    for trial in range(60):
        stream, slots = synth.frame_stream(seed=100 + trial, nframes=int(rng.integers(3, 30)), lead_in=int(rng.integers(0, 400)),
                                           pad=int(rng.integers(600, 900)))
        s = stream.copy()
        tr = [i for i in range(0, len(s) - 60) if (s[i:i + 22] == SEQ_N).all() or (s[i:i + 22] == SEQ_P).all() or (s[i:i + 38] == SEQ_Y).all()]
        for i in tr:
            if rng.random() < 0.12:
                s[i + int(rng.integers(0, 22))] ^= 1
        if trial % 3 == 0 and len(tr) > 6:        # runs of damaged slots, the last slots of the stream
            j = int(rng.integers(0, len(tr) - 3))
            for i in tr[j:j + 3] + tr[-2:]:
                s[i + 3] ^= 1
        for _ in range(int(rng.integers(0, 6))):
            kind, p = int(rng.integers(0, 3)), int(rng.integers(0, len(s) - 60))
            if kind == 0:
                s[p:p + 38] = SEQ_Y
            elif kind == 1:
                s[p:p + 22] = SEQ_N
            else:
                s[p:p + int(rng.integers(1, 2500))] = 0
        r = _dev_form(np.ascontiguousarray(s), int(rng.choice([32, 64, 64, 128])))
        if r is None:
            continue
        if r[0] == "ok":
            nok += 1
        else:
            nfb += 1
            assert r[1] in (2, 3, 4), r       # TGW_WHY_WINDOW / _EARLY / _OFFGRID: the bytes have to decide
    assert nok >= 30, (nok, nfb)


def test_device_form_on_a_recording_of_bench_size():
    """one bench channel (125 000 slots, 1 % damaged training sequences, 64-byte feeds): no fallback, same outcome"""
    import bench
    st, _, _ = bench.make_mix_stream(T, 125000, 5, mnc=47, cc=6)
    r = _dev_form(np.ascontiguousarray(st), 64)
    assert r[0] == "ok" and r[1] > 1500


def test_device_form_on_a_recording_beyond_one_workgroups_arrays():
    """a channel of 266 000 slots (more than the 262 144 the LDS form holds): the same steps with the caps of k_walk_big's
    scratch area -- no hand-over to the host walk, same outcome as the host walk"""
    import bench
    st, _, _ = bench.make_mix_stream(T, 266000, 3, mnc=44, cc=9)
    r = _dev_form(np.ascontiguousarray(st), 64)
    assert r[0] == "ok" and r[1] > 3000


def test_device_form_with_feeds_of_128_and_256_bytes():
    """the search window of a slot reaches up to 765 bytes with such feeds: the classification word of a slot with nothing
    in its window carries the first sequence further on in the 832-byte view (TG_CLS_VIEWHIT), which is what the longer
    windows of these feeds find -- the device form settles damaged streams of these feeds without the bytes (before: every
    damaged slot was a hand-over), outcome == host walk == oracle"""
    rng = np.random.default_rng(1357)
    res = {128: [0, 0], 256: [0, 0]}
    for trial in range(40):
        stream, slots = synth.frame_stream(seed=300 + trial, nframes=int(rng.integers(3, 24)), lead_in=int(rng.integers(0, 400)),
                                           pad=int(rng.integers(600, 900)))
        s = stream.copy()
        tr = [i for i in range(0, len(s) - 60) if (s[i:i + 22] == SEQ_N).all() or (s[i:i + 22] == SEQ_P).all() or (s[i:i + 38] == SEQ_Y).all()]
        for i in tr:
            if rng.random() < 0.15:
                s[i + int(rng.integers(0, 22))] ^= 1
        if trial % 4 == 0 and len(tr) > 6:
            j = int(rng.integers(0, len(tr) - 3))
            for i in tr[j:j + 3] + tr[-2:]:
                s[i + 3] ^= 1
        for _ in range(int(rng.integers(0, 4))):        # spurious sequences anywhere: also in the stretch behind a damaged slot
            p = int(rng.integers(0, len(s) - 60))
            seq = (SEQ_N, SEQ_P, SEQ_Y)[int(rng.integers(0, 3))]
            s[p:p + len(seq)] = seq
        chunk = (128, 256)[trial & 1]
        r = _dev_form(np.ascontiguousarray(s), chunk)
        if r is None:
            continue
        res[chunk][0 if r[0] == "ok" else 1] += 1
        if r[0] != "ok":
            assert r[1] in (2, 3, 4), r
    assert res[128][0] >= 12 and res[256][0] >= 12, res


def test_device_form_finds_the_next_slots_sequence_in_a_late_window():
    """feeds of 256 bytes: the slot in front of a SYNC burst loses the lock, the SYNC burst gives it back, the slot behind it
    is handled one call late -- and has lost its own training sequence too: the reference's longer window (up to 1021 bytes)
    then finds the NEXT slot's sequence at offset 754 (a misplaced NORM sequence, lock kept).  The device form takes that from
    the word's TG_CLS_VIEWHIT field -- the view for such feeds is 1088 bytes, TG_VIEW_OF --: same events as the host walk and
    the oracle at every alignment of the slots against the feeds, no hand-over; a window that ends in front of offset 776
    finds nothing, which the word says as well"""
    found = 0
    for lead in range(0, 256, 6):
        stream, slots = synth.frame_stream(seed=78, nframes=5, lead_in=lead, pad=700)
        s = stream.copy()
        ys = [i for i in range(0, len(s) - 60) if (s[i:i + 38] == SEQ_Y).all()]
        sb = ys[2] - 214
        s[sb - 510 + 244 + 3] ^= 1
        s[sb + 510 + 244 + 3] ^= 1
        s = np.ascontiguousarray(s)
        r = _dev_form(s, 256)
        assert r is not None and r[0] == "ok", (lead, r)
        ev = T.sync_walk(s, chunk=256, burst_events=False)["events"]
        far = [e for e in ev if e[0] in (3, 4) and e[2] >= 510]        # misplaced-sequence events beyond the slot itself
        assert all(e[0] == 4 and e[2] == 754 for e in far), far
        found += bool(far)
    assert found >= 30, found
