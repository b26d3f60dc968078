"""offline fuzz of the device form of the synchroniser walk (csrc/tg_walk_core.h) on the CPU: random damaged streams
(bit flips, inserted / deleted bytes, spurious training sequences, zeroed stretches, non-binary bytes) through
tgpu_sync_walk_emul -- k_walk's phases run on the host -- against the host walk (tgpu_sync_walk_plain, grid mode) fed
the same numpy statement of the classification words; where the device form reports "fallback" only that is counted.
usage: fuzz_walk_dev.py [N [seed]]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import osmo_tetra_amd as T
import synth
from test_stream_sync_cpu import SEQ_Y, SEQ_N
import emul

KEYS = ("nslots", "ngrid", "final_state", "burst_seq", "tail_tn_adds")


def damaged(rng, big=False):
    stream, _ = synth.frame_stream(seed=int(rng.integers(1, 1 << 30)), nframes=int(rng.integers(2, 120 if big else 40)),
                                   lead_in=int(rng.integers(0, 600)), pad=int(rng.integers(600, 900)),
                                   ber=float(rng.choice([0.0, 0.0, 0.02])))
    s = stream.copy()
    keep = bool(rng.integers(0, 2))      # half of the streams: damage that leaves the slot grid where it is
    if keep:
        nsl = (len(s) - 700) // 510
        for _ in range(int(rng.integers(0, max(2, nsl // 6)))):     # damaged training sequences, as the bench streams have them
            i = int(rng.integers(0, max(1, nsl)))
            p = int(rng.integers(0, 510))
            q = (len(stream) - len(s)) + i * 510 + p
            if 0 <= q < len(s):
                s[q] ^= 1
        # (training sequences sit at 214 / 244 of a slot wherever the lead-in put the grid: hit them on purpose too)
        starts = [i for i in range(0, len(s) - 60) if s[i:i + 22].tolist() == SEQ_N.tolist() or s[i:i + 38].tolist() == SEQ_Y.tolist()]
        for i in starts:
            if rng.random() < 0.15:
                s[i + int(rng.integers(0, 22))] ^= 1
    for _ in range(int(rng.integers(0, 30 if big else 10))):
        kind = int(rng.choice([0, 3, 4, 5])) if keep else int(rng.integers(0, 7))
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
        elif kind == 5:
            s[p:p + int(rng.integers(1, 1500))] = 0
        else:
            s[p] = int(rng.integers(2, 256))
    return np.ascontiguousarray(s)


def one(s, chunk):
    """returns None (never locks), 'fallback:<why>' or 'ok' / raises on a mismatch"""
    ref0 = T.sync_walk(s, chunk=chunk, burst_events=False)
    if not len(ref0["slots"]) and not any(e[0] == 1 for e in ref0["events"]):
        return None
    ev0 = [e for e in ref0["events"] if e[0] == 1]
    anchor = ev0[0][1] + ev0[0][2] + 296
    if anchor + 510 > len(s):
        return None
    cls, ys = emul.cls_ysum(s, anchor, chunk)
    if not len(cls):
        return None
    ref = T.sync_walk(s, chunk=chunk, anchor=anchor, cls=cls, ysum=ys, burst_events=False, grid=True, plain=True)
    got, st, why = T.sync_walk_emul(s, chunk, anchor, cls, ys)
    if st:
        return "fallback:%d" % why
    if ref["noffgrid"]:
        raise AssertionError("host walk left the grid, device form did not notice")
    assert got["events"] == ref["events"], (got["events"][:12], ref["events"][:12])
    for k in KEYS:
        assert got[k] == ref[k], (k, got[k], ref[k])
    assert (np.asarray(got["grid_bits"]) == np.asarray(ref["grid_bits"])).all()
    return "ok"


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 24680)
    T.build_library()
    tally = {}
    by_chunk = {}
    # the bench's kind of stream first: long, 1 % damaged training sequences, nothing else
    import bench
    for g in range(4):
        st, _, _ = bench.make_mix_stream(T, 20000, g, mnc=42 + g, cc=1 + g)
        r = one(np.ascontiguousarray(st), 64)
        print("bench-like stream", g, "->", r)
        assert r == "ok"
    for trial in range(N):
        s = damaged(rng, big=(trial % 4 == 0))
        chunk = int(rng.choice([32, 64, 64, 64, 128, 256]))
        try:
            r = one(s, chunk)
        except AssertionError as ex:
            np.save("/tmp/fz/bad_%d.npy" % trial, s)
            print("MISMATCH trial", trial, "chunk", chunk, "len", len(s), str(ex)[:400])
            r = "MISMATCH"
        tally[r] = tally.get(r, 0) + 1
        by_chunk[(chunk, r)] = by_chunk.get((chunk, r), 0) + 1
    print(tally)
    print(sorted(by_chunk.items()))
    sys.exit(1 if tally.get("MISMATCH") else 0)
