"""host-walk cost split: steady-state per slot vs per re-lock (grid mode, no per-burst events)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
types = np.tile(pat, n // 8 + 1)[:n]
eng = T.Engine(0)
for frac in (0.0, 0.01, 0.05):
    rng = np.random.default_rng(7)
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=11, scramb_init=0x41802A07)
    bad = np.flatnonzero(rng.random(n) < frac) + 1
    for i in bad:
        off = 214 if slots[i, 214:252].tolist() == slots[0, 214:252].tolist() else 244
        slots[i, off + 5] ^= 1
    stream = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
    d = torch.from_numpy(np.concatenate([stream, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    r0 = T.sync_stream(eng, stream, d.data_ptr(), burst_events=False)
    anchor = r0["anchor"]
    ng = (len(stream) - anchor) // 510
    cls, ys = T.sync_classify(eng, d.data_ptr(), len(stream), 64, anchor, ng, with_ysum=True)
    for _ in range(3):
        t0 = time.perf_counter()
        r = T.sync_walk(stream, chunk=64, anchor=anchor, cls=cls, ysum=ys, burst_events=False, grid=True)
        el = time.perf_counter() - t0
    nre = sum(1 for e in r["event_arr"]["ev"] if e == 1)
    print("corrupted %.0f%%: walk %.3f ms, %d slots delivered, %d re-locks" % (100 * frac, el * 1e3, r["nslots"], nre))
    good = np.isin(cls & 0x01FFFFFF, [3 | 214 << 8, 0 | 244 << 8, 1 | 244 << 8])
    runs = np.diff(np.flatnonzero(~good))
    print("   classification words that are not plain deliveries: %d of %d; flagged EARLY21 %d, CLIPPED %d, NONBINARY %d"
          % ((~good).sum(), len(cls), ((cls >> 24) & 1).sum(), ((cls >> 26) & 1).sum(), ((cls >> 25) & 1).sum()))
