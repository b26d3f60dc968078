"""end-to-end rate of the drop-in channel API: host bytes -> tetra_burst_sync_in() (64-byte reads like tetra-rx.c)
-> queued bursts -> GPU batches -> callbacks in the reference's order"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import osmo_tetra_amd as T
n = 200_000
pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
types = np.tile(pat, n // 8 + 1)[:n]
slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=11, scramb_init=0x41802A07)
rng = np.random.default_rng(1)
stream = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
eng = T.Engine(0)
for batch in (1, 64, 1024, 16384):
    cnt = [0]
    def cb(ud, offset):
        cnt[0] += 1
        return -1
    ch = T.Channel(eng, batch_slots=batch, on_unitdata=None)
    t0 = time.perf_counter()
    ch.feed(stream, chunk=64) if "chunk" in T.Channel.feed.__code__.co_varnames else ch.feed(stream)
    ch.flush()
    el = time.perf_counter() - t0
    print("batch %5d: %.2f s for %d bursts -> %.0f bursts/s end to end (%d blocks delivered)" % (batch, el, n, n / el, len(ch.records)))
    ch.close()
