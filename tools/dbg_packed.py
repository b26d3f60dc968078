import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import osmo_tetra_amd as T
n = 64
rng = np.random.default_rng(7)
pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
types = np.tile(pat, n // 8 + 1)[:n]
slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=11, scramb_init=0x41802A07)
stream = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
eng = T.Engine(0)
d_stream = torch.from_numpy(np.concatenate([stream, np.zeros(T.STREAM_SLACK + 4096, np.uint8)])).cuda()
plan = T.Plan(eng, n + 8, 1)
hs = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("1", "0"):
    T.set_option(T.OPT_STREAM_EXACT, int(mode))
    g = T.GridSync(eng, plan, stream, d_stream.data_ptr(), 64, hs)
    out = g.finish(burst_events=False)
    res[mode] = plan.read_packed().reshape(-1, 20).copy()
a, b = res["1"], res["0"]
bad = np.argwhere(a != b)
print("differing words:", len(bad), "of", a.size)
for s in range(6):
    print("slot", s, "type", a[s, 19] & 0xff)
    for w in range(20):
        if a[s, w] != b[s, w]:
            print("   w%02d want %08x got %08x xor %08x" % (w, a[s, w], b[s, w], a[s, w] ^ b[s, w]))
