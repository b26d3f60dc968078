"""experiment helper: k_front_stream (packed bits) vs k_front_stream_v1 (per position) on a 1 M-slot config-3 stream:
classification outputs equal, microseconds per launch (HIP events around tgk_front_stream's launches)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(7)
pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
types = np.tile(pat, n // 8 + 1)[:n]
slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=11, scramb_init=0x41802A07)
bad = np.flatnonzero(rng.random(n) < 0.01) + 1
for i in bad:
    off = 214 if slots[i, 214:252].tolist() == slots[0, 214:252].tolist() else 244
    slots[i, off + 5] ^= 1
stream = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
eng = T.Engine(0)
d_stream = torch.from_numpy(np.concatenate([stream, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
plan = T.Plan(eng, n + 8, 1)
hs = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("1", "0"):
    T.set_option(T.OPT_STREAM_EXACT, int(mode))
    ts = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g = T.GridSync(eng, plan, stream, d_stream.data_ptr(), 64, hs)      # classification launch + D2H of cls / ysum
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
        out = g.finish(burst_events=False)
    res[mode] = (out, plan.read_packed())
    print("TGPU_OPT_STREAM_EXACT=%s: front + D2H of 6 B/slot: min %.1f us  median %.1f us  (delivered %d of %d grid slots)"
          % (mode, min(ts), float(np.median(ts)), out["nslots"], out["ngrid"]))
a, b = res["1"], res["0"]
print("packed equal:", bool((a[1] == b[1]).all()), " outcome equal:", a[0]["nslots"] == b[0]["nslots"] and a[0]["events"] == b[0]["events"])
