"""compare the host-walk cost of tgpu_sync_stream (slot table) and tgpu_sync_stream_grid (bitmap) on one box"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
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
for rep in range(3):
    r = T.sync_stream(eng, stream, d_stream.data_ptr(), 64, hs, burst_events=False)
    g = T.sync_stream_grid(eng, plan, stream, d_stream.data_ptr(), 64, hs, burst_events=False)
# serialised stage times of the decode that follows (HIP events between the launches)
d_rec = torch.empty((n + 8) * T.REC_BYTES, dtype=torch.uint8, device="cuda")
prof = T.Prof(10)
for k in range(10):
    plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), hs, prof, k)
torch.cuda.synchronize()
ms = prof.read(10)[3:].mean(axis=0)
print("decode stages (us):", {nm: round(float(v) * 1e3, 1) for nm, v in zip(T.Prof.stage_names(), ms)})
