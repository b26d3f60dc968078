"""experiment: k_front time vs slot alignment (510-byte pitch = 2-byte aligned slots, 512 = dword-aligned,
640 = line-aligned)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
rng = np.random.default_rng(1000)
types = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
slots = T.synth_slots(types, seed=1, scramb_init=0)
eng = T.Engine(0)
K = 60
st = torch.cuda.current_stream().cuda_stream
for pitch in (510, 512, 516, 640, 510):
    buf = np.zeros((n, pitch), np.uint8)
    buf[:, :510] = slots
    d_stream = torch.from_numpy(buf.reshape(-1)).cuda()
    d_rec = torch.empty(n * 320, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1); plan.load(np.arange(n, dtype=np.uint64) * pitch, types)
    prof = T.Prof(K)
    for k in range(K): plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), st, prof, k)
    torch.cuda.synchronize()
    ms = prof.read(K)
    print("pitch %d: k_front %.1f us" % (pitch, 1e3 * float(np.mean(ms[20:, 0]))))
    plan.close(); del d_stream, d_rec
