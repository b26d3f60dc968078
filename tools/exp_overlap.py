import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
rng = np.random.default_rng(1000)
types = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
slots = T.synth_slots(types, seed=1, scramb_init=0)
eng = T.Engine(0)
d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
d_rec = torch.empty(n * 320, dtype=torch.uint8, device="cuda")
plan = T.Plan(eng, n, 1); plan.load(np.arange(n, dtype=np.uint64) * 510, types)
prof = T.Prof(20)
st = torch.cuda.current_stream().cuda_stream
for mode in ("overlap", "serial", "overlap", "serial"):
    for _ in range(3): plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(20):
        if mode == "overlap": plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), st)
        else: plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), st, prof, k)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(mode, "ms/step", el / 20 * 1e3, "G bursts/s", n * 20 / el / 1e9)
