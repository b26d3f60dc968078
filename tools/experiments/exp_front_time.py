"""experiment helper: k_front / trellis stage times of config 2 without any correctness check (for ablation
builds whose output is deliberately wrong)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
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
K = 80
prof = T.Prof(K)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for k in range(K): plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), st, prof, k)
    torch.cuda.synchronize()
    ms = prof.read(K)
    print("stage us:", " ".join("%s %.1f" % (nm, 1e3 * float(np.mean(ms[30:, i]))) for i, nm in enumerate(T.Prof.stage_names())))
