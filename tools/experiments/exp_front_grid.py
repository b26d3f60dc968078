"""experiment: k_front time vs number of workgroups (tgpu_engine_set_option: TGPU_OPT_FRONT_BLOCKS)"""
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
K = 30
prof = T.Prof(K)
st = torch.cuda.current_stream().cuda_stream
for blocks in (4096, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 4096):
    T.set_option(T.OPT_FRONT_BLOCKS, int(blocks))
    for k in range(K): plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), st, prof, k)
    torch.cuda.synchronize()
    ms = prof.read(K)
    print("blocks %5d  k_front %.1f us" % (blocks, 1e3 * float(np.mean(ms[5:, 0]))))
