"""experiment: do consecutive batches overlap (front end of batch n+1 under the trellis kernels of batch n)
when they are issued on two streams with two plans?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
rng = np.random.default_rng(1000)
types = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
slots = T.synth_slots(types, seed=1, scramb_init=0)
eng = T.Engine(0)
d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
plans, recs, streams = [], [], []
for i in range(2):
    p = T.Plan(eng, n, 1); p.load(np.arange(n, dtype=np.uint64) * 510, types)
    plans.append(p); recs.append(torch.empty(n * 320, dtype=torch.uint8, device="cuda")); streams.append(torch.cuda.Stream())
for mode in ("single plan", "one stream", "two streams", "single plan", "one stream", "two streams", "single plan, rec alternates", "single rec, plan alternates"):
    for _ in range(3):
        for i in range(2): plans[i].execute(d_stream.data_ptr(), recs[i].data_ptr(), streams[i].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for k in range(K):
        i = 0 if mode == "single plan" else (k & 1)
        s = streams[i] if mode == "two streams" else streams[0]
        pi = 0 if mode.startswith("single plan") else i
        ri = 0 if mode.startswith("single rec") else i
        plans[pi].execute(d_stream.data_ptr(), recs[ri].data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(mode, "ms/step %.4f" % (el / K * 1e3), "G bursts/s %.3f" % (n * K / el / 1e9))
