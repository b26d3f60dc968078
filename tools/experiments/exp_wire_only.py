"""config 2 step time with full records / full + wire records / wire records only (tgpu_plan_set_wire_only)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
rng = np.random.default_rng(1)
types = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
slots = T.synth_slots(types, seed=1, scramb_init=0)
eng = T.Engine(0)
d = torch.from_numpy(slots.reshape(-1)).cuda()
d_rec = torch.empty(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
d_wire = torch.empty(n * T.WIRE_BYTES, dtype=torch.uint8, device="cuda")
hs = torch.cuda.current_stream().cuda_stream
for name, wire, only in (("full records", False, False), ("full + wire", True, False), ("wire only", True, True)):
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types)
    plan.set_wire(d_wire.data_ptr() if wire else 0)
    plan.set_wire_only(only)
    for _ in range(10):
        plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 60
    for _ in range(K):
        plan.execute(d.data_ptr(), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    print("%-14s %.3f ms per 1 M bursts  (%.2fe9 bursts/s)" % (name, el * 1e3, n / el / 1e9))
    plan.close()
