"""clean-block fast path (tgpu_plan_set_fastpath): step time of config 2 with the flag off / on at several channel BERs"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
n = 1_000_000
rng = np.random.default_rng(1000)
types = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
eng = T.Engine(0)
st = torch.cuda.current_stream().cuda_stream
d_rec = torch.empty(n * 320, dtype=torch.uint8, device="cuda")
for ber in (0.0, 1e-5, 1e-4, 1e-3, 1e-2):
    slots = T.synth_slots(types, seed=1, scramb_init=0, ber=ber)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    out = []
    for fast in (False, True):
        plan = T.Plan(eng, n, 1); plan.set_fastpath(fast); plan.load(np.arange(n, dtype=np.uint64) * 510, types)
        for _ in range(30): plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), st)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 100
        ok = int(T.parse_records(d_rec.view(n, 320)[:20000].cpu().numpy())["crc_ok"][:, 0].sum())
        out.append((el * 1e3, n / el / 1e9, ok))
        plan.close()
    print("BER %-7g  off: %.3f ms (%.2f G bursts/s)   on: %.3f ms (%.2f G bursts/s)   crc-ok of first 20000: %d / %d" %
          (ber, out[0][0], out[0][1], out[1][0], out[1][1], out[0][2], out[1][2]))
    del d_stream
