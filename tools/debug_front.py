import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T, emul
eng = T.Engine(0)
types = np.array([0, 1, 3, 0, 1, 3, 1, 0] * 40, np.uint8)
slots = T.synth_slots(types, seed=7, scramb_init=0x41802a07)
n = len(types)
d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
d_rec = torch.zeros(n * 320, dtype=torch.uint8, device="cuda")
plan = T.Plan(eng, n, 1)
plan.load(np.arange(n, dtype=np.uint64) * 510, types)
plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), 0)
got = plan.read_packed()
bad = 0
for i in range(n):
    want = emul.pack_slot(int(types[i]), slots[i])
    want[19] = got[i, 19]
    if not (got[i] == want).all():
        bad += 1
        if bad <= 4:
            print("slot", i, "type", types[i]); print(" got ", [hex(x) for x in got[i]]); print(" want", [hex(x) for x in want])
print("bad", bad, "of", n)
