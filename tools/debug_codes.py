import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T, oraclelib as O
from test_gpu_parity import run_plan
eng = T.Engine(0)
cells = [(262, 42, 1), (901, 77, 9)]
rng = np.random.default_rng(5)
parts, types = [], []
for (mcc, mnc, cc) in cells:
    code = O.scramb_get_init(mcc, mnc, cc)
    ty = np.array([3, 0, 1, 1, 0] * 6, np.uint8)
    parts.append(T.synth_slots(ty, seed=int(rng.integers(1 << 30)), scramb_init=code, mcc=mcc, mnc=mnc, cc=cc, ber=0.01))
    types.append(ty)
slots, types = np.concatenate(parts), np.concatenate(types)
pre_t = np.array([1, 0], np.uint8)
pre = T.synth_slots(pre_t, seed=99, scramb_init=0)
slots, types = np.concatenate([pre, slots]), np.concatenate([pre_t, types])
rec, p, codes_out = run_plan(T, eng, slots, types)
print("types", types.tolist())
print("code ", [hex(x) for x in p["code"]])
print("crcok", p["crc_ok"].tolist())
print("sbcode", [hex(x) for x in p["sbcode"][types == 3]])
print("expected", hex(O.scramb_get_init(*cells[0])), hex(O.scramb_get_init(*cells[1])), "final", [hex(x) for x in codes_out])
for i in np.where(types == 3)[0]:
    r = O.decode_block(O.T_SB1, slots[i][94:214], 3)
    print(i, "oracle sb1 ok", r[2], hex(r[1]), "gpu crc", hex(p["crc"][i, 0]))
