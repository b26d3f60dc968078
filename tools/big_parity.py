"""one-off large differential run: n noisy NDB slots, every record field against the oracle"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T, oraclelib as O
import test_gpu_parity as G
eng = T.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
for ber, code in ((0.03, 0), (0.07, 0x41802A07), (0.12, 0xDEADBEEF | 3)):
    rng = np.random.default_rng(int(ber * 1000))
    ty = np.where(rng.random(n) < 0.5, 0, 1).astype(np.uint8)
    slots = T.synth_slots(ty, seed=int(ber * 1e4), scramb_init=code, ber=ber)
    t0 = time.time()
    rec, p, _ = G.run_plan(T, eng, slots, ty, codes=np.array([code], np.uint32))
    ok, _ = G.check_against_oracle(T, rec, ty, slots, code)
    print("ber %.2f code %08x: %d slots bit-exact (type-1 bits, crc16, crc_ok), %d CRC-ok blocks, %.1f s" % (ber, code, n, ok, time.time() - t0))
