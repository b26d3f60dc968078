"""one-off large differential run of the soft path (tgpu_plan_execute_float: fused slicer + soft gather + packed 16-bit
soft trellis) against the oracle's soft chain: 3 x 200 k slots at three noise levels, three cells"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T
import oraclelib as O
n = 200_000
eng = T.Engine(0)
pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
types = np.tile(pat, n // 8 + 1)[:n]
for sigma, cell, seed in ((0.5, (262, 42, 1), 1), (0.9, (901, 77, 9), 2), (2.0, (1, 2, 3), 3)):
    t0 = time.perf_counter()
    rng = np.random.default_rng(seed)
    code = O.scramb_get_init(*cell)          # (the SYNC PDUs name the cell: the code learnt from SB1 is the one the blocks carry)
    slots = T.synth_slots(types, seed=40 + seed, scramb_init=code, mcc=cell[0], mnc=cell[1], cc=cell[2])
    bits = slots.reshape(-1)
    phi = (O.bits_to_phase(bits) + rng.normal(0, sigma, len(bits) // 2)).astype(np.float32)
    d_phi = torch.from_numpy(phi).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * 510, types, None, np.array([code], np.uint32))
    plan.execute_float(d_phi.data_ptr(), len(phi), d_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = T.parse_records(d_rec.cpu().numpy().reshape(n, T.REC_BYTES))
    soft = O.float_to_soft(phi)
    ok, want, wcrc = O.bench_decode_slots_soft(soft.reshape(n, 510), types, code)
    n1, n2, sb = types == 0, types == 1, types == 3
    good = (p["bbk"] == want[:, :14]).all() and (p["bits1"][n1] == want[n1, 14:282]).all() and \
        (p["bits1"][n2][:, :124] == want[n2, 14:138]).all() and (p["bits2"][n2] == want[n2, 138:262]).all() and \
        (p["bits1"][sb][:, :60] == want[sb, 14:74]).all() and (p["bits2"][sb] == want[sb, 138:262]).all() and \
        (p["crc"][:, 0] == wcrc[:, 0]).all() and (p["crc"][n2 | sb, 1] == wcrc[n2 | sb, 1]).all()
    print("sigma %.1f code %08x: %d slots %s (type-1 bits, BBK, crc16), %d CRC-ok blocks, %.1f s"
          % (sigma, code, n, "bit-exact" if good else "DIFFER", ok, time.perf_counter() - t0))
    plan.close()
    assert good
