"""offline fuzz of the stream synchroniser's bitmap steady state: N random damaged streams (bit flips, inserted /
deleted bytes, spurious training sequences, zeroed stretches, non-binary bytes) through tgpu_sync_stream_grid with and
without per-burst events, and the oracle receiver on each"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T
import oraclelib as O
import synth
from test_stream_sync_cpu import SEQ_Y, SEQ_N
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = T.Engine(0)
hs = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(987654)
nloss = nbad = 0
for trial in range(N):
    stream, _ = synth.frame_stream(seed=int(rng.integers(1, 1 << 30)), nframes=int(rng.integers(2, 40)),
                                   lead_in=int(rng.integers(0, 600)), pad=int(rng.integers(600, 900)),
                                   ber=float(rng.choice([0.0, 0.0, 0.02])))
    s = stream.copy()
    for _ in range(int(rng.integers(0, 10))):
        kind = int(rng.integers(0, 7))
        p = int(rng.integers(0, len(s) - 60))
        if kind == 0:
            s[p] ^= 1
        elif kind == 1:
            s = np.concatenate([s[:p], rng.integers(0, 2, int(rng.integers(1, 40))).astype(np.uint8), s[p:]])
        elif kind == 2:
            s = np.concatenate([s[:p], s[p + int(rng.integers(1, 40)):]])
        elif kind == 3:
            s[p:p + 38] = SEQ_Y
        elif kind == 4:
            s[p:p + 22] = SEQ_N
        elif kind == 5:
            s[p:p + int(rng.integers(1, 1500))] = 0
        else:
            s[p] = int(rng.integers(2, 256))
    s = np.ascontiguousarray(s)
    chunk = int(rng.choice([32, 64, 64, 64, 128, 256]))
    _, wev = O.run_rx(s, chunk=chunk)
    d = torch.from_numpy(np.concatenate([s, np.zeros(T.STREAM_SLACK, np.uint8)])).cuda()
    pa, pb = T.Plan(eng, len(s) // 510 + 8, 1), T.Plan(eng, len(s) // 510 + 8, 1)
    a = T.sync_stream_grid(eng, pa, s, d.data_ptr(), chunk, hs, burst_events=False)
    b = T.sync_stream_grid(eng, pb, s, d.data_ptr(), chunk, hs, burst_events=True)
    quiet = [e for e in wev if e[0] != 2]
    ok = b["events"] == wev and a["events"] == quiet and all(a[k] == b[k] for k in ("nslots", "ngrid", "noffgrid", "anchor", "final_state", "burst_seq", "tail_tn_adds"))
    if ok and a["ngrid"] and not a["noffgrid"]:
        ok = bool((np.asarray(a["grid_bits"]) == np.asarray(b["grid_bits"])).all())
    if not ok:
        nbad += 1
        print("MISMATCH in trial", trial, "chunk", chunk, "len", len(s))
    nloss += sum(1 for e in quiet if e[0] in (3, 5))
    pa.close(); pb.close()
print("%d streams, %d lock losses, %d mismatches" % (N, nloss, nbad))
sys.exit(1 if nbad else 0)
