"""where the K-steps-between-two-synchronisations form of the bench spends its time: per step the host time of the launch
call and of the collect, with D steps in flight (one recorded channel set, rotating captures as bench.py)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
import bench
Cn, per, D, K, NB = 8, 125000, 8, 20, 8
def capture(b):
    sts = [bench.make_mix_stream(T, per, c + 1000 * b, mnc=42 + c, cc=1 + c % 60)[0] for c in range(Cn)]
    offs, o = [], 0
    for st in sts:
        offs.append(o); o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(sts, offs):
        buf[f:f + len(st)] = st
    return sts, offs, buf
streams, offs, buf = capture(0)
eng = T.Engine(0)
bases = [torch.from_numpy(buf).cuda()] + [torch.from_numpy(capture(b)[2]).cuda() for b in range(1, NB)]
cap = sum(len(st) // 510 + 32 for st in streams)
chans = T.multi_chan_table(streams, offs)
plans = [T.Plan(eng, cap, Cn) for _ in range(D)]
for p in plans:
    p.set_side_stream(False)
recs = [torch.empty(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda") for _ in range(D)]
strm = [torch.cuda.Stream() for _ in range(D)]
import collections
def run(total):
    fl = collections.deque(); tl, tc = [], []
    for k in range(total):
        j = k % D
        if len(fl) == D:
            t0 = time.perf_counter(); old = fl.popleft(); old.collect_begin(); old.collect_end(raw=True); tc.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        fl.append(T.MultiSyncDev(eng, plans[j], None, bases[k % NB].data_ptr(), None, recs[j].data_ptr(), 64, strm[j].cuda_stream, chans=chans))
        tl.append(time.perf_counter() - t0)
    while fl:
        t0 = time.perf_counter(); old = fl.popleft(); old.collect_begin(); old.collect_end(raw=True); tc.append(time.perf_counter() - t0)
    return tl, tc
run(3 * D); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); tl, tc = run(K); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("K=%d: %.3f ms per step; launch calls ms %s ; collects ms %s" % (K, el / K * 1e3, [round(x * 1e3, 2) for x in tl], [round(x * 1e3, 2) for x in tc]))
# one batch at a time: launch, collect, next (what a caller who waits per batch sees)
D1 = []
for k in range(24):
    t0 = time.perf_counter()
    ms = T.MultiSyncDev(eng, plans[0], None, bases[k % NB].data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
    ms.collect_begin(); ms.collect_end(raw=True)
    D1.append((time.perf_counter() - t0) * 1e3)
print("one batch at a time, ms per batch:", [round(x, 3) for x in D1[4:]], "median %.3f" % sorted(D1[4:])[len(D1[4:]) // 2])
for p in plans:
    p.set_side_stream(True)
D1 = []
for k in range(24):
    t0 = time.perf_counter()
    ms = T.MultiSyncDev(eng, plans[0], None, bases[k % NB].data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
    ms.collect_begin(); ms.collect_end(raw=True)
    D1.append((time.perf_counter() - t0) * 1e3)
print("one batch at a time, the plan's side stream in play, ms per batch:", [round(x, 3) for x in D1[4:]], "median %.3f" % sorted(D1[4:])[len(D1[4:]) // 2])
