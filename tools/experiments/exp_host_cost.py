"""host cost of a device-walk step: wall time of tgpu_sync_multi_launch and _collect per step, steady state"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
import bench
C, per = 8, 125000
streams = [bench.make_mix_stream(T, per, c, mnc=42 + c, cc=1 + c)[0] for c in range(C)]
offs, o = [], 0
for st in streams:
    offs.append(o); o += (len(st) + T.STREAM_SLACK + 15) & ~15
buf = np.zeros(o + 4096, np.uint8)
for st, f in zip(streams, offs):
    buf[f:f + len(st)] = st
eng = T.Engine(0)
d_base = torch.from_numpy(buf).cuda()
cap = sum(len(st) // 510 + 32 for st in streams)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 3
chans = T.multi_chan_table(streams, offs)
plans = [T.Plan(eng, cap, C) for _ in range(D)]
recs = [torch.empty(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda") for _ in range(D)]
strm = [torch.cuda.Stream() for _ in range(D)]
import collections
fl = collections.deque()
tl, tc = [], []
for rep in range(3):
    torch.cuda.synchronize()
    t00 = time.perf_counter(); c00 = time.process_time(); th0 = time.thread_time()
    for k in range(60):
        j = k % D
        if len(fl) == D:
            a = time.perf_counter(); fl.popleft().collect(raw=True); tc.append(time.perf_counter() - a)
        a = time.perf_counter()
        fl.append(T.MultiSyncDev(eng, plans[j], None, d_base.data_ptr(), None, recs[j].data_ptr(), 64, strm[j].cuda_stream, chans=chans))
        tl.append(time.perf_counter() - a)
    while fl:
        fl.popleft().collect(raw=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t00
    print("rep %d: %.3f ms/step wall, process cpu %.3f ms/step, this thread %.3f ms/step; launch call %.3f ms (median), collect call %.3f ms (median)"
          % (rep, el / 60 * 1e3, (time.process_time() - c00) / 60 * 1e3, (time.thread_time() - th0) / 60 * 1e3,
             np.median(tl[-60:]) * 1e3, np.median(tc[-50:]) * 1e3))
    print("   first launches (ms):", [round(x * 1e3, 3) for x in tl[-60:][:6]], "first collects:", [round(x * 1e3, 3) for x in tc[-57:][:6]])
