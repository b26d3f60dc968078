import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import osmo_tetra_amd as T
import bench
C = 8; per = 125000
streams = []; codes = []
for c in range(C):
    st, code, _ = bench.make_mix_stream(T, per, 0, mnc=42 + c)
    streams.append(st); codes.append(code)
offs = []; o = 0
for st in streams:
    offs.append(o); o += (len(st) + T.STREAM_SLACK + 15) & ~15
buf = np.zeros(o + 4096, np.uint8)
for st, f in zip(streams, offs):
    buf[f:f + len(st)] = st
eng = T.Engine(0)
d_base = torch.from_numpy(buf).cuda()
cap = sum(len(st) // 510 + 32 for st in streams)
plan = T.Plan(eng, cap, C)
rec = torch.empty(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda")
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
acc = {}
def add(k, t): acc[k] = acc.get(k, 0.0) + t
N = 60
for it in range(N + 5):
    if it == 5: acc.clear()
    c0 = time.process_time()
    a = time.perf_counter(); ms = T.MultiSync(eng, plan, streams, d_base.data_ptr(), offs, 64, s1.cuda_stream); b = time.perf_counter(); add("begin", b - a)
    s1.synchronize(); a = time.perf_counter(); add("wait_gpu_front", a - b)
    outs = ms.finish(burst_events=False, nthreads=1); b = time.perf_counter(); add("finish", b - a)
    plan.set_wire(0); plan.execute(d_base.data_ptr(), rec.data_ptr(), s2.cuda_stream); a = time.perf_counter(); add("execute_launch", a - b)
    s2.synchronize(); b = time.perf_counter(); add("wait_gpu_decode", b - a)
    add("cpu", time.process_time() - c0)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()})
