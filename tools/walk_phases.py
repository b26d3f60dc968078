"""phases of k_walk (build with TGPU_HIPCC_FLAGS=-DTGW_TIMING): shader-clock cycles between the stamps of workgroup c"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import osmo_tetra_amd as T
import bench
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = 1000000 // Cn
streams = [bench.make_mix_stream(T, per, c, mnc=42 + c, cc=1 + c)[0] for c in range(Cn)]
offs, o = [], 0
for st in streams:
    offs.append(o); o += (len(st) + T.STREAM_SLACK + 15) & ~15
buf = np.zeros(o + 4096, np.uint8)
for st, f in zip(streams, offs):
    buf[f:f + len(st)] = st
eng = T.Engine(0)
d_base = torch.from_numpy(buf).cuda()
cap = sum(len(st) // 510 + 32 for st in streams)
chans = T.multi_chan_table(streams, offs)
plan = T.Plan(eng, cap, Cn)
rec = torch.empty(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda")
hs = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    ms = T.MultiSyncDev(eng, plan, None, d_base.data_ptr(), None, rec.data_ptr(), 64, hs, chans=chans)
    outs = ms.collect(raw=True)
torch.cuda.synchronize()
st = np.zeros((64, 12), np.uint64)
T.lib().tgk_walk_stamps(st.ctypes.data_as(C.c_void_p))
names = ["A+B bitmap, nodes", "C nodes through tgw_run", "D reachability", "E spans", "F bitmap out", "G events"]
for c in range(Cn):
    d = np.diff(st[c, :7].astype(np.int64))
    print("channel", c, " ".join("%s %d" % (n, x) for n, x in zip(names, d)), "total", int(st[c, 6] - st[c, 0]), "cycles (100 MHz clock?)")
