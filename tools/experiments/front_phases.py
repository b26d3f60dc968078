"""where a wave of k_front_stream spends its time (library built with TGPU_HIPCC_FLAGS=-DTGS_TIMING): reference-clock
ticks (100 MHz) between the marks of a group, summed over all waves of one launch on the bench's batch of 8 channels,
printed per group and as shares"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import osmo_tetra_amd as T
import bench
Cn, per = 8, 125000
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
acc = np.zeros(8, np.uint64)
L = T.lib()
L.tgk_front_stream_stamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(4):
    L.tgk_front_stream_stamps(acc.ctypes.data_as(C.c_void_p), 1)
    ms = T.MultiSyncDev(eng, plan, None, d_base.data_ptr(), None, rec.data_ptr(), 64, hs, chans=chans)
    outs = ms.collect(raw=True)
    torch.cuda.synchronize()
L.tgk_front_stream_stamps(acc.ctypes.data_as(C.c_void_p), 0)
ngroups = (int(ms.ngrid) + 3) // 4
names = ["fetch of the next group issued", "wait for this group's bytes", "bits -> LDS -> column", "copies + search + ballots",
         "classification (bpermutes)", "four gathers", "staged stores"]
tot = float(acc[:7].sum())
print("groups", ngroups, "ticks per group (10 ns each), share")
for n, x in zip(names, acc[:7]):
    print("  %-34s %7.1f  %5.1f %%" % (n, float(x) / ngroups, 100.0 * float(x) / tot))
print("  %-34s %7.1f" % ("sum", tot / ngroups))
