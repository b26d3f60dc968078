"""round 6: which phase the waves of a SIMD are in, and when (a -DTG_SLOT_TIMING build of the library): per task of k_slot its HW_ID and
the 100 MHz clock at start / front phase done / trellis done / end.  Prints the phases' durations and, per SIMD, how much of the kernel's
time 0 / 1 / 2 / 3 of its waves spent in the front phase.  usage: python tools/experiments/slot_phases.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import osmo_tetra_amd as T  # noqa: E402

Cn, per = 8, 125000
streams, offs, o = [], [], 0
for c in range(Cn):
    st, _, _ = bench.make_mix_stream(T, per, c, mnc=42 + c, cc=1 + c)
    streams.append(st)
    offs.append(o)
    o += (len(st) + T.STREAM_SLACK + 15) & ~15
buf = np.zeros(o + 4096, np.uint8)
for st, f in zip(streams, offs):
    buf[f:f + len(st)] = st
eng = T.Engine(0)
d = torch.from_numpy(buf).cuda()
cap = sum(len(st) // 510 + 32 for st in streams)
plan = T.Plan(eng, cap, Cn)
rec = torch.zeros(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda")
hs = torch.cuda.current_stream().cuda_stream
for k in range(3):
    m = T.MultiSyncDev(eng, plan, streams, d.data_ptr(), offs, rec.data_ptr(), 64, hs)
    fused, ngrid = m.fused, m.ngrid
    m.collect()
torch.cuda.synchronize()
assert fused
st = np.zeros(5 * 16384, np.uint64)
assert T.lib().tgk_slot_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(-1, 5)
n = ((ngrid + 3) // 4 + 15) // 16           # the launch's tasks: 64 grid slots each
assert 0 < n <= 16384
st = st[:n]
assert (st[:, 1:] > 0).all(), "a task left no stamp"      # (a clock value of 0 would put the time axis at the box's uptime)
t0 = int(st[:, 1].min())
t = (st[:, 1:].astype(np.int64) - t0) / 100.0          # us
assert 0 < t[:, 3].max() < 5000.0, t[:, 3].max()       # a kernel of well under 5 ms: the 0.5 us grid below stays small
hw = st[:, 0].astype(np.uint32)
wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("tasks", n, "kernel span us", t[:, 3].max())
print("front phase us: median %.1f p10 %.1f p90 %.1f" % tuple(np.percentile(t[:, 1] - t[:, 0], [50, 10, 90])))
print("trellis phase us: median %.1f p10 %.1f p90 %.1f" % tuple(np.percentile(t[:, 2] - t[:, 1], [50, 10, 90])))
print("finish us: median %.1f" % np.median(t[:, 3] - t[:, 2]))
print("wave ids seen", np.bincount(wave.astype(np.int64)))
# first generation: start times by wave id
g1 = t[:, 0] < 1.0
print("tasks starting within 1 us:", int(g1.sum()))
# over the whole chip: how many waves are alive, and what share of them is in the front phase, every 10 us -- waves in step with each
# other show as a share that swings between 0 and 1 with the generations, waves out of step as a steady one
grid = np.arange(0.0, float(t[:, 3].max()), 10.0)
assert len(grid) < 1000
alive = ((t[:, 0][:, None] <= grid[None, :]) & (grid[None, :] < t[:, 3][:, None])).sum(0)
infront = ((t[:, 0][:, None] <= grid[None, :]) & (grid[None, :] < t[:, 1][:, None])).sum(0)
share = infront / np.maximum(alive, 1)
print("t (us)     :", " ".join("%4d" % x for x in grid[:48]))
print("alive waves:", " ".join("%4d" % x for x in alive[:48]))
print("in front % :", " ".join("%4d" % round(100 * x) for x in share[:48]))
mid = (grid > 0.1 * grid[-1]) & (grid < 0.8 * grid[-1])
print("share of waves in the front phase, 10 %% .. 80 %% of the kernel: mean %.2f min %.2f max %.2f" % (share[mid].mean(), share[mid].min(), share[mid].max()))
