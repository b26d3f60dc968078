"""round 6: what a caller who waits for every batch sees -- tgpu_sync_multi_launch + collect of one 1 M-slot batch at a time in a process
that holds ONE plan and ONE stream (the runtime puts all streams of a process on four hardware queues: in bench.py the plan's own
streams share queues with the eight pipelined batches' streams), for every TGPU_OPT_SLOT setting, side streams on / off.
usage: python tools/experiments/one_batch.py [slot modes ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import osmo_tetra_amd as T  # noqa: E402

modes = [int(x) for x in sys.argv[1:]] or [1, 3]
Cn, per, NB = 8, 125000, 3
bufs = []
for b in range(NB):
    streams, offs, o = [], [], 0
    for c in range(Cn):
        st, _, _ = bench.make_mix_stream(T, per, c + 1000 * b, mnc=42 + c, cc=1 + c)
        streams.append(st)
        offs.append(o)
        o += (len(st) + T.STREAM_SLACK + 15) & ~15
    buf = np.zeros(o + 4096, np.uint8)
    for st, f in zip(streams, offs):
        buf[f:f + len(st)] = st
    bufs.append(torch.from_numpy(buf).cuda())
eng = T.Engine(0)
cap = sum(len(st) // 510 + 32 for st in streams)
chans = T.multi_chan_table(streams, offs)
rec = torch.zeros(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
for mode in modes:
    for side in (True, False):
        T.set_option(T.OPT_SLOT, mode)
        plan = T.Plan(eng, cap, Cn)
        plan.set_side_stream(side)
        xs, forms = [], set()
        for k in range(40):
            t0 = time.perf_counter()
            m = T.MultiSyncDev(eng, plan, None, bufs[k % NB].data_ptr(), None, rec.data_ptr(), 64, stream.cuda_stream, chans=chans)
            forms.add(m.fused)
            m.collect_begin()
            m.collect_end(raw=True)
            xs.append((time.perf_counter() - t0) * 1e3)
        plan.close()
        xs = sorted(xs[8:])
        print("TGPU_OPT_SLOT %d, side streams %s: one batch at a time %.3f ms median (min %.3f, max %.3f), forms seen %s"
              % (mode, "on " if side else "off", xs[len(xs) // 2], xs[0], xs[-1], sorted(forms)))
