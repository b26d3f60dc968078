"""k_front_stream alone on the bench's batch of 8 channels: mean microseconds over nrep launches (HIP events around the
kernel).  Used with measurement builds of the library (tools/front_ablate.sh)."""
import sys, os
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
plan = T.Plan(eng, cap, Cn)
hs = torch.cuda.current_stream().cuda_stream
r = [T.sync_front_prof_multi(eng, plan, streams, d_base.data_ptr(), offs, 64, 20, hs) for _ in range(3)]
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join("front %.1f us fix %.1f us" % (a, b) for a, b in r))
if os.environ.get("FRONT_COLD"):
    # the same launch on NB copies of the batch in turn (as bench.py rotates its captures), one launch per measurement;
    # "dirty": 1.2 GB of other traffic between the launches (what the rest of a step puts through the caches)
    NB = 8
    bases = [d_base] + [d_base.clone() for _ in range(NB - 1)]
    junk = torch.empty(600_000_000, dtype=torch.uint8, device="cuda")
    chans = T.multi_chan_table(streams, offs)
    rec = torch.empty(cap * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan2 = T.Plan(eng, cap, Cn)
    for dirty in (0, 1, 2):
        xs = []
        for i in range(4 * NB):
            if dirty == 1:
                junk.add_(1)
            if dirty == 2:      # a whole step (walk, trellis kernels) in front of the launch, as in bench.py's per-kernel pass
                T.MultiSyncDev(eng, plan2, None, bases[(i + 3) % NB].data_ptr(), None, rec.data_ptr(), 64, hs, chans=chans).collect(raw=True)
            xs.append(T.sync_front_prof_multi(eng, plan, streams, bases[i % NB].data_ptr(), offs, 64, 1, hs))
        xs = xs[NB:]
        print("   rotating %d copies%s: front %.1f us fix %.1f us" % (NB, ["", ", caches dirtied between launches", ", a whole decode step between launches"][dirty],
              sum(a for a, _ in xs) / len(xs), sum(b for _, b in xs) / len(xs)))
