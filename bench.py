#!/usr/bin/env python3
"""bench.py -- decoded bursts/s of the MI355X TETRA lower-MAC receive path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default workload (config.workload) = BASELINE.json's metric: the SB+NDB mix through the burst-sync front end
(configs[2]'s composition in configs[3]'s layout): per GPU 8 recorded channels of 125,000 slots each (own cell each),
frames of 8 slots [SB, N1, N2, N1, N2, N1, N2, N1], cell scrambling code learnt from SB1, 1 % of the slots with a
damaged training sequence (dropped burst, loss of lock, re-lock), 1 bit per byte, resident in HBM.  One step = one
pass of the whole path over all of a GPU's channels as ONE batch, enqueued by ONE call (tgpu_sync_multi_launch): GPU
training-sequence search + demux/de-interleave of every grid slot, the reference's synchroniser walk of every channel
ON THE GPU (k_walk, tetra_burst_sync_in() semantics at 64-byte feeds), device-built lists, SB1 -> code fill -> masks ->
both trellis kernels; records stay in HBM.  value counts DELIVERED bursts (what tetra_burst_rx_cb() would have been
handed), not grid slots.  One host thread per GPU keeps `--depth` (8) steps in flight, each on a stream of its own, all
of a batch's kernels on that one stream (tgpu_plan_set_side_stream(plan, 0)): the HIP runtime multiplexes a process's
streams onto four hardware queues, and a batch that is split over two streams spreads the batches unevenly over them
(round 4; `round3_form` in the line is the old arrangement measured beside it).

Timing: ONE continuous run of warm-up + windows x K steps + tail; a HIP event behind every step's last operation.
Batches in flight share the GPU and complete in bunches, so a window boundary is the MEAN completion time of the D most
recent steps (D = steps in flight): window r = (boundary(w + (r + 1) K) - boundary(w + r K)) / K, exactly K classifications,
K walks and K decodes per window whatever the phase of the bunches; ms_per_step = the median window; the windows, the
same windows without the averaging and the contract's own form (K steps between two synchronisations, ramp-up and drain
included) are all in the line (timing.*).

The JSON line also carries
  roofline     : the dominant kernel's algorithmic bytes / its HIP-event duration vs HBM peak (+ PMC traffic)
  cpu_baseline : the oracle's receiver (CPU restatement of tetra-rx's path) on channel 0's stream, 64-byte feeds,
                 one thread, with the Viterbi the reference really runs (libosmocore's accelerated form), the
                 generic one beside it, and the same receiver on every usable host core at once (all_cores)
  end_to_end   : host buffer -> H2D -> the same step -> D2H of wire records -> a callback per delivered record
  breakdown_ms : host CPU per step (process_time), the per-kernel HIP-event durations
  config2      : BASELINE configs[1] (1 M aligned NDB bursts, no sync front end) as a secondary measurement
  N > 1        : single_gpu_reference (rank 0 alone, same job), decode_only and gathered (every step's 40-byte wire
                 blocks to rank 0 in the compact transport form -- delivered bursts only, csrc/tg_cwire.h -- through the
                 library's tgpu_comm_gatherv, RCCL, on a stream of its own; under a watchdog), per_gpu_efficiency of
                 both; value = the decode rate of all ranks (no data-path collective), `gathered` beside it
--workload config5 | conv | config2: the other BASELINE configs / the generic trellis (own roofline, cpu_baseline).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per burst = 510 B slot in + type-1 bits out (1 B/bit) + 16 B record per block
ALG_BYTES = {0: 510 + 14 + 268 + 2 * 16, 1: 510 + 14 + 124 + 124 + 3 * 16, 3: 510 + 60 + 14 + 124 + 3 * 16}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(slots, types, budget_s=12.0, all_cores_s=4.0):
    """time the oracle (tests/ infrastructure, checker only) on a bounded sample, one thread"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oraclelib as O
    lib = O.lib()
    # a -march=native build of the same restatement, made on this box when gcc is there
    try:
        tmp = os.path.join(tempfile.gettempdir(), "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-std=gnu11",
                               os.path.join(ROOT, "oracle", "tetra_oracle.c"), "-o", tmp],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = C.CDLL(tmp)
        nat.orc_bench_decode_slots.restype = C.c_uint64
        nat.orc_bench_decode_slots.argtypes = lib.orc_bench_decode_slots.argtypes
        lib, build = nat, "gcc -O3 -march=native"
    except Exception:
        build = "gcc -O3 (prebuilt)"
    chunk, done, t0 = 20000, 0, time.perf_counter()
    while True:
        lo = done % (len(types) - chunk)
        s = np.ascontiguousarray(slots[lo:lo + chunk])
        t = np.ascontiguousarray(types[lo:lo + chunk])
        lib.orc_bench_decode_slots(O._p(s), O._p(t), chunk, 0, 0, None, None)
        done += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    out = {"value": done / el, "unit": "bursts/s", "cores": 1, "kind": "port",
           "sample": f"{done} bursts of the same workload in {el:.1f} s, oracle/tetra_oracle.c ({build}), "
                     f"generic libosmocore Viterbi restatement, no callbacks/printing",
           "host_cores_available": os.cpu_count()}
    # the same port on every host core at once (one thread per core, disjoint slices of the same
    # workload, the C call releases the GIL): the "one process per channel" deployment of the reference
    try:
        import threading
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:        # a container may see every core of the box and still be throttled to a few of them
            q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                ncores = max(1, min(ncores, int(int(q) / int(per_us))))
        except Exception:
            pass
        ncores = min(ncores, 64)      # keeps the leg to seconds and the slices inside the sample
        per = min(max(2000, int(out["value"] * min(all_cores_s, budget_s))), len(types) // 2)   # slices may overlap: read-only
        pieces = []
        for i in range(ncores):
            lo = (i * per) % max(1, len(types) - per)
            pieces.append((np.ascontiguousarray(slots[lo:lo + per]), np.ascontiguousarray(types[lo:lo + per])))
        ths = [threading.Thread(target=lib.orc_bench_decode_slots, args=(O._p(a), O._p(b), per, 0, 0, None, None))
               for a, b in pieces]
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        el2 = time.perf_counter() - t1
        out["all_cores"] = {"value": ncores * per / el2, "unit": "bursts/s", "cores": ncores,
                            "sample": f"{ncores} threads x {per} bursts in {el2:.1f} s "
                                      f"(speed-up {ncores * per / el2 / out['value']:.1f}x over one thread; "
                                      f"{os.cpu_count()} logical CPUs visible)"}
    except Exception as e:       # the single-thread figure stands on its own
        out["all_cores"] = {"error": repr(e)}
    return out


def cpu_baseline_stream(stream, budget_s=7.0):
    """the oracle's receiver (orc_rx_feed: synchroniser + lower MAC, no callbacks, 64-byte feeds like tetra-rx.c:82-95)
    on a prefix of the same stream; one thread; both restated libosmocore decoders"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oraclelib as O
    lib = O.lib()
    build = "gcc -O3 (prebuilt)"
    try:
        tmp = os.path.join(tempfile.gettempdir(), "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-std=gnu11",
                               os.path.join(ROOT, "oracle", "tetra_oracle.c"), "-o", tmp],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = C.CDLL(tmp)
        nat.orc_rx_init.argtypes = lib.orc_rx_init.argtypes
        nat.orc_rx_feed.argtypes = lib.orc_rx_feed.argtypes
        lib, build = nat, "gcc -O3 -march=native"
    except Exception:
        pass
    res = {}
    for acc in (1, 0):
        # size the sample from a short probe so that the leg stays inside its budget
        n = 2000 * 510
        rate = None
        for _ in range(2):
            rx = O.Rx()
            lib.orc_rx_init(C.byref(rx), O.UPPER_CB(), O.EVENT_CB(), None)
            rx.use_acc = acc
            piece = np.ascontiguousarray(stream[:min(n, len(stream))])
            t0 = time.perf_counter()
            lib.orc_rx_feed(C.byref(rx), O._p(piece), len(piece), 64)
            el = time.perf_counter() - t0
            rate = rx.burst_seq / el
            n = int(min(len(stream), max(n, rate * budget_s * 510)))
        # a channel's stream is a second's worth of CPU work: further passes over it (a fresh receiver each) until the leg
        # has run for about five seconds; the rate is bursts over time of all passes
        nb, tt, passes = int(rx.burst_seq), el, 1
        while acc == 1 and tt < 5.0 and n == len(stream):
            rx = O.Rx()
            lib.orc_rx_init(C.byref(rx), O.UPPER_CB(), O.EVENT_CB(), None)
            rx.use_acc = acc
            t0 = time.perf_counter()
            lib.orc_rx_feed(C.byref(rx), O._p(piece), len(piece), 64)
            tt += time.perf_counter() - t0
            nb += int(rx.burst_seq)
            passes += 1
        res[acc] = (nb / tt, nb, tt, passes)
    # the same receiver on every usable host core at once (one recorded channel per thread, like the reference's one
    # process per channel): what the box's CPUs deliver together
    ncpu = host_threads_default()
    allc = None
    try:
        nper = int(min(len(stream), max(2000 * 510, res[1][0] * 3.0 * 510)))
        piece = np.ascontiguousarray(stream[:nper])
        rxs = [O.Rx() for _ in range(ncpu)]
        for rx in rxs:
            lib.orc_rx_init(C.byref(rx), O.UPPER_CB(), O.EVENT_CB(), None)
            rx.use_acc = 1
        ths = [threading.Thread(target=lambda rx=rx: lib.orc_rx_feed(C.byref(rx), O._p(piece), len(piece), 64)) for rx in rxs]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        el_all = time.perf_counter() - t0
        allc = {"value": sum(int(rx.burst_seq) for rx in rxs) / el_all, "unit": "bursts/s", "cores": ncpu,
                "sample": f"{ncpu} threads, each the receiver above on {int(rxs[0].burst_seq)} bursts of the stream, {el_all:.1f} s"}
    except Exception as ex:      # pragma: no cover
        allc = {"error": repr(ex)}
    return {"value": res[1][0], "unit": "bursts/s", "cores": 1, "kind": "port", "all_cores": allc,
            "sample": f"{res[1][1]} bursts delivered in {res[1][3]} pass(es) over the first {res[1][1] // res[1][3] * 510 // 1000} kB of channel 0's stream (the bytes the GPU run had), "
                      f"{res[1][2]:.1f} s: oracle/tetra_oracle.c receiver ({build}; synchroniser + demux + descramble + "
                      f"de-interleave + de-puncture + Viterbi + CRC, no callbacks / printing), 64-byte feeds, one thread, "
                      f"libosmocore's accelerated Viterbi restated (what osmo_conv_decode() dispatches N=4, K=5 to)",
            "generic_viterbi": {"value": res[0][0], "unit": "bursts/s",
                                "sample": f"{res[0][1]} bursts in {res[0][2]:.1f} s with the generic conv.c form instead"},
            "host_cores_available": os.cpu_count()}


def host_threads_default():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # a container may see every core of the box and still be throttled to a few of them
        q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per_us))))
    except Exception:
        pass
    return n


def make_mix_stream(T, n, seed, mcc=262, mnc=42, cc=1, damaged=0.01, ber=0.0):  # noqa: D401
    """config 3's stream: 100 random lead-in bits, a lock-only SB, n slots in frames of 8, 1 % damaged training
    sequences, 700 pad bytes; `ber` = bit error rate in the coded fields (the payload the trellis kernels decode)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rng = np.random.default_rng(7 + seed)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    code = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3
    slots = T.synth_slots(np.concatenate([[3], types]).astype(np.uint8), seed=11 + seed, scramb_init=code,
                          mcc=mcc, mnc=mnc, cc=cc, ber=ber)
    bad = np.flatnonzero(rng.random(n) < damaged) + 1
    y = slots[0, 214:252].tolist()
    for i in bad:
        off = 214 if slots[i, 214:252].tolist() == y else 244
        slots[i, off + 5] ^= 1
    stream = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), slots.reshape(-1), np.zeros(700, np.uint8)])
    return stream, types, code


def pick_roofline_kernel(ms_by_kernel, alg_bytes):
    """the kernel the roofline object is about: the longest one -- and where several are within 5 % of the longest (the
    front end and the SCH/F trellis kernel are, since round 5), the one among those that moves the most algorithmic
    bytes: that is the kernel HBM bounds; the others' bound is vector issue and they are listed beside it.  Returns
    (kernel, the kernels that tied)"""
    longest = max(ms_by_kernel.values())
    tied = [k for k, v in ms_by_kernel.items() if v >= 0.95 * longest]
    return max(tied, key=lambda k: (alg_bytes.get(k, 0), ms_by_kernel[k])), tied


def _median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) & 1 else 0.5 * (xs[m - 1] + xs[m])


def bench_mix(args, T, torch, dist, rank, world, local):
    """the metric's workload (BASELINE configs[2] composition, laid out as configs[3]'s per-GPU share): every GPU has
    C recorded channels of its own, all of them in one batch per step.  One host thread per rank; a step is ONE call
    (tgpu_sync_multi_launch: classification, the synchroniser walks on the device, lists, decode -- nothing waited for
    on the host), `depth` steps are in flight on as many streams / plans."""
    import collections
    n, C = args.bursts, max(1, args.channels)
    per = n // C
    D = max(1, args.depth)      # (1: one step at a time -- what the profiles of single kernels are taken with)
    # INPUT ROTATION (round 5): a receiver sees every byte once (tetra-rx.c:82-95), so the steps in flight must not read the
    # same bytes -- NB distinct captures (own seed each: other payload bits, other damaged slots; the cells, and with them the
    # channel table, are the same), step k decodes capture k % NB.  Default NB = steps in flight: no two concurrent front
    # ends share a byte, and a capture comes round again only after NB x 510 MB of other input went through the caches.
    NB = max(1, args.input_buffers if args.input_buffers > 0 else D)

    def capture(b, ber):
        """capture number b of this rank: C channel recordings in one buffer (16-byte aligned starts, slack behind each)"""
        sts, cds = [], []
        for c in range(C):
            g = rank * C + c
            st, _, code = make_mix_stream(T, per, g + 1000 * b, mnc=42 + g, cc=1 + g % 60, ber=ber)
            sts.append(st)
            cds.append(code)
        fs, o = [], 0
        for st in sts:
            fs.append(o)
            o += (len(st) + T.STREAM_SLACK + 15) & ~15
        bb = np.zeros(o + 4096, np.uint8)
        for st, f in zip(sts, fs):
            bb[f:f + len(st)] = st
        return sts, cds, fs, bb

    t_gen = time.perf_counter()
    streams, codes, offs, buf = capture(0, args.ber)
    eng = T.Engine(local)
    if args.walk_wide:
        T.set_option(T.OPT_WALK_WIDE, 1)
    if args.slot_mode >= 0:
        T.set_option(T.OPT_SLOT, args.slot_mode)
    d_bases = [torch.from_numpy(buf).cuda()]
    for b in range(1, NB):
        sts_b, cds_b, offs_b, buf_b = capture(b, args.ber)
        assert cds_b == codes and offs_b == offs and [len(x) for x in sts_b] == [len(x) for x in streams]
        d_bases.append(torch.from_numpy(buf_b).cuda())
        del sts_b, buf_b
    d_base = d_bases[0]
    t_gen = time.perf_counter() - t_gen
    cap = sum(len(st) // 510 + 32 for st in streams)
    K, R, W = args.steps, max(1, args.windows), max(args.warmup, 6 * D) + D     # (the first ~20 steps of a run are 4 % slower: clocks, queues filling)
    chans = T.multi_chan_table(streams, offs)         # carry-in codes 0: every cell's code is learnt from SB1 inside the batch
    D2 = D
    plans = [T.Plan(eng, cap, C) for _ in range(D2)]
    # every batch on ONE stream (= one hardware queue): with several batches in flight the plans' side streams spread the batches
    # unevenly over the runtime's four hardware queues (tgpu_plan_set_side_stream; DESIGN.md section 5 has the A/B)
    for p_ in plans:
        p_.set_side_stream(args.side_stream)
    # (pre-set to a value no record holds: what a step fails to write must not read as the previous step's -- equal -- answer)
    recs = [torch.full((cap * T.REC_BYTES,), 0xa5, dtype=torch.uint8, device="cuda") for _ in range(D2)]
    strm = [torch.cuda.Stream() for _ in range(D2)]
    gather = world > 1 or args.force_gather
    nccl = args.backend == "nccl"
    state = {"fellback": 0, "ngrid": 0, "ccomm": None, "impl": None, "sizes": None}
    wires = sink = None
    compact = args.wire_form == "compact"
    cwcap = (T.cwire_bound(cap, C) + 255) & ~255       # a rank's share of the sink, and the capacity of a compact buffer
    cws = csink = cstream = gg = None                  # compact form: two buffers per plan, the sink, the exchange's own stream, the size group
    pend = collections.deque()                         # compact form: steps whose sizes are on their way round the ranks
    gdone = {}                                         # compact form: buffer -> event behind the gather that last read it

    def finish(ms):
        outs = ms.collect_end(raw=True)     # (collect_begin() has run: the batch's successor is launched in between)
        assert all(x["noffgrid"] == 0 for x in outs)
        state["fellback"] += int(ms.fellback)
        state["ngrid"] = ms.ngrid
        return sum(x["nslots"] for x in outs)

    def exchange(j):
        """this step's decoded blocks (40-byte wire records of the batch's grid slots) to rank 0, on the step's own stream:
        it waits for this decode only and runs beside the other streams' kernels"""
        nbytes = cap * T.WIRE_BYTES      # the same on every rank (recordings of equal length): grid slots + 32 slack slots per channel
        w = wires[j][:nbytes]
        if nccl and state["ccomm"] is not None:
            state["ccomm"].gather(w.data_ptr(), nbytes, sink[j].data_ptr() if rank == 0 else 0, 0, strm[j].cuda_stream)
        elif nccl:
            with torch.cuda.stream(strm[j]):
                dist.gather(w, gather_list=list(sink[j].view(world, -1)[:, :nbytes].unbind(0)) if rank == 0 else None, dst=0)
        else:       # control-flow check on a box with fewer GPUs than ranks: staged through the host
            strm[j].synchronize()
            dist.gather(w.cpu(), gather_list=list(sink[j].view(world, -1)[:, :nbytes].unbind(0)) if rank == 0 else None, dst=0)

    ge = args.gather_every if args.gather_every > 0 else (4 if (world > 1 and nccl and compact) else 1)
    G = max(1, min(ge, D))                             # compact form: steps whose blocks go to rank 0 in ONE exchange
    batch = []                                         # collected steps waiting for their batch to fill: (bytes, buffer, step)
    unposted = set()                                   # buffers of steps that are collected but not handed to the exchange yet
    nposted = [0]

    def post_compact(evs):
        """the oldest pending batch: its sizes have been round the ranks (control plane: the gloo group), now the payload --
        every rank's compact buffers of the batch's steps to rank 0, exact sizes, ONE grouped RCCL send / receive on the
        exchange's own stream (tgpu_comm_gatherv_batch; a batch of one step = tgpu_comm_gatherv)"""
        work, szs, items, slot = pend.popleft()
        work.wait()
        sizes = [[int(szs[r][m].item()) for r in range(world)] for m in range(len(items))]
        assert all(sizes[m][rank] == items[m][0] for m in range(len(items)))
        offs_ = [[(m * world + r) * cwcap for r in range(world)] for m in range(len(items))]
        state["sizes"], state["last_slot"], state["last_off"] = sizes[-1], slot, offs_[-1][0]
        if nccl and state["ccomm"] is not None:
            if len(items) == 1:
                state["ccomm"].gatherv(items[0][1].data_ptr(), sizes[0], csink[slot].data_ptr() if rank == 0 else 0,
                                       offs_[0] if rank == 0 else None, 0, cstream.cuda_stream)
            else:
                state["ccomm"].gatherv_batch([it[1].data_ptr() for it in items], sizes, csink[slot].data_ptr() if rank == 0 else 0,
                                             offs_ if rank == 0 else None, 0, cstream.cuda_stream)
        else:       # torch.distributed.gather wants one size: padded to the largest (nccl: --torch-gather; gloo: staged through the host)
            for m_, it in enumerate(items):
                m = (max(sizes[m_]) + 15) & ~15
                lst = list(csink[slot][offs_[m_][0]:offs_[m_][0] + world * cwcap].view(world, -1)[:, :m].unbind(0)) if rank == 0 else None
                with torch.cuda.stream(cstream):
                    if nccl:
                        dist.gather(it[1][:m], gather_list=lst, dst=0)
                    else:
                        cstream.synchronize()
                        dist.gather(it[1][:m].cpu(), gather_list=lst, dst=0)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(cstream)
        for it in items:
            gdone[it[1].data_ptr()] = ev
            unposted.discard(it[1].data_ptr())
            evs.append(ev)          # a step is complete when its blocks are on the collecting rank

    def flush_batch(evs):
        """the collected steps' compact sizes start their way round the ranks (asynchronous); the batch before is posted"""
        items = batch[:]
        del batch[:]
        mine = torch.tensor([it[0] for it in items] + [0] * (G - len(items)), dtype=torch.int64)
        szs = [torch.zeros(G, dtype=torch.int64) for _ in range(world)]
        work = dist.all_gather(szs, mine, group=gg, async_op=True)
        pend.append((work, szs, items, nposted[0] & 1))
        nposted[0] += 1
        while len(pend) > 1:
            post_compact(evs)

    def start_compact(ms, buf, k, evs):
        """a collected step joins the batch; a full batch is sent on its way"""
        assert ms.fellback or ms.cwire_bytes > 0
        batch.append((int(ms.cwire_bytes), buf, k))
        unposted.add(buf.data_ptr())
        if len(batch) == G:
            flush_batch(evs)

    def run(total, with_gather, D=D, S=None, bases=None):
        """`total` steps back to back, at most D in flight, step k on capture k % len(bases); returns (delivered bursts per
        step, completion events)"""
        bases = bases or d_bases
        evs, fl, delivered = [], collections.deque(), []
        S = S or D           # (S < D: plan j runs on stream j % S -- a stream then holds its next batch before the host has collected the last)
        for k in range(total):
            j = k % D
            old = None
            if len(fl) == D:            # wait for the oldest batch and take its outcome out of the plan's buffers ...
                old = fl.popleft()
                old[0].collect_begin()
            plans[j].set_wire(wires[j].data_ptr() if with_gather else 0)
            buf = None
            if with_gather and compact:
                buf = cws[j][(k // D) & 1]
                if buf.data_ptr() in unposted:  # (a batch that has not gone out yet still holds this buffer: send it on its way now)
                    if batch:
                        flush_batch(evs)
                    while pend:
                        post_compact(evs)
                if buf.data_ptr() in gdone:     # the gather that last read this buffer (2 D steps ago) must be through with it
                    gdone.pop(buf.data_ptr()).synchronize()
                plans[j].set_cwire(buf.data_ptr(), cwcap)
            else:
                plans[j].set_cwire(0)
            fl.append((T.MultiSyncDev(eng, plans[j], None, bases[k % len(bases)].data_ptr(), None, recs[j].data_ptr(), 64,
                                      strm[j % S].cuda_stream, chans=chans), buf, k))
            if with_gather and not compact:
                exchange(j)
            if not (with_gather and compact):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(strm[j % S])
                evs.append(ev)
            if old is not None:         # ... its successor is on the stream: now look at what it delivered
                if with_gather and compact:
                    start_compact(old[0], old[1], old[2], evs)
                delivered.append(finish(old[0]))
        while fl:
            old = fl.popleft()
            old[0].collect_begin()
            if with_gather and compact:
                start_compact(old[0], old[1], old[2], evs)
            delivered.append(finish(old[0]))
        if batch:
            flush_batch(evs)
        while pend:
            post_compact(evs)
        return delivered, evs

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    S0 = min(D, args.streams) if args.streams else None

    def measure(with_gather, alone=False, D=D, W=W, S=S0, K=K, R=R, bases=None):
        """one continuous run of W + R K + D steps (the pipeline stays full before, through and after the timed steps);
        window r = completion of step W - 1 + r K  ->  completion of step W - 1 + (r + 1) K: exactly K classifications,
        K walks, K decodes (and K exchanges) complete inside it.  Plus the contract's form: K steps between two
        synchronisations (ramp-up and drain included)."""
        nb = len(bases or d_bases)
        run(max(D, K if not alone else D), with_gather, D, S, bases)    # allocations, first-use paths, clocks
        if not alone:
            sync_all()
        else:
            torch.cuda.synchronize()
        gc.collect()
        gc.disable()          # (a collection of the interpreter's in the middle of the run is a multi-millisecond hole in one window)
        c0, t0, h0 = time.process_time(), time.perf_counter(), time.thread_time()
        try:
            delivered, evs = run(W + R * K + D, with_gather, D, S, bases)
        finally:
            gc.enable()
        torch.cuda.synchronize()
        cpu_s, wall_s, thr_s = time.process_time() - c0, time.perf_counter() - t0, time.thread_time() - h0
        base = W - 1 - (D - 1)
        tk = [evs[base].elapsed_time(e) for e in evs[base:]]
        for i in range(1, len(tk)):               # (steps run on D streams: completion times are made monotone)
            tk[i] = max(tk[i], tk[i - 1])
        raw = [(tk[D - 1 + (r + 1) * K] - tk[D - 1 + r * K]) / K for r in range(R)]
        # steps complete in bunches (the D batches in flight share the GPU and finish together): a window boundary is the MEAN
        # completion time of the D most recent steps -- the D spans of K steps that end in a boundary start at all D phases of
        # the pattern once, so the estimate does not depend on where in a bunch the window happens to be cut
        sm = [sum(tk[i - D + 1:i + 1]) / D for i in range(D - 1, len(tk))]
        win = [(sm[(r + 1) * K] - sm[r * K]) / K for r in range(R)]
        tk = tk[D - 1:]
        if os.environ.get("BENCH_STEP_TIMES"):
            print("step completion deltas (ms), depth %d:" % D, [round(tk[i + 1] - tk[i], 3) for i in range(min(48, len(tk) - 1))], file=sys.stderr)
        assert all(x == delivered[i % nb] for i, x in enumerate(delivered)), "the same recording gave different numbers of bursts"
        per_step = sum(delivered[W:W + R * K]) / float(R * K)      # (captures differ in their damaged slots: the mean over the timed steps)
        if not alone:
            sync_all()
        t1 = time.perf_counter()
        run(K, with_gather, D, S, bases)
        if not alone:
            sync_all()
        else:
            torch.cuda.synchronize()
        bracket = (time.perf_counter() - t1) / K * 1e3
        if world > 1 and not alone:
            dev = "cuda" if nccl else "cpu"
            t = torch.tensor(win + [bracket], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            win, bracket = t[:-1].tolist(), float(t[-1].item())
            d = torch.tensor([float(per_step)], dtype=torch.float64, device=dev)
            dist.all_reduce(d, op=dist.ReduceOp.SUM)
            tot = float(d.item())
        else:
            tot = per_step
        med = _median(win)
        if world > 1 and not alone:
            t = torch.tensor(raw, dtype=torch.float64, device="cuda" if nccl else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            raw = t.tolist()
        return {"value": tot / (med * 1e-3), "ms_per_step": med, "windows_ms_per_step": [round(x, 5) for x in win],
                "unsmoothed_windows_ms_per_step": [round(x, 5) for x in raw],
                "window_spread": (max(win) - min(win)) / med, "all_windows_ms_per_step": (tk[R * K] - tk[0]) / (R * K),
                "sync_bracketed_ms_per_step": bracket, "bursts_delivered_per_step": tot,
                "host_cpu_ms_per_step": cpu_s / (W + R * K + D) * 1e3, "host_wall_ms_per_step": wall_s / (W + R * K + D) * 1e3,
                "launch_thread_cpu_ms_per_step": thr_s / (W + R * K + D) * 1e3}

    single = None
    if world > 1:
        # what ONE of these GPUs does on its own, measured in this very job (rank 0 alone, the others wait): the
        # reference the per-GPU efficiencies below are quoted against
        if rank == 0:
            single = measure(False, alone=True)
        sync_all()
    if args.trace_dump:     # (a -DTG_TRACE build of the library: tools/experiments/trace_untraced.py)
        import ctypes
        ctypes.CDLL(None)
        T.lib().tgk_trace_read(None, ctypes.byref(ctypes.c_uint(0)), 1)
    decode_only = measure(False)
    if args.trace_dump:
        nrec = ctypes.c_uint(0)
        buf = np.zeros((1 << 20, 3), np.uint64)
        T.lib().tgk_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nrec), 1)
        np.save(args.trace_dump, buf[:nrec.value])
    # what a caller who waits for every batch sees: launch, collect, next -- one batch in flight, rotating captures
    one_at_a_time = None
    if world == 1:
        xs = []
        for k in range(28):
            t0_ = time.perf_counter()
            m_ = T.MultiSyncDev(eng, plans[0], None, d_bases[k % NB].data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
            m_.collect_begin()
            m_.collect_end(raw=True)
            xs.append((time.perf_counter() - t0_) * 1e3)
        one_at_a_time = {"ms_per_batch_median": _median(xs[4:]), "ms_per_batch_max": max(xs[4:]), "batches": len(xs) - 4,
                         "note": "tgpu_sync_multi_launch + tgpu_sync_multi_collect of one 1 M-slot batch at a time (host wall clock, the launch "
                                 "call, all kernels, the outcome's way down and the wait included), in the timed run's own arrangement (every "
                                 "kernel of a batch on one stream)"}
        # ... and in the arrangement a caller who WAITS for every batch would choose: the plan's side streams in play (the library's
        # default) and TGPU_OPT_SLOT 3 -- the trellises of every plain slot start behind the front end on the channels' hinted codes
        # (k_slot_e) and run BESIDE the walk and the code look-back
        try:
            if args.no_latency_form:
                raise RuntimeError("skipped (--no-latency-form)")
            keep = int(T.get_option(T.OPT_SLOT))
            T.set_option(T.OPT_SLOT, 3)
            plan_l = T.Plan(eng, cap, C)
            ys, early = [], 0
            for k in range(28):
                t0_ = time.perf_counter()
                m_ = T.MultiSyncDev(eng, plan_l, None, d_bases[k % NB].data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
                early += int(m_.fused == 2)
                m_.collect_begin()
                m_.collect_end(raw=True)
                ys.append((time.perf_counter() - t0_) * 1e3)
            plan_l.close()
            T.set_option(T.OPT_SLOT, keep)
            one_at_a_time["latency_form"] = {"ms_per_batch_median": _median(ys[4:]), "ms_per_batch_max": max(ys[4:]), "batches": len(ys) - 4,
                                             "batches_with_the_trellises_beside_the_walk": early,
                                             "note": "the same, with the plan's side streams in play and TGPU_OPT_SLOT 3 (k_slot_e: the trellises of "
                                                     "every plain slot on the hinted codes, beside the walk; k_slot_t re-decodes what the look-back hands "
                                                     "back) -- in THIS process the plan's streams share the runtime's four hardware queues with the eight "
                                                     "pipelined batches' streams; tools/experiments/one_batch.py is the same in a process of its own"}
        except Exception as ex:      # pragma: no cover
            one_at_a_time["latency_form"] = {"error": repr(ex)}
    # what the rotation is worth: the same run with every step in flight on ONE capture (rounds 1-4 measured this)
    one_input = None
    if world == 1 and not args.no_secondary and NB > 1:
        one_input = measure(False, bases=[d_base], R=max(2, R // 2))
    # the rate does not depend on the payload: the same run on captures with bit errors in the coded fields (own seeds), with
    # its own oracle check of delivered records (below)
    ber2 = ber2_streams = ber2_base = None
    if world == 1 and not args.no_secondary and args.ber_secondary > 0:
        bases2 = []
        for b in range(max(1, min(NB, args.ber_buffers))):
            sts_b, cds_b, offs_b, buf_b = capture(500 + b, args.ber_secondary)
            assert cds_b == codes and offs_b == offs
            bases2.append(torch.from_numpy(buf_b).cuda())
            if b == 0:
                ber2_streams = sts_b
            del buf_b
        ber2 = measure(False, bases=bases2, R=max(2, R // 2))
        ber2_base = bases2[0]
        del bases2
    # the round-3 form beside it: four batches in flight, each plan's side stream in play
    r3form = None
    if world == 1 and not args.no_secondary and not args.side_stream:
        for p_ in plans:
            p_.set_side_stream(True)
        r3form = measure(False, D=min(4, D))
        for p_ in plans:
            p_.set_side_stream(False)
    gathered = gather_error = None
    if gather:
        armed = [True]

        def bail():
            if not armed[0]:
                return
            if rank == 0:
                print(json.dumps({"metric": "decoded bursts/s", "value": decode_only["value"], "unit": "bursts/s", "n_gpus": world,
                                  "steps": K, "warmup": args.warmup, "ms_per_step": decode_only["ms_per_step"],
                                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
                                  "data": "synthetic",
                                  "config": {"workload": "SB+NDB mix through the burst-sync front end, %d channels per GPU "
                                                         "(decode only: the per-step gather to rank 0 did not finish "
                                                         "within %d s and was abandoned)" % (C, args.gather_timeout)},
                                  "decode_only": decode_only, "single_gpu_reference": single,
                                  "gathered": {"error": "timeout after %d s" % args.gather_timeout}}), flush=True)
            os._exit(0)

        tmr = threading.Timer(args.gather_timeout, bail)
        tmr.daemon = True
        tmr.start()
        try:
            wires = [torch.full((cap * T.WIRE_BYTES,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(D)]
            sink = [torch.empty(world * cap * T.WIRE_BYTES, dtype=torch.uint8, device="cuda" if nccl else "cpu")
                    for _ in range(D)] if rank == 0 else [None] * D
            if compact:
                cws = [[torch.zeros(cwcap, dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(D)]
                csink = [torch.zeros(G * world * cwcap, dtype=torch.uint8, device="cuda" if nccl else "cpu") for _ in range(2)] if rank == 0 else [None] * 2
                cstream = torch.cuda.Stream()
                gg = dist.new_group(backend="gloo") if nccl else None      # the sizes' way round the ranks (control plane)
            state["impl"] = "torch.distributed.gather (gloo, staged through the host)"
            if nccl and not args.torch_gather:
                try:     # the library's communicator: rank 0 draws the id, the process group carries its 128 bytes
                    uid = torch.from_numpy(T.comm_unique_id() if rank == 0 else np.zeros(T.COMM_ID_BYTES, np.uint8)).cuda()
                    dist.broadcast(uid, 0)
                    torch.cuda.synchronize()
                    state["ccomm"] = T.Comm(eng, uid.cpu().numpy(), rank, world)
                    state["impl"] = ("tgpu_comm_gatherv (C ABI, grouped RCCL send / receive, a size per rank; the sizes go round the ranks "
                                     "on a gloo group one step ahead of the payload)" if compact else
                                     "tgpu_comm_gather (C ABI, grouped RCCL send / receive)")
                except Exception as ex:      # pragma: no cover
                    state["ccomm"] = None
                    state["impl"] = "torch.distributed.gather (nccl); tgpu_comm_create failed: %r" % (ex,)
            elif nccl:
                state["impl"] = "torch.distributed.gather (nccl)"
            gathered = measure(True)
        except Exception as ex:      # pragma: no cover
            gather_error = repr(ex)
        armed[0] = False
        tmr.cancel()
    if gathered:
        sent = max(state["sizes"]) if compact else cap * T.WIRE_BYTES
        ref_ms = (single or decode_only)["ms_per_step"]
        gathered["exchange"] = state["impl"]
        gathered["wire_form"] = ("compact (csrc/tg_cwire.h): delivered bursts only, 36 / 33 / 25 bytes per NORM_1 / NORM_2 / SYNC burst (41 with a flag or "
                                 "a failed CRC) + the delivered bitmap; made on the device behind every step's decode (k_cw_*)" if compact else
                                 "grid: one 40-byte wire record per grid slot")
        gathered["bytes_per_rank_and_step"] = sent
        gathered["steps_per_exchange"] = G if compact else 1
        # one direction of one xGMI link (~77 GB/s) carries a peer's blocks to rank 0: the step cannot be shorter than that takes
        gathered["link_bound"] = {"xgmi_one_direction_gbs": 77.0, "min_ms_per_step": sent / 77e9 * 1e3,
                                  "max_per_gpu_efficiency": min(1.0, ref_ms / (sent / 77e9 * 1e3)),
                                  "note": "bytes_per_rank_and_step over one direction of the peer's own link to rank 0; efficiency bound = the "
                                          "single-GPU step time (%.3f ms) / that, capped at 1 -- whatever RCCL does, `gathered` cannot scale better" % ref_ms}
        if compact:
            gathered["bytes_per_rank_and_step_all_ranks"] = state["sizes"]
            gathered["bytes_per_delivered_burst"] = sent / (gathered["bursts_delivered_per_step"] / world)
        gathered["link_arithmetic"] = ("every peer sends %.1f MB per step to rank 0 over its own xGMI link: %.1f GB/s per link at the "
                                       "single-GPU step time (%.3f ms) = %.2f of the ~77 GB/s one direction of a link offers, %.1f GB/s "
                                       "at the gathered step time; rank 0 takes in %.1f GB/s in total (the grid form would be %.1f MB: %.1f GB/s)"
                                       % (sent / 1e6, sent / 1e6 / ref_ms, ref_ms, sent / 1e6 / ref_ms / 77.0, sent / 1e6 / gathered["ms_per_step"],
                                          (world - 1) * sent / 1e6 / gathered["ms_per_step"], cap * T.WIRE_BYTES / 1e6,
                                          cap * T.WIRE_BYTES / 1e6 / ref_ms))
    if gathered and NB > 1:     # the check below compares the collecting rank's copy with a step on capture 0: make that the last one gathered
        run(D, True, D, S0, [d_base])
        sync_all()
    if rank != 0:
        return None
    hs = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    # one more step, collected in full: the outcome of the timed path for the checks below
    plans[0].set_wire(wires[0].data_ptr() if gathered else 0)
    if gathered and compact:
        plans[0].set_cwire(cws[0][0].data_ptr(), cwcap)
    torch.cuda.synchronize()
    recs[0].fill_(0xa5)         # (the checked step writes into records that hold nothing of the timed steps' -- equal -- output)
    torch.cuda.synchronize()
    ms = T.MultiSyncDev(eng, plans[0], None, d_base.data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
    outs = ms.collect()
    torch.cuda.synchronize()
    # (a batch the device walk hands to the host walks -- a channel beyond the walk kernel's capacity: --channels 1 / 2 put 1 M /
    # 500 k slots into one channel -- is decoded all the same, at the host walk's pace; the line says how many there were)
    handed_over = state["fellback"] + int(ms.fellback)
    check = None
    rec_all = recs[0].view(-1, T.REC_BYTES)

    def oracle_check(sts, outs_, with_wire, noisy=False):
        """correctness guard on the timed output: delivered bursts of every channel against the oracle (checker only) --
        type-1 bits, BBK, CRC words and codes of up to 512 delivered grid slots per channel; returns (bursts checked,
        blocks whose CRC failed among them)"""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oraclelib as O
        nchk = nbad = 0
        wgrid = None
        for c, (st, out) in enumerate(zip(sts, outs_)):
            first = T.grid_indices(out)[:512]
            p = T.parse_records(rec_all[torch.from_numpy(out["grid_base"] + first).cuda()].cpu().numpy())
            anchor = out["anchor"]
            sl = np.stack([st[anchor + 510 * int(g):anchor + 510 * int(g) + 510] for g in first])
            ty = p["type"].astype(np.uint8)
            ok, want, wcrc = O.bench_decode_slots(sl, ty, codes[c], use_acc=1, want_out=True, want_crc=True)
            n1, n2, sb = ty == 0, ty == 1, ty == 3
            # (with payload errors a channel's first SB1 may fail its CRC: the bursts in front of the first good SYNC PDU are
            # decoded under the carry-in code 0, by the reference as by this path -- the oracle call below knows one code, so
            # those are left out; with BER 0 every burst is in)
            kn = p["code"] == codes[c]
            assert kn.mean() > (0.9 if noisy else 0.999), "channel %d: %d of %d bursts not under the cell's code" % (c, int((~kn).sum()), len(kn))
            n1, n2, sb = n1 & kn, n2 & kn, sb & kn
            parts = {"bbk": (p["bbk"][kn] == want[kn, :14]).all(), "SCH/F bits": (p["bits1"][n1] == want[n1, 14:282]).all(),
                     "NDB block 1": (p["bits1"][n2][:, :124] == want[n2, 14:138]).all(), "NDB block 2": (p["bits2"][n2] == want[n2, 138:262]).all(),
                     "SB1": (p["bits1"][sb][:, :60] == want[sb, 14:74]).all(), "SB2": (p["bits2"][sb] == want[sb, 138:262]).all(),
                     "crc 1": (p["crc"][kn, 0] == wcrc[kn, 0]).all(), "crc 2": (p["crc"][n2 | sb, 1] == wcrc[n2 | sb, 1]).all(),
                     "types": ((ty == 0) | (ty == 1) | (ty == 3)).all()}
            assert all(parts.values()), "decoded records of channel %d differ from the oracle: %s" % (c, [k for k, v in parts.items() if not v])
            nchk += int(kn.sum())
            nbad += int((p["crc_ok"][kn, 0] == 0).sum() + (p["crc_ok"][n2 | sb, 1] == 0).sum())
            if with_wire:        # ... and what arrived on the collecting rank is this rank's share, byte for byte
                idx = torch.from_numpy(out["grid_base"] + first)
                if compact:     # (the last gathered step decoded capture 0: rank 0's share of that gather is this step's buffer)
                    if c == 0:
                        got = csink[state["last_slot"]][state["last_off"]:state["last_off"] + ms.cwire_bytes].cpu().numpy()
                        mine_cw = cws[0][0][:ms.cwire_bytes].cpu().numpy()
                        assert ms.cwire_bytes == state["sizes"][0] and (got == mine_cw).all()
                        wgrid, _ = T.cwire_expand(got)
                    w0 = torch.from_numpy(wgrid)
                else:
                    w0 = sink[0].view(world, -1)[0].view(-1, T.WIRE_BYTES)
                assert torch.equal(w0[idx.to(w0.device)].cpu(), wires[0].view(-1, T.WIRE_BYTES)[idx.cuda()].cpu())
                back = T.wire_unpack(w0[idx.to(w0.device)].cpu().numpy(), (out["grid_base"] + first).tolist(), [codes[c]] * len(first))
                pw = T.parse_records(back)
                assert (pw["bbk"] == p["bbk"]).all() and (pw["crc"][:, 0] == p["crc"][:, 0]).all()
        return nchk, nbad

    if not args.no_cpu_baseline:
        nchk, _ = oracle_check(streams, outs, bool(gathered))
        check = "type-1 bits, BBK, CRC words and scrambling codes of %d delivered bursts (all %d channels) equal the oracle's%s" % (
            nchk, C, "; the same bursts' wire records on the collecting rank equal the sender's" if gathered else "")
        if ber2:
            plans[0].set_wire(0)
            plans[0].set_cwire(0)
            recs[0].fill_(0xa5)
            torch.cuda.synchronize()
            ms2 = T.MultiSyncDev(eng, plans[0], None, ber2_base.data_ptr(), None, recs[0].data_ptr(), 64, strm[0].cuda_stream, chans=chans)
            outs2 = ms2.collect()
            torch.cuda.synchronize()
            n2_, bad2 = oracle_check(ber2_streams, outs2, False, noisy=True)
            ber2["check"] = "type-1 bits, BBK, CRC words and codes of %d delivered bursts equal the oracle's (%d of their blocks fail the CRC on both sides)" % (n2_, bad2)
            del ms2, outs2
    ber2_base = ber2_streams = None

    # per-kernel durations: the same step with HIP events between all of its stages, on the launch stream
    prof = T.Prof(8)
    dev = []
    for q in range(8):
        dev.append(T.sync_multi_launch_prof(eng, plans[1], chans, d_bases[(7 - q) % NB].data_ptr(), recs[1].data_ptr(), prof, q, 64, hs))   # (the last pass leaves capture 0's records in recs[1])
    st_ms = prof.read(8)[2:].mean(axis=0)
    names = T.Prof.stage_names()
    kern_ms = {k: float(np.mean([x[k] for x in dev[2:]])) for k in dev[0]}
    for i in range(1, len(names)):
        if names[i] not in kern_ms:          # (SB1, code fill and masks are stages in front of the walk on this path)
            kern_ms[names[i]] = float(st_ms[i])
    kern_ms = {k: v for k, v in kern_ms.items() if not (k in ("k_fill", "k_masks") and v < 8e-3)}
    # round 6 (TGPU_OPT_SLOT >= 1): the two trellis stages are ONE launch, k_slot_t (one lane per slot, both lists), timed in the first
    # one's place; with the option at 2 and hints in place the front-end stage is k_slot (front end + trellises in one launch)
    slot_mode = int(T.get_option(T.OPT_SLOT))
    empty_bracket_ms = None
    if slot_mode >= 1 and "k_vit<216>" in kern_ms:
        # (the launch sits between the first stage's two events; the second stage's bracket is EMPTY -- two event records back to back --
        # and says what a bracket costs by itself: reported beside kernel_ms, not added to it)
        kern_ms["k_slot_t"] = kern_ms.pop("k_vit<216>")
        empty_bracket_ms = kern_ms.pop("k_vit<432>", None)
    fused_front = slot_mode >= 2 and kern_ms.get("k_slot_t", 1.0) < 0.5 * kern_ms.get("k_front_stream", 0.0)
    if fused_front:
        kern_ms["k_slot"] = kern_ms.pop("k_front_stream")
    ngrid = sum(x["ngrid"] for x in outs)
    nd = sum(x["nslots"] for x in outs)
    n_sb, n_n1, n_n2 = nd // 8, nd // 2, nd - nd // 8 - nd // 2     # delivered bursts by type (the damage is uniform)
    # 510 B in per slot the front end looks at; type-1 bits at 1 B/bit + 16 B per block out
    alg = {"k_front_stream": ngrid * 510,
           "k_vit<SB1>": n_sb * (60 + 16),
           "k_vit<216>": n_n2 * (14 + 124 + 124 + 3 * 16) + n_sb * (14 + 124 + 2 * 16),
           "k_vit<432>": n_n1 * (14 + 268 + 2 * 16)}
    # k_slot_t writes every block of a delivered burst (a SYNC burst's SB1 included); k_slot reads the grid as well
    alg["k_slot_t"] = alg["k_vit<216>"] + alg["k_vit<432>"] + alg["k_vit<SB1>"]
    alg["k_slot"] = alg["k_front_stream"] + alg["k_slot_t"]
    dom, tied = pick_roofline_kernel(kern_ms, alg)
    hbk = "k_front_stream" if ("k_front_stream" in kern_ms and dom != "k_front_stream") else None
    achieved = alg.get(dom, 0) / (kern_ms[dom] * 1e-3) / 1e9
    traffic = valu_busy = valu_pipe = traffic_prov = None
    per_kernel = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        # the pipelined configuration's own counters where they exist (mix_depth8: PMC passes of the run with 8 steps in flight
        # on rotating captures), else the one-step-at-a-time passes
        tsrc = "mix_depth8" if dom in tj.get("mix_depth8", {}) else "mix"
        traffic = tj.get(tsrc, {}).get(dom)
        traffic_prov = {"key": tsrc, "provenance": tj.get("_%s_provenance" % tsrc)}
        valu_busy = tj.get("mix_valu_busy", {}).get(dom)
        vi = tj.get("mix_valu_insts_per_step")
        # every heavy kernel against both of its bounds: HBM (algorithmic bytes / this run's duration) and vector issue (the
        # profile's wave instructions x 4 cycles / 1024 SIMDs / this run's duration at the profile's clock)
        per_kernel = {k: {"ms": kern_ms[k], "hbm_frac": alg[k] / (kern_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "vector_issue_frac (instruction count from profiles/traffic.json)":
                              (vi[k] * 4.0 / (1024.0 * kern_ms[k] * 1e-3 * tj.get("mix_sclk_ghz", 2.3) * 1e9)) if vi and k in vi else None}
                      for k in alg if k in kern_ms}
        if vi:      # the whole pipelined step against what 1024 SIMDs can issue: every wave instruction at its 4-cycle minimum
            n_in = float(sum(vi.values()))
            valu_pipe = {"wave_instructions_per_step": int(n_in), "sclk_ghz": tj.get("mix_sclk_ghz", 2.3),
                         "frac_of_issue_capacity": n_in * 4.0 / (1024.0 * decode_only["ms_per_step"] * 1e-3 * tj.get("mix_sclk_ghz", 2.3) * 1e9),
                         "note": "SQ_INSTS_VALU per launch of the step's kernels (profiles/traffic.json, rocprofv3 PMC pass of the same "
                                 "workload) x 4 cycles / (1024 SIMDs x this run's ms_per_step x the shader clock the profile saw)"}
    except Exception:
        pass

    # the second number SURVEY 8(d) asks for: end to end -- pinned host buffer -> H2D -> the same step -> D2H of the
    # wire records -> every delivered record handed to a (no-op) callback; 2 steps in flight
    # a run an outside observer can corroborate: the headline configuration for >= 2 s of GPU time in one go (the driver's
    # SMI samples then see the GPU busy; the windows above are 8 ms each)
    sustained = None
    if world == 1 and not args.no_sustained:
        ns = max(200, args.sustained_steps)
        run(2 * D, False)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            t0 = time.perf_counter()
            dl, evs_ = run(ns, False)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        finally:
            gc.enable()
        inner = evs_[D].elapsed_time(evs_[-1]) / (ns - 1 - D)          # steady state: from the D-th completion to the last
        sustained = {"value": sum(dl) / el, "unit": "bursts/s", "steps": ns, "seconds": el, "ms_per_step": el / ns * 1e3,
                     "ms_per_step_between_step_events": inner, "steps_in_flight": D,
                     "note": "%d steps back to back (%d in flight), wall clock around the whole run, ramp-up and drain included" % (ns, D)}

    # the second number SURVEY 8(d) asks for: end to end -- pinned host buffer -> H2D -> the same step -> D2H of the
    # wire records -> every delivered record handed to a (no-op) callback.  Three steps in flight; the copies up, the copies
    # down and the decodes on streams of their own, so that the link carries the next step's bytes while this one decodes
    e2e = None
    if world == 1 and not args.no_e2e:
        try:
            NB = 3
            h_in = torch.from_numpy(buf).pin_memory()
            d_in = [torch.empty_like(d_base) for _ in range(NB)]
            wr = [torch.full((cap * T.WIRE_BYTES,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(NB)]
            h_w = [torch.empty(cap * T.WIRE_BYTES, dtype=torch.uint8).pin_memory() for _ in range(NB)]
            s_up, s_down = torch.cuda.Stream(), torch.cuda.Stream()
            ne = max(6, args.e2e_steps)
            fl, handed = collections.deque(), 0
            up_ms, done_at, per_step = [], [], []

            def e2e_finish(item):
                msd, j, down, u0, u1 = item
                outs_ = msd.collect(raw=True)
                down.synchronize()
                up_ms.append(u0.elapsed_time(u1))
                done_at.append(time.perf_counter())
                return T.wire_foreach_noop(h_w[j].numpy()[:msd.ngrid * T.WIRE_BYTES], None, msd.ngrid), sum(x["nslots"] for x in outs_)

            dec_done = [None] * NB      # the decode that last read d_in[j] / wrote wr[j]
            down_done = [None] * NB     # the copy down that last read wr[j]
            for phase, steps in (("warm", NB), ("timed", ne)):
                if phase == "timed":
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    handed = 0
                    up_ms.clear()
                    done_at.clear()
                    per_step.clear()
                for k in range(steps):
                    j = k % NB
                    if len(fl) == NB:
                        a, b = e2e_finish(fl.popleft())
                        assert a == b, (a, b)
                        handed += a
                        per_step.append(a)
                    u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    if dec_done[j] is not None:
                        s_up.wait_event(dec_done[j])
                    with torch.cuda.stream(s_up):
                        u0.record(s_up)
                        d_in[j].copy_(h_in, non_blocking=True)
                        u1.record(s_up)
                    strm[j].wait_event(u1)
                    if down_done[j] is not None:
                        strm[j].wait_event(down_done[j])
                    plans[j].set_wire(wr[j].data_ptr())
                    msd = T.MultiSyncDev(eng, plans[j], None, d_in[j].data_ptr(), None, recs[j].data_ptr(), 64, strm[j].cuda_stream,
                                         chans=chans)
                    dec_done[j] = torch.cuda.Event()
                    dec_done[j].record(strm[j])
                    s_down.wait_event(dec_done[j])
                    with torch.cuda.stream(s_down):
                        h_w[j].copy_(wr[j], non_blocking=True)
                    down_done[j] = torch.cuda.Event()
                    down_done[j].record(s_down)
                    fl.append((msd, j, down_done[j], u0, u1))
                while fl:
                    a, b = e2e_finish(fl.popleft())
                    assert a == b, (a, b)
                    handed += a
                    per_step.append(a)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            # the pipeline's steady state: from the hand-over of the first timed step's records to that of the last (the first
            # step's 9 ms copy up has nothing to hide under, the last step's decode and copy down nothing behind them)
            steady = sum(per_step[1:]) / (done_at[-1] - done_at[0])
            um = sorted(up_ms)
            med_up = um[len(um) // 2]
            e2e = {"value": steady, "unit": "bursts/s", "steps": ne, "ms_per_step": (done_at[-1] - done_at[0]) / (ne - 1) * 1e3, "steps_in_flight": NB,
                   "with_fill_and_drain": {"value": handed / el, "ms_per_step": el / ne * 1e3},
                   "h2d_bytes_per_step": int(buf.nbytes), "d2h_bytes_per_step": int(cap * T.WIRE_BYTES),
                   "h2d_ms_per_step": {"median": med_up, "min": um[0], "max": um[-1], "gb_per_s_at_median": buf.nbytes / med_up / 1e6,
                                       "slow_fraction (> 1.25 x median)": sum(1 for x in um if x > 1.25 * med_up) / len(um)},
                   "pcie_bound_bursts_per_s": 63e9 / 510.0,
                   "note": "pinned host buffer (1 bit per byte, %.0f MB) -> H2D -> classification / walks / decode -> D2H of the "
                           "40-byte wire records -> every delivered record handed to a no-op C callback (tgpu_wire_foreach); "
                           "%d steps in flight, the copies up and down on streams of their own; bound by the 510 B per burst over "
                           "PCIe (63 GB/s -> 1.2e8 bursts/s), not by the kernels" % (buf.nbytes / 1e6, NB)}
            for p_ in plans:
                p_.set_wire(0)
        except Exception as ex:      # pragma: no cover
            e2e = {"error": repr(ex)}

    # packed ingest (optional path, never `value`; the API's input format stays 1 bit per byte): the same capture in host
    # memory is packed one bit per BIT by host threads (tgpu_pack_bits), 64 MB cross PCIe instead of 510, and the front end
    # starts behind its own bytes -> bits step (tgpu_sync_multi_launch_packed)
    e2ep = None
    if world == 1 and not args.no_e2e:
        try:
            NB = 3
            nth = args.pack_threads or max(1, min(128, host_threads_default()))
            npk = (len(buf) + 7) // 8
            h_pk = [torch.empty(npk, dtype=torch.uint8).pin_memory() for _ in range(NB)]
            d_pk = [torch.empty(npk, dtype=torch.uint8, device="cuda") for _ in range(NB)]
            wr = [torch.full((cap * T.WIRE_BYTES,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(NB)]
            h_w = [torch.empty(cap * T.WIRE_BYTES, dtype=torch.uint8).pin_memory() for _ in range(NB)]
            s_up, s_down = torch.cuda.Stream(), torch.cuda.Stream()
            torch.cuda.synchronize()
            ref_rec = recs[1].clone()        # (plan 1's records as the byte path left them: the packed path must write the same bytes)
            _, bad = T.pack_bits(buf, out=h_pk[0].numpy(), nthreads=nth)
            assert bad == 0
            from concurrent.futures import ThreadPoolExecutor
            packer = ThreadPoolExecutor(max_workers=1)       # the next step's packing runs beside this step's hand-over (ctypes drops the GIL)

            def pack_into(j):
                tp = time.perf_counter()
                T.pack_bits(buf, out=h_pk[j].numpy(), nthreads=nth)
                return (time.perf_counter() - tp) * 1e3
            ne = max(6, 3 * args.e2e_steps)
            fl, done_at, per_step, pack_ms = collections.deque(), [], [], []
            dec_done, down_done, up_done = [None] * NB, [None] * NB, [None] * NB

            def pk_finish(item):
                msd, j, down = item
                outs_ = msd.collect(raw=True)
                down.synchronize()
                done_at.append(time.perf_counter())
                return T.wire_foreach_noop(h_w[j].numpy()[:msd.ngrid * T.WIRE_BYTES], None, msd.ngrid), sum(x["nslots"] for x in outs_)

            for phase, steps in (("warm", NB), ("timed", ne)):
                if phase == "timed":
                    torch.cuda.synchronize()
                    done_at.clear()
                    per_step.clear()
                    pack_ms.clear()
                pending = None
                for k in range(steps):
                    j = k % NB
                    if pending is None:
                        if up_done[j] is not None:
                            up_done[j].synchronize()
                        pending = packer.submit(pack_into, j)
                    if len(fl) == NB:
                        a, b = pk_finish(fl.popleft())
                        assert a == b, (a, b)
                        per_step.append(a)
                    pack_ms.append(pending.result())
                    pending = None
                    if k + 1 < steps:               # the next step's packing starts now: its pinned buffer's last copy up is two steps old
                        jn = (k + 1) % NB
                        if up_done[jn] is not None:
                            up_done[jn].synchronize()
                        pending = packer.submit(pack_into, jn)
                    if dec_done[j] is not None:
                        s_up.wait_event(dec_done[j])
                    with torch.cuda.stream(s_up):
                        d_pk[j].copy_(h_pk[j], non_blocking=True)
                    up_done[j] = torch.cuda.Event()
                    up_done[j].record(s_up)
                    strm[j].wait_event(up_done[j])
                    if down_done[j] is not None:
                        strm[j].wait_event(down_done[j])
                    plans[j].set_wire(wr[j].data_ptr())
                    msd = T.MultiSyncDev(eng, plans[j], None, d_pk[j].data_ptr(), None, recs[j].data_ptr(), 64, strm[j].cuda_stream,
                                         chans=chans, packed=True)     # (a channel's byte offset in the capture is its bit offset in the packed copy)
                    dec_done[j] = torch.cuda.Event()
                    dec_done[j].record(strm[j])
                    s_down.wait_event(dec_done[j])
                    with torch.cuda.stream(s_down):
                        h_w[j].copy_(wr[j], non_blocking=True)
                    down_done[j] = torch.cuda.Event()
                    down_done[j].record(s_down)
                    fl.append((msd, j, down_done[j]))
                while fl:
                    a, b = pk_finish(fl.popleft())
                    assert a == b, (a, b)
                    per_step.append(a)
            torch.cuda.synchronize()
            same = bool(torch.equal(recs[1], ref_rec))
            del ref_rec
            pm = sorted(pack_ms)
            e2ep = {"value": sum(per_step[1:]) / (done_at[-1] - done_at[0]), "unit": "bursts/s", "steps": ne,
                    "ms_per_step": (done_at[-1] - done_at[0]) / (ne - 1) * 1e3, "steps_in_flight": NB,
                    "host_pack": {"threads": nth, "gb_per_s_of_input_at_the_median": len(buf) / (pm[len(pm) // 2] * 1e-3) / 1e9, "ms_per_step_median": pm[len(pm) // 2]},
                    "h2d_bytes_per_step": int(npk), "d2h_bytes_per_step": int(cap * T.WIRE_BYTES),
                    "records_equal_the_byte_path": same,
                    "note": "the capture in pinned host memory (1 bit per byte, %.0f MB) -> tgpu_pack_bits on %d host threads (%.0f MB) -> "
                            "H2D -> tgpu_sync_multi_launch_packed (same kernels behind the front end's own bytes -> bits step) -> D2H of "
                            "the wire records -> callback per delivered record; bound by the host's packing rate" % (buf.nbytes / 1e6, nth, npk / 1e6)}
            packer.shutdown()
            for p_ in plans:
                p_.set_wire(0)
        except Exception as ex:      # pragma: no cover
            e2ep = {"error": repr(ex)}

    # N > 1: the headline is the decode rate of all ranks -- channels shard with no exchange (the reference runs a process per
    # channel), and the task's rule for a path that partitions is "no data-path collective".  The per-step gather of every rank's
    # decoded blocks to rank 0 (north_star's "final decoded-block gather", done here per step or per --gather-every steps) runs in
    # the same job, is checked record by record on the collecting rank, and stands beside it as `gathered` with the link bound that
    # caps it (33 B per burst x the decode rate is more than one xGMI link direction carries).
    # (round 6, ADVICE r5: BASELINE configs[3] is "sharded 8-per-GPU ... RCCL gather of type-1 blocks over xGMI", and its >= 0.9 x
    # per-GPU target refers to THAT -- so at N > 1 the headline is the gathered rate again, as in rounds 3 and 4; `decode_only`, the
    # replica number, stands beside it with the link arithmetic that separates the two.)
    head = gathered if (gathered and world > 1) else decode_only
    value_is = "gathered" if head is gathered else "decode_only"
    out = {"metric": "decoded bursts/s", "value": head["value"], "unit": "bursts/s", "n_gpus": world,
           "steps": K, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
           "config": {"workload": "SB+NDB mix through the burst-sync front end (BASELINE configs[2] composition in configs[3]'s layout): "
                                  "per GPU %d recorded channels of %d slots each (own cell each), frames [SB,N1,N2,N1,N2,N1,N2,N1], cell "
                                  "code from SB1, 1%% damaged training sequences, payload BER %g (bit errors in the coded fields), %d distinct "
                                  "captures resident in HBM (own seed each; step k decodes capture k mod %d, so steps in flight never read "
                                  "the same bytes); step = one pass over all of a GPU's "
                                  "channels as ONE batch (GPU sequence search + demux of every grid slot, the synchroniser walks of "
                                  "all channels on the GPU at 64-byte feeds, device lists, SB1 / fill / masks / trellis); value = "
                                  "delivered bursts/s%s; one host thread per GPU, %d steps in flight" %
                                  (C, per, args.ber, NB, NB,
                                   (" of all ranks with their decoded blocks gathered to rank 0 over RCCL / xGMI, %d steps per exchange (`gathered` = BASELINE "
                                    "configs[3] as written: %d channels sharded %d per GPU + the gather of type-1 blocks); `decode_only` beside it = the same "
                                    "ranks without the gather (independent replicas: the reference's own one-process-per-channel model)" % (G, world * C, C))
                                   if head is gathered else
                                   (" of all ranks (decode_only); every step's decoded blocks gathered to rank 0 in the same job (wire records, RCCL): `gathered`" if gathered else ""), D),
                      "payload_ber": args.ber, "input_buffers": NB, "input_generation_s (host synthesis of the captures, before any timing)": round(t_gen, 2), "input_bytes_resident_per_gpu": int(sum(x.numel() for x in d_bases)),
                      "channels_per_gpu": C, "slots_per_channel": per, "grid_slots_per_step": int(ngrid),
                      "delivered_per_step": int(nd), "host_threads_per_gpu": 1, "steps_in_flight": D,
                      "batches_handed_to_the_host_walk": handed_over,
                      "parallelism": "channels sharded over GPUs (%d per GPU), no collective in decoding" % C +
                                     ("; `gathered`: one gather of wire records per step to rank 0, on a stream of its own" if gathered else ""),
                      "check": check},
           "timing": {"method": "one continuous run of %d warm-up steps + %d x %d steps + tail with %d steps in flight; a HIP event behind each "
                                "step's last operation; window boundary = mean completion time of the %d most recent steps (batches in flight "
                                "complete in bunches: the mean makes a window independent of where in a bunch it is cut), window = (boundary "
                                "K steps later - boundary) / K: exactly K classifications, K walks, K decodes per window; ms_per_step = median "
                                "over the %d windows; max over ranks per window" % (W, R, K, D, D, R),
                      "windows_ms_per_step": head["windows_ms_per_step"], "window_spread": head["window_spread"],
                      "windows_ms_per_step_without_the_averaging": head["unsmoothed_windows_ms_per_step"],
                      "all_windows_ms_per_step": head["all_windows_ms_per_step"],
                      "sync_bracketed_ms_per_step (K steps between two synchronisations, ramp-up and drain included)": head["sync_bracketed_ms_per_step"],
                      "front_end_launches_per_window": K, "one_batch_at_a_time": one_at_a_time},
           "breakdown_ms": {"host cpu per step (process_time over the continuous run: the launching thread + the HIP runtime's own)": head["host_cpu_ms_per_step"],
                            "host cpu per step, the launching thread alone (thread_time: launch call, collect, the polled wait, the interpreter); "
                            "the rest is a thread of the HIP runtime that spins while kernel completions arrive back to back": head["launch_thread_cpu_ms_per_step"],
                            "host wall per step of the continuous run": head["host_wall_ms_per_step"],
                            "gpu kernels per step (serialised, HIP events on the launch stream)": kern_ms},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": float(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": float(achieved) / HBM_PEAK_GBS, "traffic": traffic, "valu_busy_frac": valu_busy,
                        "kernel_ms": kern_ms[dom],
                        "empty_event_bracket_ms": empty_bracket_ms,
                        "kernel_ms_note": "HIP events around the one launch on its stream: the kernel as rocprofv3's trace times it (profiles/r06_mix_rocprofv3.md) "
                                          "+ what a bracket measures around nothing (empty_event_bracket_ms) + the launch's dispatch and the drain of its "
                                          "stores (k_slot_t: 340 MB of records) in front of the closing event",
                        "kernels_within_5_percent_of_the_longest": {k: kern_ms[k] for k in tied},
                        "heavy_kernels": per_kernel,
                        "hbm_bound_kernel": ({"kernel": hbk, "achieved": alg[hbk] / (kern_ms[hbk] * 1e-3) / 1e9,
                                              "frac": alg[hbk] / (kern_ms[hbk] * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kern_ms[hbk],
                                              "note": "the step's kernel that HBM bounds (the stream front end: 510 input bytes per grid slot); the dominant "
                                                      "kernel above is bound by vector-instruction issue, its HBM fraction says little"}
                                             if hbk else None),
                        "pipeline_achieved_gbs_per_gpu": float(decode_only["value"] / world * 820 / 1e9),
                        "pipeline_vector_issue": valu_pipe,
                        "measured_in_this_run": ["achieved", "frac", "kernel_ms", "pipeline_achieved_gbs_per_gpu"],
                        "read_from_profiles/traffic.json (rocprofv3 PMC passes of this command, committed; NOT measured by this run)":
                            ["traffic", "valu_busy_frac", "pipeline_vector_issue.wave_instructions_per_step", "pipeline_vector_issue.sclk_ghz"],
                        "traffic_provenance": traffic_prov,
                        "note": "achieved = the dominant kernel's share of SURVEY 8(d)'s algorithmic bytes (k_front_stream: the 510 "
                                "input bytes of every grid slot; a trellis kernel: type-1 bits at 1 B/bit + 16 B per block of the "
                                "bursts it decodes) / its mean HIP-event duration on its launch stream, measured after the timed "
                                "region; traffic = PMC bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate passes) and "
                                "valu_busy_frac from profiles/traffic.json of the same command.  kernel = the longest kernel of a step "
                                "(round 6: k_slot_t, the trellises of all delivered bursts in one launch -- bound by vector-instruction issue, "
                                "valu_busy_frac; its HBM fraction is small by nature: ~35 instructions per byte); where several are within 5 % "
                                "of the longest, the one of them with the most algorithmic bytes; hbm_bound_kernel = the front end, the kernel "
                                "HBM does bound; heavy_kernels has both fractions for each (DESIGN.md sections 4 and 5)"}}
    if r3form:
        out["round3_form"] = {k_: r3form[k_] for k_ in ("value", "ms_per_step", "windows_ms_per_step", "window_spread", "all_windows_ms_per_step", "host_cpu_ms_per_step")}
        out["round3_form"]["note"] = ("the same measurement as round 3 ran it: 4 batches in flight and the plans' side streams in play (k_vit<432> beside "
                                      "k_vit<216>, the SB1 decode beside the walk): 12 streams on the runtime's 4 hardware queues, batches spread unevenly "
                                      "over them; host_cpu_ms_per_step includes a runtime thread that spins on the cross-stream events")
    if one_input:
        out["one_input_buffer"] = {k_: one_input[k_] for k_ in ("value", "ms_per_step", "windows_ms_per_step", "window_spread")}
        out["one_input_buffer"]["note"] = ("the same run with all %d steps in flight reading ONE capture (what rounds 1-4 reported): the difference to "
                                           "`value` is what a shared input was worth in L2 / Infinity Cache hits" % D)
    if ber2:
        out["ber_%g" % args.ber_secondary] = {k_: ber2[k_] for k_ in ("value", "ms_per_step", "windows_ms_per_step", "window_spread", "bursts_delivered_per_step", "check") if k_ in ber2}
        out["ber_%g" % args.ber_secondary]["note"] = ("the same run on %d other captures with bit error rate %g in the coded fields: the rate is independent "
                                                      "of the payload (a Viterbi and a CRC do the same work whatever the bits)" % (max(1, min(NB, args.ber_buffers)), args.ber_secondary))
    if sustained:
        out["sustained"] = sustained
    if e2e:
        out["end_to_end"] = e2e
    if e2ep:
        out["end_to_end_packed"] = e2ep
    if gathered or gather_error:
        out["value_is"] = value_is
        out["which_figure_answers_which_config"] = {
            "gathered": "BASELINE configs[3] (64 streams sharded 8 per GPU, RCCL gather of type-1 blocks over xGMI) as written: value at N > 1",
            "decode_only": "the same ranks as independent replicas, no data-path collective (BASELINE configs[2] per GPU; the reference's receiver1 / "
                           "receiver2 model): what the decode itself scales like"}
        out["ranks_seen"] = {"world_size": world, "backend": args.backend,
                             "rccl_communicator_ranks": (world if state.get("ccomm") is not None else None)}
        out["decode_only"] = decode_only
        out["gathered"] = gathered if gathered else {"error": gather_error}
    if single:
        out["single_gpu_reference"] = single
        out["per_gpu_efficiency"] = {"decode_only": decode_only["value"] / world / single["value"],
                                     "gathered": (gathered["value"] / world / single["value"]) if gathered else None,
                                     "gathered_link_bound": gathered["link_bound"]["max_per_gpu_efficiency"] if gathered else None,
                                     "scaling_claim": value_is,
                                     "note": "per-GPU rate / the rate of rank 0 running alone in this same job (single_gpu_reference).  `value` and the "
                                             "claim are `gathered` (BASELINE configs[3] as written) whenever the exchange ran; it ships every rank's decoded "
                                             "blocks to rank 0 and is bound by one xGMI link per peer and by rank 0's seven incoming links "
                                             "(gathered_link_bound), not by the decode.  `decode_only` is what the decode scales like: channels shard with "
                                             "no exchange (the reference runs one process per channel, src/receiver1)"}
    return out


def cpu_baseline_config5(phi, types, code, budget_s=8.0):
    """the oracle's config-5 chain on a prefix of the same float stream, one thread: orc_float_to_soft (our soft
    definition next to the reference slicer) + the soft-decision lower MAC (demux, de-interleave, descramble, soft
    Viterbi as libosmocore's accelerated decoder computes it, CRC); and the reference-faithful hard chain
    (float_to_bits.c slicer + hard-decision decode) beside it"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    res = {}
    for soft in (1, 0):
        n = 4000
        for _ in range(2):
            n = min(n, len(types))
            ph = np.ascontiguousarray(phi[:n * 255])
            t0 = time.perf_counter()
            if soft:
                sv = O.float_to_soft(ph)
                ok = O.bench_decode_slots_soft(sv.reshape(n, 510), types[:n], code)[0]
            else:
                bits = O.float_to_bits(ph)
                ok = O.bench_decode_slots(bits.reshape(n, 510), types[:n], code)[0]
            el = time.perf_counter() - t0
            rate = n / el
            nn = int(min(len(types), max(n, rate * budget_s)))
            if nn == n:
                break
            n = nn
        res[soft] = (rate, n, el, int(ok))
    return {"value": res[1][0], "unit": "bursts/s", "cores": 1, "kind": "port",
            "sample": f"the first {res[1][1]} bursts of the same float stream in {res[1][2]:.1f} s: oracle/tetra_oracle.c, "
                      f"orc_float_to_soft + soft-decision decode of every block (one thread, gcc -O3 as prebuilt; "
                      f"{res[1][3]} blocks passed their CRC)",
            "hard_decision_chain": {"value": res[0][0], "unit": "bursts/s",
                                    "sample": f"{res[0][1]} bursts in {res[0][2]:.1f} s: float_to_bits.c slicer restated + the "
                                              f"hard-decision decode (what the reference computes on this input)"},
            "host_cores_available": os.cpu_count()}


def bench_config5(args, T, torch, rank, world, local):
    """BASELINE config 5 (secondary measurement, N=1): float32 phase stream (sigma 0.6 noise) resident in HBM ->
    soft-decision decode of n aligned slots.  value: the fused path (tgpu_plan_execute_float: slicer + soft gather in
    one kernel); two_stage: tgpu_float_to_bits (bit stream + soft stream written, as the float_to_bits program would)
    followed by tgpu_plan_execute_soft."""
    n = args.bursts
    rng = np.random.default_rng(5)
    pat = np.array([3, 0, 1, 0, 1, 0, 1, 0], np.uint8)
    types = np.tile(pat, n // 8 + 1)[:n]
    code = 0x41802A07
    slots = T.synth_slots(types, seed=21, scramb_init=code)
    bits = slots.reshape(-1).astype(np.int64).reshape(-1, 2)
    phi = np.where(bits[:, 0] == 0, 1.0, -1.0) * np.where(bits[:, 1] == 0, 1.0, 3.0)
    phi = (phi + rng.normal(0, 0.6, len(phi))).astype(np.float32)
    eng = T.Engine(local)
    d_phi = torch.from_numpy(phi).cuda()
    d_bits = torch.empty(2 * len(phi) + 64, dtype=torch.uint8, device="cuda")
    d_soft = torch.empty(2 * len(phi) + 64, dtype=torch.int8, device="cuda")
    # (zeroed: a record has bytes no kernel writes -- the SYNC fields of a NORM burst's header, the tail --, and the runs below
    # are compared byte for byte)
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    d_rec2 = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * T.SLOT_BYTES, types, None, np.array([code], np.uint32))
    hs = torch.cuda.current_stream().cuda_stream
    # the timed region: K fused passes, D of them in flight -- every pass on a plan and a stream (= a hardware queue) of its
    # own, so that one pass' slicer (memory-shaped) runs beside another's trellis kernels (issue-shaped), as in the mix
    D = max(1, min(args.depth5, args.steps))
    plans = [plan] + [T.Plan(eng, n, 1) for _ in range(D - 1)]
    recs = [d_rec] + [torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda") for _ in range(D - 1)]
    strm = [torch.cuda.Stream() for _ in range(D)]
    for p_ in plans[1:]:
        p_.load(np.arange(n, dtype=np.uint64) * T.SLOT_BYTES, types, None, np.array([code], np.uint32))
    if D > 1 and not args.side_stream:
        for p_ in plans:
            p_.set_side_stream(False)
    torch.cuda.synchronize()

    def passes(count):
        for k in range(count):
            j = k % D
            plans[j].execute_float(d_phi.data_ptr(), len(phi), recs[j].data_ptr(), strm[j].cuda_stream if D > 1 else hs)
    passes(max(args.warmup, D))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for j in range(1, D):
        assert torch.equal(recs[j], d_rec), "passes in flight gave different records"
    for p_ in plans[1:]:
        p_.close()
    del recs[1:]
    plan.set_side_stream(True)
    # the same passes one at a time (rounds 2-4's form: one plan, its side stream in play)
    for k in range(3):
        plan.execute_float(d_phi.data_ptr(), len(phi), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        plan.execute_float(d_phi.data_ptr(), len(phi), d_rec.data_ptr(), hs)
    torch.cuda.synchronize()
    el_serial = time.perf_counter() - t0
    # the two-stage path, same steps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_f2b = t_dec = 0.0
    for k in range(3 + args.steps):
        if k == 3:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            t_f2b = t_dec = 0.0
        ev[0].record()
        eng.float_to_bits(d_phi.data_ptr(), len(phi), d_bits.data_ptr(), d_soft.data_ptr(), hs)
        ev[1].record()
        plan.execute_soft(d_soft.data_ptr(), d_rec2.data_ptr(), hs)
        ev[2].record()
        torch.cuda.synchronize()
        t_f2b += ev[0].elapsed_time(ev[1]); t_dec += ev[1].elapsed_time(ev[2])
    el2 = time.perf_counter() - t1
    assert torch.equal(d_rec, d_rec2), "fused and two-stage records differ"
    # per-stage durations of the fused path: HIP events between the stages on the launch stream
    ps = min(args.steps, 24)
    prof = T.Prof(ps)
    for k in range(ps):
        plan.execute_float_prof(d_phi.data_ptr(), len(phi), d_rec.data_ptr(), hs, prof, k)
    torch.cuda.synchronize()
    stage_ms = prof.read(ps)[min(2, ps - 1):].mean(axis=0)
    names = T.Prof.stage_names()
    names[0] = "k_front_soft<float>"
    names = [nm.replace("k_vit<", "k_vit_soft<") for nm in names]
    dom = int(np.argmax(stage_ms))
    n1, n2, nsb = int((types == 0).sum()), int((types == 1).sum()), int((types == 3).sum())
    out_b = {0: ALG_BYTES[0] - 510, 1: ALG_BYTES[1] - 510, 3: ALG_BYTES[3] - 510}
    alg_of = {0: n * 1020 + (n1 + n2) * 462 + nsb * 366,            # floats in, type-5 soft values (blocks + BBK) out
              4: n2 * (432 + out_b[1]) + nsb * (216 + 14 + 124 + 32),  # soft values in, type-1 bits + headers out
              5: n1 * (432 + out_b[0])}
    alg = alg_of.get(dom, n * 1020)
    achieved = alg / (stage_ms[dom] * 1e-3) / 1e9
    whole = n * 1020 + n1 * out_b[0] + n2 * out_b[1] + nsb * out_b[3]
    traffic = valu_busy = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tj.get("config5", {}).get(names[dom])
        valu_busy = tj.get("config5_valu_busy", {}).get(names[dom])
    except Exception:
        pass
    p = T.parse_records(d_rec.view(-1, T.REC_BYTES)[:4096].cpu().numpy())
    # the timed output against the oracle's soft chain (first 2048 slots)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    chk = 2048
    sv = O.float_to_soft(phi[:chk * 255])
    ok, want, wcrc = O.bench_decode_slots_soft(sv.reshape(chk, 510), types[:chk], code)
    t = types[:chk]
    a, bq, sb = t == 0, t == 1, t == 3
    assert (p["bbk"][:chk] == want[:, :14]).all() and (p["bits1"][:chk][a] == want[a, 14:282]).all()
    assert (p["bits1"][:chk][bq][:, :124] == want[bq, 14:138]).all() and (p["bits2"][:chk][bq] == want[bq, 138:262]).all()
    assert (p["bits1"][:chk][sb][:, :60] == want[sb, 14:74]).all() and (p["bits2"][:chk][sb] == want[sb, 138:262]).all()
    out = {"metric": "decoded bursts/s", "value": n * args.steps / el, "unit": "bursts/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
           "config": {"workload": "BASELINE config 5: %d bursts (12.5%% SB, 43.75%% NORM_1, 43.75%% NORM_2) as float32 phases "
                                  "(sigma 0.6) resident in HBM -> slicer + soft gather in one kernel -> soft-decision decode "
                                  "(packed 16-bit trellis), records left in HBM" % n,
                      "crc_ok_blocks_first_4096_slots": int(p["crc_ok"].sum()),
                      "checked": "records of the first %d slots == the oracle's soft chain; fused == two-stage on every byte" % chk},
           "passes_in_flight": D,
           "one_pass_at_a_time": {"ms_per_step": el_serial / args.steps * 1e3, "value": n * args.steps / el_serial,
                                  "note": "one plan, one pass behind the other on one stream (k_vit_soft<432> on the plan's side stream "
                                          "beside k_vit_soft<216>): the form of rounds 2-4"},
           "two_stage": {"ms_per_step": el2 / args.steps * 1e3, "value": n * args.steps / el2,
                         "k_float_to_bits_ms (255 floats in, 510 B bits + 510 B soft values out per burst)": t_f2b / args.steps,
                         "execute_soft_ms (k_front_soft + trellis kernels)": t_dec / args.steps},
           "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": float(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": float(achieved) / HBM_PEAK_GBS, "traffic": traffic, "valu_busy_frac": valu_busy,
                        "kernel_ms": float(stage_ms[dom]),
                        "stage_ms": {names[i]: float(stage_ms[i]) for i in range(len(names))},
                        "algorithmic_bytes_per_launch": int(alg),
                        "pipeline_achieved_gbs": float(n * args.steps / el * (whole / n) / 1e9),
                        "note": "stage durations from HIP events between the stages on the launch stream of a pass that runs alone; in the "
                                "timed region passes_in_flight passes run side by side, each on a stream of its own (one pass' slicer beside "
                                "another's trellis kernels); the trellis kernels are bound by vector-instruction issue, not by HBM"}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_config5(phi, types, code)
    print(json.dumps(out))


def bench_conv(args, T, torch, rank, world, local):
    """SURVEY 8(f) item 1 (secondary measurement, N=1): the generic trellis kernel on every (type2, type3, mother
    code, puncturer) row of lower_mac/tetra_conv_enc.c:257-267 -- n blocks of type-3 bits (1 per byte) resident in
    HBM -> type-2 bits.  value = the TCH/4.8 shape (292/432)."""
    n = args.bursts
    shapes = [("BSCH 80/120 r2/3", 80, 120, 4, 0), ("TCH/4.8 292/432", 292, 432, 4, 2), ("TCH/2.4 148/432", 148, 432, 4, 3),
              ("SCH/HD 144/216 r2/3", 144, 216, 4, 0), ("SCH/HU 112/168 r2/3", 112, 168, 4, 0),
              ("SCH/F 288/432 r2/3", 288, 432, 4, 0), ("speech class 1 112/168", 112, 168, 3, 4),
              ("speech class 2 72/162", 72, 162, 3, 5), ("speech class 2 STCH 38/80", 38, 80, 3, 6)]
    polys = {4: ((0, 1, 4), (0, 2, 3, 4), (0, 1, 2, 4), (0, 1, 3, 4)), 3: ((0, 1, 2, 3, 4), (0, 1, 3, 4), (0, 2, 4))}
    eng = T.Engine(local)
    hs = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(8)
    res = {}
    base = 4096
    for name, L, K, mother, pu in shapes:
        t2 = rng.integers(0, 2, (base, L)).astype(np.uint8)
        t2[:, -4:] = 0
        pad = np.concatenate([np.zeros((base, 4), np.uint8), t2], axis=1)
        m = np.zeros((base, L, mother), np.uint8)
        for g, taps in enumerate(polys[mother]):
            for d in taps:
                m[:, :, g] ^= pad[:, 4 - d:4 - d + L]
        t3 = np.stack([T.get_punctured_rate(pu, m[i].reshape(-1), K)[1] for i in range(base)])
        t3 ^= (rng.random(t3.shape) < args.ber).astype(np.uint8)
        if L == 292:
            sample = t3.copy()
        d_in = torch.from_numpy(t3.reshape(-1)).cuda().repeat(-(-n // base))[:n * K].contiguous()
        d_out = torch.zeros(n * L, dtype=torch.uint8, device="cuda")
        cv = T.ConvDecoder(eng, pu, mother, K, L)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for k in range(args.warmup + args.steps):
            if k == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ev[0].record()
            cv.execute(d_in.data_ptr(), n, d_out.data_ptr(), hs)
        ev[1].record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms = ev[0].elapsed_time(ev[1]) / args.steps
        got = d_out.view(n, L)[:base].cpu().numpy()
        res[name] = {"blocks_per_s": n * args.steps / el, "kernel_ms": ms, "trellis_steps_per_s": n * (L + 4) / (ms * 1e-3),
                     "algorithmic_GBps": n * (K + L) / (ms * 1e-3) / 1e9,
                     "blocks_equal_to_tx_first_4096": int((got == t2).all(axis=1).sum())}
        cv.close()
        del d_in, d_out
    head = res["TCH/4.8 292/432"]
    traffic = busy = None
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        if n == 1_000_000:
            traffic = tj.get("k_conv<0, 5, false>")
        busy = tj.get("valu_busy", {}).get("k_conv<0, 5, false>")
    except (OSError, ValueError):
        pass
    out = {"metric": "decoded blocks/s", "value": head["blocks_per_s"], "unit": "blocks/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["kernel_ms"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
           "config": {"workload": "SURVEY 8(f)1: %d type-3 blocks per shape resident in HBM (1 bit per byte, BER %g) -> "
                                  "de-puncture + 16-state Viterbi (k_conv) -> type-2 bits; value = TCH/4.8 (292/432)" % (n, args.ber)},
           "shapes": res,
           "roofline": {"bound": "hbm", "kernel": "k_conv<0, 5, false> (TCH/4.8)", "achieved": head["algorithmic_GBps"],
                        "peak": 8000.0, "unit": "GB/s", "frac": head["algorithmic_GBps"] / 8000.0, "traffic": traffic,
                        "valu_busy_frac": busy, "kernel_ms": head["kernel_ms"],
                        "note": "achieved = (432 type-3 bytes in + 292 type-2 bytes out) x blocks / mean HIP-event duration of the "
                                "launches of the timed region; the kernel is VALU-issue bound like the specialised trellis kernels "
                                "(about 37 vector instructions per trellis step, profiles/r01_conv_rocprofv3.md)"}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import oraclelib as O
        name, L, K, mother, pu = shapes[1]
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < 10.0:
            for i in range(256):
                O.conv_decode_block(pu, mother, sample[(done + i) % base], L, 0)
            done += 256
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / el, "unit": "blocks/s", "cores": 1, "kind": "port",
                               "sample": "%d TCH/4.8 blocks of the same workload in %.1f s, oracle/tetra_oracle.c (depuncture + generic "
                                         "libosmocore Viterbi restatement)" % (done, el)}
    print(json.dumps(out))


def bench_config2(args, T, torch, dist, rank, world, local, steps, warmup, with_cpu=True):
    """BASELINE configs[1]: per GPU n aligned NDB bursts (50 % NORM_1 / 50 % NORM_2), scramb_init 0, no SYNC slot, no
    sync front end; returns the result dict on rank 0"""
    n = args.bursts
    rng = np.random.default_rng(1000 + rank)
    types = np.where(rng.random(n) < 0.5, T.TRAIN_NORM_1, T.TRAIN_NORM_2).astype(np.uint8)
    slots = T.synth_slots(types, seed=1 + rank, scramb_init=0, ber=args.ber)

    eng = T.Engine(local)
    d_stream = torch.from_numpy(slots.reshape(-1)).cuda()
    d_rec = torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda")   # (zeroed: compared byte for byte below)
    plan = T.Plan(eng, n, 1)
    plan.load(np.arange(n, dtype=np.uint64) * T.SLOT_BYTES, types)
    prof = T.Prof(steps)
    stream = torch.cuda.current_stream().cuda_stream

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # D passes in flight, each on a plan and a stream (= a hardware queue) of its own: one pass' gather (memory-shaped) beside
    # another's trellis kernels (issue-shaped), as in the mix
    D = max(1, min(args.depth5, steps))
    plans = [plan] + [T.Plan(eng, n, 1) for _ in range(D - 1)]
    recs = [d_rec] + [torch.zeros(n * T.REC_BYTES, dtype=torch.uint8, device="cuda") for _ in range(D - 1)]
    strm = [torch.cuda.Stream() for _ in range(D)]
    for p_ in plans[1:]:
        p_.load(np.arange(n, dtype=np.uint64) * T.SLOT_BYTES, types)
    if D > 1 and not args.side_stream:
        for p_ in plans:
            p_.set_side_stream(False)

    def passes(count):
        for k in range(count):
            j = k % D
            plans[j].execute(d_stream.data_ptr(), recs[j].data_ptr(), strm[j].cuda_stream if D > 1 else stream)
    passes(max(warmup, D))
    sync_all()
    t0 = time.perf_counter()
    passes(steps)
    sync_all()
    el = time.perf_counter() - t0
    for j in range(1, D):
        assert torch.equal(recs[j], d_rec), "passes in flight gave different records"
    for p_ in plans[1:]:
        p_.close()
    del recs[1:]
    plan.set_side_stream(True)
    # the same passes one at a time (rounds 1-4's form: one plan, k_vit<432> on its side stream beside k_vit<216>)
    for k in range(3):
        plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), stream)
    sync_all()
    t0 = time.perf_counter()
    for k in range(steps):
        plan.execute(d_stream.data_ptr(), d_rec.data_ptr(), stream)
    sync_all()
    el_serial = time.perf_counter() - t0

    # per-stage durations: the same K steps once more with one HIP event between stages on the launch stream
    for k in range(steps):
        plan.execute_prof(d_stream.data_ptr(), d_rec.data_ptr(), stream, prof, k)
    torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # correctness guard on the timed output: every block must have passed its CRC at BER 0
    p = T.parse_records(d_rec.view(n, T.REC_BYTES)[:4096].cpu().numpy())
    if args.ber == 0.0:
        two = types[:4096] == T.TRAIN_NORM_2
        assert (p["crc_ok"][:, 0] == 1).all() and (p["crc_ok"][two, 1] == 1).all(), "decode failed"

    ms = prof.read(steps)  # (steps, stages) milliseconds from HIP events on the launch stream
    stage_ms = ms[min(2, steps - 1):].mean(axis=0)
    names = T.Prof.stage_names()
    n1 = int((types == T.TRAIN_NORM_1).sum())
    n2 = n - n1
    units_bytes = {"k_front": n * 510, "k_vit<432>": n1 * (ALG_BYTES[0] - 510), "k_vit<216>": n2 * (ALG_BYTES[1] - 510)}
    if int(T.get_option(T.OPT_SLOT)) >= 1:      # round 6: both trellis stages are one launch (k_slot_t), timed in the first one's place
        names = ["k_slot_t" if x == "k_vit<216>" else ("empty_event_bracket" if x == "k_vit<432>" else x) for x in names]
        units_bytes["k_slot_t"] = units_bytes["k_vit<432>"] + units_bytes["k_vit<216>"]
    domname, tied2 = pick_roofline_kernel({names[i]: float(stage_ms[i]) for i in range(len(names))}, units_bytes)
    dom = names.index(domname)
    alg = units_bytes.get(names[dom], n1 * ALG_BYTES[0] + n2 * ALG_BYTES[1])
    achieved = alg / (stage_ms[dom] * 1e-3) / 1e9
    value = world * n * steps / el
    traffic = valu_busy = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tj.get(names[dom])
        valu_busy = tj.get("valu_busy", {}).get(names[dom])
    except Exception:
        pass
    out = {
        "metric": "decoded bursts/s", "value": value, "unit": "bursts/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: per GPU %d NDB bursts (50%% NORM_1 SCH/F, 50%% NORM_2 2xNDB, + AACH), "
                               "scramb_init=0, BER %g, aligned 510-B slots resident in HBM, no sync front end, records left in HBM" % (n, args.ber),
                   "bursts_per_gpu": n, "parallelism": "independent channels per GPU, no collective in decoding"},
        "passes_in_flight": D,
        "one_pass_at_a_time": {"ms_per_step": el_serial / steps * 1e3, "value": world * n * steps / el_serial},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": float(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": float(achieved) / HBM_PEAK_GBS, "traffic": traffic, "valu_busy_frac": valu_busy,
                     "kernel_ms": float(stage_ms[dom]), "kernels_within_5_percent_of_the_longest": tied2,
                     "stage_ms": {names[i]: float(stage_ms[i]) for i in range(len(names))},
                     "pipeline_achieved_gbs_per_gpu": float(value / world * ((n1 * ALG_BYTES[0] + n2 * ALG_BYTES[1]) / n) / 1e9)},
    }
    if rank == 0 and world == 1 and with_cpu and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(slots, types)
    plan.close()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="K: steps per timed window")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--windows", type=int, default=6, help="mix: K-step windows inside the one continuous run (median reported)")
    ap.add_argument("--depth", type=int, default=8, help="mix: steps in flight per GPU (plans / streams)")
    ap.add_argument("--trace-dump", default=None, help="mix, with a -DTG_TRACE build of the library: the heavy kernels' workgroup "
                                                       "time stamps of the decode-only run go to this .npy file")
    ap.add_argument("--depth5", type=int, default=4, help="config2 / config5: passes in flight (plans / streams)")
    ap.add_argument("--side-stream", action="store_true", help="mix: keep the plans' side streams in play (the round-3 form)")
    ap.add_argument("--walk-wide", action="store_true", help="mix: the device walk's per-channel launches as 1024 threads / 128 KB of LDS "
                                                             "(rounds 3 and 4) instead of 256 threads and LDS sized per launch")
    ap.add_argument("--no-latency-form", action="store_true", help="mix: skip the one-batch-at-a-time measurement under TGPU_OPT_SLOT 3 (profiling runs: "
                                                                   "its kernels would mix into the per-kernel averages)")
    ap.add_argument("--slot-mode", type=int, default=-1, help="TGPU_OPT_SLOT for this run (A/B): 0 = k_vit<216> + k_vit<432> (rounds 1-5), 1 = k_slot_t, 2 = k_slot "
                                                             "(front end + trellises in one launch) where a channel has a code to decode on; -1 = the library's default")
    ap.add_argument("--streams", type=int, default=0, help="mix: streams the steps in flight run on (0 = one per step in flight; fewer: "
                                                           "plan j runs on stream j %% streams)")
    ap.add_argument("--bursts", type=int, default=1_000_000, help="bursts (slots) per GPU per step")
    ap.add_argument("--ber", type=float, default=0.0, help="bit error rate in the coded fields of the synthetic captures (mix: the headline's captures)")
    ap.add_argument("--input-buffers", type=int, default=0, help="mix: distinct captures resident per GPU, step k decodes capture k %% N (0 = one per step in flight)")
    ap.add_argument("--ber-secondary", type=float, default=0.02, help="mix: payload BER of the secondary run (ber_<x> in the line; 0 = skip)")
    ap.add_argument("--ber-buffers", type=int, default=3, help="mix: distinct captures of the secondary BER run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-2 object of the default run")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (host buffer -> H2D -> step -> D2H -> callback)")
    ap.add_argument("--e2e-steps", type=int, default=32)
    ap.add_argument("--pack-threads", type=int, default=0, help="host threads of the packed-ingest leg (0 = the CPUs this rank may use, at most 128)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the long run (>= 2 s of GPU time in one go)")
    ap.add_argument("--sustained-steps", type=int, default=5000)
    ap.add_argument("--channels", type=int, default=8, help="mix: recorded channels per GPU (BASELINE config 4: 8), all in one batch")
    ap.add_argument("--workload", default="mix", choices=["mix", "config3", "config2", "config5", "conv"],
                    help="mix (default, = config3: the metric's workload): SB+NDB recordings through the GPU burst-sync front end, "
                         "1%% damaged training sequences; config2: aligned NDB slots, no front end; config5: float phases -> "
                         "soft-decision decode; conv: the generic trellis kernel")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the exchange phase with a single rank too (exercises the RCCL path on a 1-GPU box)")
    ap.add_argument("--wire-form", default="compact", choices=["compact", "grid"],
                    help="what a rank hands to the gather: the compact form (delivered bursts only, csrc/tg_cwire.h) or one 40-byte "
                         "wire record per grid slot")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1, compact form: the decoded blocks of this many steps go to rank 0 in ONE exchange (one RCCL group = one "
                         "launch per rank; at most the steps in flight).  1 = an exchange behind every step; 0 (default) = 4 with the compact "
                         "form over RCCL at N > 1 (the group launch and the control message amortised over four steps' 128 MB), else 1")
    ap.add_argument("--torch-gather", action="store_true",
                    help="N > 1: exchange through torch.distributed.gather instead of the library's tgpu_comm_gather")
    ap.add_argument("--gather-timeout", type=int, default=150,
                    help="N > 1: seconds the exchange phase (per-step gather to rank 0) may take before the run reports "
                         "the decode-only number and leaves")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = control-flow check on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher -- N ranks of this very command under
        # torch.distributed.run on a free local port, one per GPU (the reference's multi-channel model is "run N processes",
        # src/receiver1, receiver2).  Rank 0 prints the one JSON line; the launcher passes output and exit code through.
        # (The driver's own form -- torch.distributed.run ... bench.py --gpus N -- sets WORLD_SIZE and lands below.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import osmo_tetra_amd as T

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (world == 1 and args.gpus == 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (launch N ranks for --gpus N, or plain `python bench.py --gpus N`)" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    pinned = numa = None
    if hasattr(os, "sched_setaffinity"):
        # the rank's threads -- and with them its pinned buffers (first touch) -- go to the CPUs of the NUMA node its GPU
        # hangs off (on a two-socket box the far socket costs a third of the host-to-device rate and slows every doorbell);
        # ranks whose GPUs share a node split that node's CPUs among them (one host thread per rank does the launching)
        try:
            allowed = set(os.sched_getaffinity(0))
            locs = [T.device_host_locality(i % torch.cuda.device_count()) for i in range(world)]
            _, node, cpus = locs[local if args.backend != "gloo" else 0]
            mine = sorted(allowed & set(cpus)) or sorted(allowed)
            if world > 1:
                peers = [i for i in range(world) if locs[i if args.backend != "gloo" else 0][1] == node] if cpus else list(range(world))
                share = max(1, len(mine) // len(peers))
                k = peers.index(int(os.environ.get("LOCAL_RANK", "0")))
                mine = mine[k * share:(k + 1) * share] or mine
                pinned = len(mine)
            for tid in os.listdir("/proc/self/task"):       # the runtime's helper threads exist already
                try:
                    os.sched_setaffinity(int(tid), mine)
                except OSError:
                    pass
            numa = {"gpu_numa_node": node, "cpus": len(mine)}
        except Exception as ex:      # pragma: no cover  (no sysfs, an exotic container: the run goes on unpinned)
            numa = {"error": repr(ex)}
    if world > 1 or args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    if args.workload == "config5":
        return bench_config5(args, T, torch, rank, world, local)
    if args.workload == "conv":
        return bench_conv(args, T, torch, rank, world, local)
    if args.workload == "config2":
        out = bench_config2(args, T, torch, dist, rank, world, local, args.steps, args.warmup)
    else:
        out = bench_mix(args, T, torch, dist, rank, world, local)
        if rank == 0:
            out["config"]["host_cores_per_rank"] = pinned if pinned else host_threads_default()
            out["config"]["host_placement"] = numa
        if rank == 0 and not args.no_cpu_baseline:
            # channel 0 of rank 0 as the timed run had it (the generator is deterministic); at N > 1 too (rank 0's host cores, the
            # other ranks idle at the barrier below), so that an N > 1 line carries the CPU figure of the box it ran on
            stream, _, _ = make_mix_stream(T, args.bursts // max(1, args.channels), 0, mnc=42, cc=1, ber=args.ber)
            out["cpu_baseline"] = cpu_baseline_stream(stream)
            if world > 1:
                out["cpu_baseline"]["measured"] = "by rank 0 of this %d-rank job after the timed regions, the other ranks waiting" % world
        if rank == 0 and world == 1:
            if not args.no_secondary:
                c2 = bench_config2(args, T, torch, dist, rank, world, local, 40, 10, with_cpu=False)
                out["config2"] = {k: c2[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "passes_in_flight", "one_pass_at_a_time",
                                                      "config", "roofline")}
    # RCCL prints a version banner through C stdio when its first communicator comes up; on a pipe that text would be
    # flushed at exit, i.e. after the result line.  Push it out on every rank now, then let rank 0 print last.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_gather:
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)


if __name__ == "__main__":
    main()
