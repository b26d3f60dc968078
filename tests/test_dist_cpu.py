"""N>1 path on CPU: world_size-2 gloo processes exercise the channel sharding and the wire-record
gather that the GPU ranks perform over RCCL, plus the host-side unpack of gathered records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import osmo_tetra_amd as T
from osmo_tetra_amd import dist as tdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _channel_records(chan, nslots):
    """REAL decoded records of one channel, made on the CPU: synthetic slots of the channel's cell (noisy, so that
    blocks fail their CRC too) decoded by the oracle, laid out as the 320-byte records the trellis kernels write"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oraclelib as O
    mcc, mnc, cc = 262, 42 + chan, 1 + chan % 60
    code = O.scramb_get_init(mcc, mnc, cc)
    types = np.array([3, 0, 1, 0, 1, 0, 1, 0] * (nslots // 8 + 1), np.uint8)[:nslots]
    slots = T.synth_slots(types, seed=300 + chan, scramb_init=code, mcc=mcc, mnc=mnc, cc=cc, ber=0.04)
    ok, want, wcrc = O.bench_decode_slots(slots, types, code, use_acc=1, want_out=True, want_crc=True)
    rec = np.zeros((nslots, T.REC_BYTES), np.uint8)
    rec[:, 0] = types
    two = types != 0
    rec[:, 2] = wcrc[:, 0] == 0x1D0F
    rec[two, 3] = wcrc[two, 1] == 0x1D0F
    rec[:, 4:8] = wcrc.view(np.uint8).reshape(nslots, 4)
    rec[~two, 6:8] = 0
    rec[:, 8:12] = np.full(nslots, code, np.uint32).view(np.uint8).reshape(nslots, 4)
    rec[:, 12:16] = np.arange(nslots, dtype=np.uint32).view(np.uint8).reshape(nslots, 4)
    rec[:, 32:46] = want[:, :14]
    n1, n2, sb = types == 0, types == 1, types == 3
    rec[n1, 48:316] = want[n1, 14:282]
    rec[n2, 48:172] = want[n2, 14:138]
    rec[sb, 48:108] = want[sb, 14:74]
    rec[two, 176:300] = want[two, 138:262]
    return rec, code


def _real_wire(chan, nslots):
    rec, _ = _channel_records(chan, nslots)
    return T.wire_pack(rec)


def _worker(rank, world, port, nchan, nslots, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = tdist.shard_channels(nchan, rank, world)
    # equal-size messages: ranks pad their shard to the largest shard (weak scaling keeps them equal anyway)
    per = max(tdist.shard_channels(nchan, r, world)[1] - tdist.shard_channels(nchan, r, world)[0] for r in range(world))
    local = np.zeros((per * nslots, T.WIRE_BYTES), np.uint8)
    for k, c in enumerate(range(lo, hi)):
        local[k * nslots:(k + 1) * nslots] = _real_wire(c, nslots)
    out, work = tdist.gather_wire(torch.from_numpy(local.reshape(-1)), dst=0, async_op=True)
    work.wait()
    if rank == 0:
        got = []
        for r in range(world):
            rlo, rhi = tdist.shard_channels(nchan, r, world)
            a = out[r].numpy().reshape(-1, T.WIRE_BYTES)
            got.append(a[:(rhi - rlo) * nslots])
        q.put(np.concatenate(got))
    dist.barrier()
    dist.destroy_process_group()


def _delivered(chan, nslots):
    """which slots of a channel count as delivered in the compact-form test (a run of lost bursts per channel)"""
    d = np.ones(nslots, bool)
    d[5 + chan:9 + chan] = False
    d[nslots - 1 - chan % 3] = False
    return d


def _worker_compact(rank, world, port, nchan, nslots, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = tdist.shard_channels(nchan, rank, world)
    # the rank's batch: its channels' grids one after the other (each padded to a multiple of 32 slots), the delivered
    # bitmap, and the compact form of both -- what k_cw_* leave on a GPU rank (tgpu_cwire_pack is their host form)
    pad = (nslots + 31) & ~31
    n = max(hi - lo, 1)
    wire = np.full((n * pad, T.WIRE_BYTES), 0xFF, np.uint8)
    bits = np.zeros(n * pad // 32, np.uint32)
    for k, c in enumerate(range(lo, hi)):
        wire[k * pad:k * pad + nslots] = _real_wire(c, nslots)
        idx = k * pad + np.flatnonzero(_delivered(c, nslots))
        np.bitwise_or.at(bits, idx >> 5, np.uint32(1) << (idx & 31).astype(np.uint32))
    cw = T.cwire_pack(wire, bits, n * pad, [k * pad for k in range(hi - lo)] or [0], [nslots] * (hi - lo) or [0])
    sizes, got = tdist.gather_compact(torch.from_numpy(cw), len(cw), dst=0)
    if rank == 0:
        q.put((sizes, [g.numpy() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_gather_of_compact_buffers_world2():
    """the compact transport form through a world-2 gather: every rank's buffer has its own size (different channel
    counts, different losses); the collecting rank reads the headers, gets exactly the delivered bursts back, and their
    records are the oracle's decode"""
    T.build_library()
    nchan, nslots, world = 5, 40, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_compact, args=(r, world, port, nchan, nslots, q)) for r in range(world)]
    for p in procs:
        p.start()
    sizes, got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(set(sizes)) == 2 and all(len(g) == n for g, n in zip(got, sizes))
    pad = (nslots + 31) & ~31
    nesc = 0
    for r in range(world):
        lo, hi = tdist.shard_channels(nchan, r, world)
        inf = T.cwire_info(got[r])
        assert inf["nchan"] == hi - lo and inf["total_bytes"] == sizes[r]
        wire, bits = T.cwire_expand(got[r])
        assert T.cwire_count(got[r]) == inf["ndelivered"] == sum(int(_delivered(c, nslots).sum()) for c in range(lo, hi))
        for k, c in enumerate(range(lo, hi)):
            rec, code = _channel_records(c, nslots)
            d = _delivered(c, nslots)
            assert inf["chans"][k] == (k * pad, nslots, int(d.sum()))
            w = wire[k * pad:k * pad + nslots]
            assert (w[~d] == 0xFF).all()
            back = T.wire_unpack(w[d], slot_ids=np.flatnonzero(d), codes=np.full(int(d.sum()), code, np.uint32))
            assert (back[:, :16] == rec[d][:, :16]).all() and (back[:, 28:] == rec[d][:, 28:]).all()
            nesc += int((T.parse_records(back)["crc_ok"][:, 0] == 0).sum())
        # less than the grid form, and the good bursts at their short size
        assert sizes[r] < (hi - lo) * nslots * T.WIRE_BYTES
    assert nesc > 0


def test_shard_channels_partition():
    for nchan in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = tdist.shard_channels(nchan, r, world)
                assert 0 <= lo <= hi <= nchan
                cover += list(range(lo, hi))
            assert cover == list(range(nchan))
            sizes = [b - a for a, b in (tdist.shard_channels(nchan, r, world) for r in range(world))]
            assert max(sizes) - min(sizes) <= 1


def test_gloo_gather_of_wire_records_world2():
    T.build_library()
    nchan, nslots, world = 5, 40, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nchan, nslots, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the collecting rank expands the gathered wire records to full records on the host: every channel's blocks are
    # the oracle's decode of that channel (type-1 bits, BBK, CRC words, crc_ok, SYNC-PDU fields of the SYNC bursts)
    nbad = 0
    for c in range(nchan):
        rec, code = _channel_records(c, nslots)
        w = got[c * nslots:(c + 1) * nslots]
        back = T.wire_unpack(w, slot_ids=np.arange(nslots), codes=np.full(nslots, code, np.uint32))
        sb = rec[:, 0] == 3
        assert (back[:, :16] == rec[:, :16]).all() and (back[:, 28:] == rec[:, 28:]).all()
        p = T.parse_records(back)
        assert (p["sbcode"][sb & (p["crc_ok"][:, 0] == 1)] == code).all()
        nbad += int((p["crc_ok"][:, 0] == 0).sum())
    assert nbad > 0
