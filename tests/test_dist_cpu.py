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
