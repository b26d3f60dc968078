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


def _fake_wire(chan, nslots):
    """deterministic wire records of one channel (content is arbitrary but well-formed)"""
    rng = np.random.default_rng(1000 + chan)
    w = np.zeros((nslots, T.WIRE_BYTES), np.uint8)
    types = rng.choice([T.TRAIN_NORM_1, T.TRAIN_NORM_2, T.TRAIN_SYNC], nslots)
    w[:, 0] = types
    w[:, 2:4] = 1
    w[:, 8:44] = rng.integers(0, 256, (nslots, 36))
    w[:, 44] = rng.integers(0, 256, nslots)
    w[:, 45] = rng.integers(0, 64, nslots)
    return w


def _worker(rank, world, port, nchan, nslots, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = tdist.shard_channels(nchan, rank, world)
    # equal-size messages: ranks pad their shard to the largest shard (weak scaling keeps them equal anyway)
    per = max(tdist.shard_channels(nchan, r, world)[1] - tdist.shard_channels(nchan, r, world)[0] for r in range(world))
    local = np.zeros((per * nslots, T.WIRE_BYTES), np.uint8)
    for k, c in enumerate(range(lo, hi)):
        local[k * nslots:(k + 1) * nslots] = _fake_wire(c, nslots)
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
    want = np.concatenate([_fake_wire(c, nslots) for c in range(nchan)])
    assert (got == want).all()
    # the collecting rank expands wire records to full records on the host
    rec = T.wire_unpack(got[:50], slot_ids=np.arange(50), codes=np.full(50, 0x41802A07, np.uint32))
    p = T.parse_records(rec)
    assert (p["type"] == got[:50, 0]).all() and (p["code"] == 0x41802A07).all()
    n1 = got[:50, 0] == T.TRAIN_NORM_1
    bits = np.unpackbits(got[:50, 8:44], axis=1, bitorder="little")
    assert (p["bits1"][n1] == bits[n1, :268]).all()
    n2 = got[:50, 0] == T.TRAIN_NORM_2
    assert (p["bits1"][n2][:, :124] == bits[n2, :124]).all() and (p["bits2"][n2] == bits[n2, 128:252]).all()
    assert (p["bbk"] == np.unpackbits(got[:50, 44:46], axis=1, bitorder="little")[:, :14]).all()
