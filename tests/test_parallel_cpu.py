"""CPU: the N > 1 path (sharding + the one collective) with gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flashgmm_amd import parallel as P


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_units, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = P.shard_units(n_units, rank, world)
    # stand-in for the coder: stream lengths are a pure function of (unit, stream)
    lens = [1000 + 17 * u + s for u in mine for s in range(2)]
    per_rank = 2 * ((n_units + world - 1) // world)
    g = P.all_gather_stream_lengths(lens, per_rank)
    idx = P.container_index(g, n_units, 2)
    q.put((rank, mine, g.tolist(), idx))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [24, 5])
def test_shard_and_gather_world2(n_units):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort()
    (r0, mine0, g0, idx0), (r1, mine1, g1, idx1) = got
    assert sorted(mine0 + mine1) == list(range(n_units)) and not set(mine0) & set(mine1)
    assert g0 == g1 and idx0 == idx1  # every rank derives the same container layout
    off = 0
    for k, (u, s, o, ln) in enumerate(idx0):
        assert (u, s) == (k // 2, k % 2) and o == off and ln == 1000 + 17 * u + s
        off += ln


def test_single_process_degenerates():
    g = P.all_gather_stream_lengths([5, 6, 7, 8], 4)
    assert g.tolist() == [[5, 6, 7, 8]]
    assert P.container_index(g, 2, 2) == [(0, 0, 0, 5), (0, 1, 5, 6), (1, 0, 11, 7), (1, 1, 18, 8)]
    assert P.owner_of(9, 8) == 1
