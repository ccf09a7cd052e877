"""CPU: the N > 1 path (sharding + the one collective) with gloo, world sizes 2, 4 and 8."""
import os
import time
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


def _worker(rank, world, port, n_units, q, spu=2, threaded=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = P.shard_units(n_units, rank, world)
    # stand-in for the coder: stream lengths are a pure function of (unit, stream)
    lens = [1000 + 17 * u + s for u in mine for s in range(spu)]
    per_rank = spu * ((n_units + world - 1) // world)
    g = P.all_gather_stream_lengths(lens, per_rank)
    idx = P.container_index(g, n_units, spu)
    # the overlapped form bench.py uses: one preallocated all_gather_into_tensor, issued asynchronously, waited for later - twice on
    # the same buffers, with other work (here: a second collective and a sleep) between start and wait
    # (threaded: the helper thread of the exchange issues the collective while this thread goes on - here straight into the sleep
    # that stands for the decode calls; no other collective may be issued before wait(), every rank keeps that order)
    ex = P.LengthExchange(per_rank, threaded=threaded)
    for rep in range(3):
        ex.start([ln + rep for ln in lens])
        if not threaded:
            assert ex.work is not None  # in flight: nothing has been waited for
            dist.barrier()
        time.sleep(0.01)
        g2 = ex.wait()
        assert ex.work is None and g2.tolist() == [[v + rep if v >= 0 else v for v in row] for row in g.tolist()], (rank, rep)
        assert ex.total_ms >= 10.0 and ex.exposed_ms < ex.total_ms
    ex.close()
    assert P.container_index(g2, n_units, spu) == [(u, s_, o + 2 * (spu * u + s_), ln + 2) for (u, s_, o, ln) in idx]
    q.put((rank, mine, g.tolist(), idx))
    dist.barrier()
    dist.destroy_process_group()


# (world, units, bitstreams per unit, threaded exchange): BASELINE configs[3] - 24 Kodak images of 2 bitstreams on 2 / 4 / 8 ranks (3 per
# rank at 8) -, configs[4] - ELIC images of 10 bitstreams, one per rank -, and ragged splits whose last ranks own fewer units (or none)
@pytest.mark.parametrize("world,n_units,spu,threaded", [(2, 24, 2, False), (2, 5, 2, True), (4, 24, 2, True), (8, 24, 2, True), (8, 8, 10, False),
                                                          (8, 11, 10, True), (4, 3, 2, False)])
def test_shard_and_gather(world, n_units, spu, threaded):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q, spu, threaded)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort()
    owned = [m for _, m, _, _ in got]
    assert sorted(u for m in owned for u in m) == list(range(n_units))  # every unit has exactly one owner
    assert all(m == list(range(r, n_units, world)) for r, m in enumerate(owned))  # unit i -> rank i mod world
    assert all(g == got[0][2] and idx == got[0][3] for _, _, g, idx in got)  # every rank derives the same container layout
    off = 0
    for k, (u, s, o, ln) in enumerate(got[0][3]):
        assert (u, s) == (k // spu, k % spu) and o == off and ln == 1000 + 17 * u + s
        off += ln


def test_single_process_degenerates():
    g = P.all_gather_stream_lengths([5, 6, 7, 8], 4)
    assert g.tolist() == [[5, 6, 7, 8]]
    assert P.container_index(g, 2, 2) == [(0, 0, 0, 5), (0, 1, 5, 6), (1, 0, 11, 7), (1, 1, 18, 8)]
    assert P.owner_of(9, 8) == 1
    ex = P.LengthExchange(4)  # no process group: the local row
    ex.start([5, 6, 7])
    assert ex.wait().tolist() == [[5, 6, 7, -1]]


# ---- container byte layout (flashgmm_amd/container.py) and its gather across ranks -------------------------------
def _fake_strings(unit: int):
    """what net.compress()["strings"] looks like for a checkerboard GMM model: two GMM streams + the hyper-latent's"""
    rng = torch.Generator().manual_seed(unit)
    def stream(k):
        n = 50 + 13 * unit + k
        data = bytes(torch.randint(0, 256, (n,), generator=rng, dtype=torch.uint8).tolist())
        zb = (torch.rand(19 + unit, generator=rng) > 0.3).to(torch.int64)
        return (data, 7 + unit + k, zb)
    return [stream(0), stream(1), [b"z" * (unit % 3), b""]], {"y": [(4, 8, 12), [2, 3]], "hyper": (2, 3), "none": None}


def test_container_round_trip_and_accounting():
    from flashgmm_amd import container as Cn

    strings, shape = _fake_strings(3)
    blob = Cn.pack(strings, shape)
    s2, shape2 = Cn.unpack(blob)
    assert shape2 == shape and len(s2) == len(strings)
    for a, b in zip(strings, s2):
        if isinstance(a, tuple):
            assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2]) and b[2].dtype == torch.int64
        else:
            assert list(a) == list(b)
    payload = sum(len(s[0]) for s in strings[:2]) + sum(len(b) for b in strings[2])
    assert Cn.num_bytes(strings, shape) == len(blob) and Cn.side_info_bytes(strings, shape) == len(blob) - payload > 0
    assert Cn.unpack(Cn.pack([], None)) == ([], None)
    assert Cn.unpack(Cn.pack([], torch.Size([3, 4])))[1] == (3, 4)
    for bad in (blob[:-1], blob + b"\0", b"XXXX" + blob[4:], blob[:20]):
        with pytest.raises(ValueError):
            Cn.unpack(bad)
    with pytest.raises(TypeError):
        Cn.pack([("not bytes", 1, torch.zeros(2))])


def test_container_carries_checkpointed_streams():
    """a stream's out-of-band checkpoints travel in the container (entry kind 2) and come back as a CheckpointedBytes that
    equals the plain stream; plain streams stay kind 1, byte for byte what they were"""
    import numpy as np
    from flashgmm_amd import CheckpointedBytes, container as Cn
    from flashgmm_amd.entropy_models import CKPT_DTYPE

    ck = np.zeros(3, CKPT_DTYPE)
    ck["x"], ck["pos"] = [1 << 31, (1 << 63) + 5, 12345678901234], [0, 7, (1 << 32) - 1]
    data = bytes(range(200)) * 3
    zb = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int64)
    plain = [(data, 17, zb), [b"zz"]]
    noted = [(CheckpointedBytes(data, ck, 1024), 17, zb), [b"zz"]]
    b0, b1 = Cn.pack(plain, (5, 4, 6)), Cn.pack(noted, (5, 4, 6))
    assert len(b1) == len(b0) + 8 + 12 * 3
    s0, _ = Cn.unpack(b0)
    s1, shape = Cn.unpack(b1)
    assert type(s0[0][0]) is bytes and isinstance(s1[0][0], CheckpointedBytes) and s1[0][0] == data == s0[0][0]
    assert s1[0][0].ckpt_stride == 1024 and np.array_equal(s1[0][0].ckpt, ck) and shape == (5, 4, 6)
    assert Cn.side_info_bytes(noted, (5, 4, 6)) == Cn.side_info_bytes(plain, (5, 4, 6)) + 8 + 36  # counted as side information
    with pytest.raises(ValueError):
        Cn.unpack(b1[:-30] + b1[-10:])


def _gather_worker(rank, world, port, n_units, q):
    from flashgmm_amd import container as Cn

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = [Cn.pack(*_fake_strings(u)) for u in P.shard_units(n_units, rank, world)]
    allc = P.gather_containers(local, n_units)
    q.put((rank, [bytes(b) for b in allc]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 6), (2, 5), (2, 1), (4, 24), (8, 24), (8, 5)])
def test_gather_containers(world, n_units):
    """every rank ends with every unit's container in UNIT order, whatever the world size - 24 units on 4 and 8 ranks (BASELINE
    configs[3]: 3 images per rank at 8), fewer units than ranks (ranks that own nothing contribute an empty payload)"""
    from flashgmm_amd import container as Cn

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [Cn.pack(*_fake_strings(u)) for u in range(n_units)]
    assert all(got[r] == want for r in range(world))  # every rank holds every unit's container, in unit order
    assert P.gather_containers(want[:1], 1) == want[:1]  # world size 1 degenerates to the identity


def test_visible_device_filters_compose_as_the_runtime_composes_them(monkeypatch):
    """ROCR_VISIBLE_DEVICES re-indexes first; HIP_VISIBLE_DEVICES and CUDA_VISIBLE_DEVICES are two names of ONE HIP-level
    filter (launchers export both): applied once, not twice"""
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert P._visible_filter(4) == [0, 1, 2, 3]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "1,0")
    assert P._visible_filter(4) == [1, 0]  # (applied twice it would read [0, 1])
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3,1")
    assert P._visible_filter(4) == [3, 2]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "2")
    assert P._visible_filter(4) == [1]
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "GPU-deadbeef")
    with pytest.raises(LookupError):
        P._visible_filter(4)


def test_numa_binding_is_confirmed_against_the_runtimes_own_address(monkeypatch):
    """Once the runtime is up its PCI address for the device is compared with the one the early sysfs lookup bound to: equal ->
    the description stands; different -> bound again to the runtime's device, and the report says so"""
    calls = []
    monkeypatch.setattr(P, "runtime_pci_address", lambda i: "0000:5d:00.0")
    monkeypatch.setattr(P, "bind_to_gpu_numa_node", lambda i, all_threads=True, bdf=None: calls.append(bdf) or f"gpu {i} ({bdf}) -> NUMA node 1")
    early = "gpu 0 (0000:5d:00.0) -> NUMA node 0, 128 CPUs, 65 thread(s) bound"
    assert P.confirm_numa_binding(0, early) == early and calls == []
    wrong = "gpu 0 (0000:1a:00.0) -> NUMA node 0, 128 CPUs, 65 thread(s) bound"
    out = P.confirm_numa_binding(0, wrong)
    assert calls == ["0000:5d:00.0"] and out.startswith("gpu 0 (0000:5d:00.0)") and "rebound" in out and "0000:1a:00.0" in out
    monkeypatch.setattr(P, "runtime_pci_address", lambda i: (_ for _ in ()).throw(RuntimeError("no device")))
    assert "not confirmed" in P.confirm_numa_binding(0, early)


def test_checkpointed_bytes_is_a_bytes_object_with_its_notes():
    """CheckpointedBytes (flashgmm_amd/entropy_models.py): the reference's stream as ``bytes`` (equal, hashable, sliceable, picklable)
    plus the encoder's notes - given as an array, or (the compiled boundary, csrc/fgmm_pybind.cpp: adopt_ckpts) as a place in the
    blob that holds the notes of all the bitstreams of a call, the view made when somebody asks for it"""
    import pickle

    import numpy as np

    from flashgmm_amd import CheckpointedBytes
    from flashgmm_amd.entropy_models import CKPT_DTYPE

    data = bytes(range(200)) * 3
    ck = np.zeros(3, CKPT_DTYPE)
    ck["x"], ck["pos"] = [1 << 40, 1 << 41, 1 << 42], [5, 9, 14]
    a = CheckpointedBytes(data, ck, 512)
    assert a == data and hash(a) == hash(data) and a[3:9] == data[3:9] and isinstance(a, bytes) and {a: 1}[data] == 1
    assert a.ckpt_stride == 512 and np.array_equal(a.ckpt, ck) and a._ckpt_addr == a.ckpt.ctypes.data and a._ck[2] == 3
    b = pickle.loads(pickle.dumps(a))
    assert type(b) is CheckpointedBytes and b == data and b.ckpt_stride == 512 and np.array_equal(b.ckpt, ck)
    # as the module attaches them: (blob, first record, records, address of the first, stride)
    blob = np.concatenate([np.zeros(2, CKPT_DTYPE), ck, np.zeros(4, CKPT_DTYPE)]).tobytes()
    c = CheckpointedBytes.__new__(CheckpointedBytes, data, np.zeros(0, CKPT_DTYPE), 0)
    c._ck = (blob, 2, 3, 12345, 512)
    assert c.ckpt_stride == 512 and c._ckpt_addr == 12345 and np.array_equal(c.ckpt, ck) and isinstance(c._ck[0], np.ndarray)  # (the view is kept)
    e = CheckpointedBytes(data, np.zeros(0, CKPT_DTYPE), 256)  # a stream too short for a note
    assert len(e.ckpt) == 0 and e._ckpt_addr == 0 and e.ckpt_stride == 256
    e._ck = (blob, 9, 0, 0, 256)
    assert len(e.ckpt) == 0
