"""Multi-GPU sharding of the path: independent units, one exchange.

Images (and, on encode, checkerboard halves) are independent bitstreams, so they are dealt to ranks round-robin and
coded with no data-path collective.  The one exchange is an all-gather of the per-stream byte lengths
(``int64[streams_per_rank]`` per rank — tens of bytes; latency-bound, so xGMI link bandwidth is irrelevant), from
which every rank derives the same stream index / byte offsets of the assembled container.  Backend "nccl" is RCCL
on ROCm; "gloo" runs the same code on CPU tensors (tests/test_parallel_cpu.py, world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_units", "owner_of", "all_gather_stream_lengths", "LengthExchange", "container_index", "gather_containers",
           "bind_to_gpu_numa_node", "confirm_numa_binding", "plan_l3"]


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """unit i -> rank i mod world  (image i -> GPU i mod G)."""
    return list(range(rank, n_units, world))


def owner_of(unit: int, world: int) -> int:
    return unit % world


def all_gather_stream_lengths(local_lengths: Sequence[int], streams_per_rank: int, device=None, group=None) -> torch.Tensor:
    """-> int64 [world, streams_per_rank]; ranks with fewer streams pad with -1."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    buf = torch.full((streams_per_rank,), -1, dtype=torch.int64)
    buf[: len(local_lengths)] = torch.tensor(list(local_lengths), dtype=torch.int64)
    if device is not None:
        buf = buf.to(device)
    if world == 1:
        return buf.unsqueeze(0)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.stack(out)


class LengthExchange:
    """The path's one exchange, made to overlap: ONE ``all_gather_into_tensor`` of the per-stream byte lengths on PREALLOCATED
    buffers, issued asynchronously right after the encode call (``start``) and waited for where the lengths are needed - when the
    containers are assembled, after the decode calls (``wait``).  The lengths only feed ``container_index``; nothing of the
    coding path depends on them, so the collective's latency (tens of microseconds over xGMI, more on a busy host) hides behind
    the decode.  With a process group of ONE rank the same call still goes through the backend (RCCL at N = 1); without a
    process group ``wait`` returns the local row.  Timing of the most recent exchange: ``issue_ms`` (the ``start`` call),
    ``exposed_ms`` (the time ``wait`` blocked), ``total_ms`` (``start`` to the end of ``wait``)."""

    def __init__(self, streams_per_rank: int, device=None, group=None, threaded: bool = False):
        import time

        # threaded: ``start`` hands the exchange to a helper thread of this object and returns at once - the tensor and collective calls
        # (0.1 - 0.2 ms of interpreter and dispatcher work at any world size) then run while the caller is inside its next native call,
        # which releases the GIL; ``wait`` joins the helper first.  One exchange in flight at a time, as without the thread.
        self._thread = None
        self._jobs = None
        self._done = None
        if threaded:
            import queue
            import threading

            self._jobs = queue.SimpleQueue()
            self._done = threading.Event()
            self._done.set()
            self._err = None
        self._time = time.perf_counter
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.n = int(streams_per_rank)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        cuda = self.device.type == "cuda"
        # TWO sets of buffers, used in turn: start() of exchange i + 1 must not rewrite the pinned source of exchange i's
        # non-blocking upload, nor the buffers its collective may still be reading or writing (wait(to_host=False) orders
        # streams, it does not block the host).  Each set remembers the event recorded after its last upload.
        self._slots = [{"host_in": torch.full((self.n,), -1, dtype=torch.int64, pin_memory=cuda),
                        "inp": torch.full((self.n,), -1, dtype=torch.int64, device=self.device),
                        "out": torch.full((self.world, self.n), -1, dtype=torch.int64, device=self.device),
                        "uploaded": None} for _ in range(2)]
        self._cur = 0
        self.work = None
        self.issue_ms = self.exposed_ms = self.total_ms = 0.0
        self._t0 = 0.0
        if threaded:  # (last: the helper sees a complete object)
            import threading

            self._thread = threading.Thread(target=self._serve, name="fgmm-lengths", daemon=True)
            self._thread.start()

    # the current exchange's buffers (what wait() returns a view of)
    @property
    def host_in(self):
        return self._slots[self._cur]["host_in"]

    @property
    def inp(self):
        return self._slots[self._cur]["inp"]

    @property
    def out(self):
        return self._slots[self._cur]["out"]

    def _serve(self):
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            try:
                self._start(job)
            except BaseException as e:  # handed to the caller by wait()
                self._err = e
            self._done.set()

    def close(self) -> None:
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join(timeout=5.0)
            self._thread = None

    def start(self, local_lengths: Sequence[int]) -> None:
        if self._thread is None:
            return self._start(local_lengths)
        self._done.wait()  # (an exchange nobody waited for)
        self._done.clear()
        self._t0 = self._time()
        self._jobs.put(list(local_lengths))

    def _start(self, local_lengths: Sequence[int]) -> None:
        t0 = self._time()
        k = len(local_lengths)
        if k > self.n:
            raise ValueError(f"{k} lengths for {self.n} streams per rank")
        if self.work is not None:  # an exchange nobody waited for: it must have finished before another is issued on the group
            self.work.wait()
            self.work = None
        self._cur ^= 1
        sl = self._slots[self._cur]
        if sl["uploaded"] is not None:  # this set's previous upload (two exchanges ago): long done, but not by construction
            sl["uploaded"].synchronize()
        h = sl["host_in"].numpy()  # (a view: two numpy stores instead of tensor construction + indexing)
        h[k:] = -1
        h[:k] = local_lengths
        sl["inp"].copy_(sl["host_in"], non_blocking=True)
        if self.device.type == "cuda":
            if sl["uploaded"] is None:
                sl["uploaded"] = torch.cuda.Event()
            sl["uploaded"].record(torch.cuda.current_stream(self.device))
        if self.active:
            self.work = dist.all_gather_into_tensor(sl["out"].view(-1), sl["inp"], group=self.group, async_op=True)
        else:
            sl["out"][0].copy_(sl["inp"])
        if self._thread is None:
            self._t0 = t0
        self.issue_ms = (self._time() - t0) * 1e3

    def wait(self, to_host: bool = True) -> torch.Tensor:
        """-> int64 [world, streams_per_rank]; ranks with fewer streams are padded with -1.  ``to_host`` (default): a host copy,
        complete on return; False: the collective's own buffer, ordered after the collective on the current stream (valid until
        the ``start`` after next: the buffers are used in turn) - for a caller that reads the lengths later, or never on the host"""
        t0 = self._time()
        if self._thread is not None:
            self._done.wait()
            if self._err is not None:
                e, self._err = self._err, None
                raise e
        if self.work is not None:
            self.work.wait()
            self.work = None
        if not to_host:
            g = self.out
        else:
            g = self.out.cpu() if self.device.type == "cuda" else self.out.clone()  # (a device tensor: the copy waits for the collective)
        t1 = self._time()
        self.exposed_ms = (t1 - t0) * 1e3
        self.total_ms = (t1 - self._t0) * 1e3
        return g


def container_index(lengths: torch.Tensor, n_units: int, streams_per_unit: int) -> List[Tuple[int, int, int, int]]:
    """From the gathered [world, streams_per_rank] lengths: (unit, stream, byte offset, byte length) in unit order —
    identical on every rank, so any rank can place its payloads without further communication."""
    world = lengths.size(0)
    L = lengths.cpu().tolist()
    out, off = [], 0
    for u in range(n_units):
        r, j = u % world, u // world
        for s in range(streams_per_unit):
            ln = L[r][j * streams_per_unit + s]
            assert ln >= 0, f"unit {u} stream {s}: length missing"
            out.append((u, s, off, ln))
            off += ln
    return out


def gather_containers(local: Sequence[bytes], n_units: int, device=None, group=None) -> List[bytes]:
    """Every rank contributes the packed containers (``flashgmm_amd.container.pack``) of ITS units, in its local unit
    order; every rank gets all ``n_units`` containers back in unit order (unit i lives on rank i mod world).

    Two collectives: the lengths (``all_gather_stream_lengths``), then one all-gather of the payloads padded to the
    longest rank — a Kodak image is ~100 KB, so this too is latency-bound.  ``device``: where the collective buffers
    live ("nccl"/RCCL needs the rank's GPU, gloo the CPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_units(n_units, rank, world)
    if len(local) != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} units, got {len(local)} containers")
    per_rank = (n_units + world - 1) // world
    lengths = all_gather_stream_lengths([len(b) for b in local], per_rank, device=device, group=group).cpu()
    totals = lengths.clamp(min=0).sum(1)
    pad = int(totals.max())
    buf = torch.zeros(max(pad, 1), dtype=torch.uint8)
    blob = b"".join(local)
    if blob:
        buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    if device is not None:
        buf = buf.to(device)
    if world == 1:
        gathered = [buf]
    else:
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
    out: List[bytes] = []
    L = lengths.tolist()
    offs = [0] * world
    for u in range(n_units):
        r, j = u % world, u // world
        ln = L[r][j]
        out.append(gathered[r][offs[r]: offs[r] + ln].cpu().numpy().tobytes())
        offs[r] += ln
    return out


def _visible_filter(n_gpus: int) -> List[int]:
    """KFD GPU ordinals HIP enumerates, in HIP's order, when ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES hold plain indices"""
    import os

    idx = list(range(n_gpus))
    # ROCR_VISIBLE_DEVICES filters at the ROCr level and COMPOSES with the HIP-level filter; HIP_VISIBLE_DEVICES and
    # CUDA_VISIBLE_DEVICES are two names of that ONE HIP-level filter (launchers often export both): the first one set is
    # applied, once
    hip_level = next((nm for nm in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if (os.environ.get(nm) or "").strip() != ""), None)
    for name in ("ROCR_VISIBLE_DEVICES", hip_level):
        v = os.environ.get(name) if name else None
        if v is None or v.strip() == "":
            continue
        try:
            pick = [int(t) for t in v.split(",") if t.strip() != ""]
        except ValueError:  # UUIDs: not resolvable from sysfs alone
            raise LookupError(f"{name}={v!r}")
        idx = [idx[i] for i in pick if 0 <= i < len(idx)]
    return idx


def cpulist_to_set(text: str) -> set:
    """Linux cpulist ("0-7,16-23") -> set of CPU numbers"""
    out = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.update(range(int(lo), int(hi or lo) + 1))
    return out


def set_to_cpulist(cpus) -> str:
    c = sorted(cpus)
    runs, i = [], 0
    while i < len(c):
        j = i
        while j + 1 < len(c) and c[j + 1] == c[j] + 1:
            j += 1
        runs.append(f"{c[i]}-{c[j]}" if j > i else f"{c[i]}")
        i = j + 1
    return ",".join(runs)


def plan_l3(local_rank: int, ranks_on_node: int):
    """Which CPUs this rank's CALLING thread keeps to itself and which its host workers get - to be decided BEFORE the library creates
    the workers (the first ``_lib.ctx`` of the process), and the same way on every rank of the node.

    The workers stream the decode-side tables (358 MB per Kodak step) through the L3 of whatever core complex they run on, and an
    interpreter that shares that L3 runs the code between the native calls from DRAM: 0.9 - 1.5 ms of Python per Kodak step instead
    of 0.5, per process, by the luck of the scheduler's placement (profiles/r05_l3_ab.txt: 969 / 979 / 989 / 977 / 923 Mpixels/s
    against 1 024 / 1 034 / 1 031 / 1 029 / 1 021, taking turns on one box).  The library by itself keeps its workers off the L3 its
    creating thread sits on (``fgmm_ctx_worker_cpus``); with several ranks on a node that is not enough - rank A's workers would stream
    through rank B's reserved L3 - so here the L3 domains of the process's (NUMA-bound) CPUs are numbered, domain 0 is left alone
    (housekeeping lands there), each rank of the NUMA node takes one of the next ones for its calling thread, and the environment
    variable FGMM_WORKER_CPUS gives the workers of EVERY rank the CPUs outside all of those.  The caller then keeps its thread on the
    returned CPUs while it drives the codec (``os.sched_setaffinity(0, cpus)``: the calling thread only - restore the wider mask
    before starting helper processes, they inherit it).
    -> (the calling thread's CPUs or None, description).  Nothing is done (the library's own rule applies) when FGMM_WORKER_CPUS is
    already set, when the topology is unreadable or when the workers would be left fewer than 32 CPUs."""
    import os

    if os.environ.get("FGMM_WORKER_CPUS"):
        return None, "FGMM_WORKER_CPUS is set by the caller: left alone"
    try:
        mask = os.sched_getaffinity(0)
        domains, seen = [], set()
        for c in sorted(mask):
            if c in seen:
                continue
            d = cpulist_to_set(open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read()) & mask
            seen |= d | {c}
            domains.append(d)
        # the ranks that share this rank's (NUMA-bound) CPUs, and this rank's place among them: from the GPUs' own NUMA nodes where
        # sysfs gives them (the GPUs of a node need not be a contiguous block of local ranks), else assuming the GPUs are spread evenly
        # over the nodes in blocks
        nodes_of = [gpu_numa_node(r) for r in range(max(ranks_on_node, 1))]
        if 0 <= local_rank < len(nodes_of) and all(n is not None for n in nodes_of):
            peers = [r for r, n in enumerate(nodes_of) if n == nodes_of[local_rank]]
            per_node, slot = len(peers), peers.index(local_rank)
        else:
            numa_nodes = max(1, len([n for n in os.listdir("/sys/devices/system/node") if n.startswith("node") and n[4:].isdigit()]))
            per_node = -(-max(ranks_on_node, 1) // numa_nodes)
            slot = local_rank % per_node
        reserved = domains[1:1 + per_node]
        workers = mask - set().union(*reserved) if reserved else mask
        if len(reserved) < per_node or len(workers) < 32:
            return None, f"not done: {len(domains)} L3 domains in {len(mask)} CPUs, {per_node} rank(s) on the NUMA node (the library's own rule applies)"
        mine = reserved[slot]
        os.environ["FGMM_WORKER_CPUS"] = set_to_cpulist(workers)
        return mine, (f"calling thread on CPUs {set_to_cpulist(mine)} (L3 domain {1 + slot} of {len(domains)}); host workers on the "
                      f"{len(workers)} CPUs outside the {per_node} reserved domain(s)")
    except (OSError, ValueError) as e:
        return None, f"not done ({e}): the library's own rule applies"


def gpu_pci_address(device_index: int) -> str:
    """PCI address of HIP device `device_index` WITHOUT initialising HIP: the KFD topology lists the GPUs in the order the
    runtime enumerates them (nodes with SIMDs; ``domain`` and ``location_id`` = bus << 8 | devfn).  Raises LookupError when
    the topology is unreadable (the caller then asks the runtime)."""
    import os

    base = "/sys/class/kfd/kfd/topology/nodes"
    gpus = []
    try:
        for n in sorted(os.listdir(base), key=int):
            props = dict(line.split()[:2] for line in open(f"{base}/{n}/properties") if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
                gpus.append(f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}")
    except (OSError, ValueError, KeyError) as e:
        raise LookupError(f"KFD topology: {e}")
    order = _visible_filter(len(gpus))
    if not 0 <= device_index < len(order):
        raise LookupError(f"device {device_index} of {len(order)} visible GPUs")
    return gpus[order[device_index]]


def gpu_numa_node(device_index: int) -> Optional[int]:
    """NUMA node of HIP device `device_index` from sysfs alone (no GPU call); None when the topology does not say"""
    try:
        node = int(open(f"/sys/bus/pci/devices/{gpu_pci_address(device_index)}/numa_node").read())
        return node if node >= 0 else None
    except (LookupError, OSError, ValueError):
        return None


def runtime_pci_address(device_index: int) -> str:
    """PCI address of HIP device `device_index` as the RUNTIME reports it (initialises the device)."""
    p = torch.cuda.get_device_properties(device_index)
    return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def confirm_numa_binding(device_index: int, description: str) -> str:
    """After the runtime is up: is the GPU that ``bind_to_gpu_numa_node`` looked up in sysfs the one the runtime calls
    `device_index`?  (UUID filters, a runtime that orders devices differently: the early lookup cannot know.)  If not, bind
    again to the runtime's device and say so; returns the description to report."""
    try:
        seen = runtime_pci_address(device_index)
        if f"({seen})" in description:
            return description
        return bind_to_gpu_numa_node(device_index, bdf=seen) + f"  [rebound: the early lookup said: {description}]"
    except Exception as e:  # pragma: no cover
        return description + f"  [not confirmed: {e}]"


def bind_to_gpu_numa_node(device_index: int, all_threads: bool = True, bdf: Optional[str] = None) -> str:
    """Pin this process to the CPUs of the NUMA node the GPU hangs off: every thread it has NOW (``all_threads``: the HIP
    runtime's signal / interrupt handling threads exist from the first GPU call on, and ``sched_setaffinity(0, ...)`` alone
    moves the calling thread only) and, by inheritance, every thread created afterwards (the host rANS workers).

    The decode-side tables cross PCIe at ~57 GB/s per GPU into pinned host memory (which the runtime places on the GPU's
    node by itself) and are then read by the host rANS workers; with 8 GPUs on a 2-socket host that traffic — and the
    wake-ups that follow every copy — should stay on the GPU's own socket.  Best effort: returns a short description,
    never raises.  Call it as early as possible (the PCI address comes from sysfs: no GPU call is needed first)."""
    import os

    try:
        if bdf is None:
            try:
                bdf = gpu_pci_address(device_index)
            except LookupError:
                bdf = runtime_pci_address(device_index)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return f"gpu {device_index} ({bdf}): no NUMA affinity reported"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"gpu {device_index} ({bdf}): node {node} has no allowed CPUs"
        os.sched_setaffinity(0, cpus)
        moved = 1
        if all_threads:
            moved = 0
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), cpus)
                    moved += 1
                except OSError:  # the thread has gone, or may not be moved
                    pass
        return f"gpu {device_index} ({bdf}) -> NUMA node {node}, {len(cpus)} CPUs, {moved} thread(s) bound"
    except Exception as e:  # pragma: no cover - topology files differ between hosts
        return f"gpu {device_index}: not bound ({e})"

