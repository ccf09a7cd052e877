"""bench_diag.py — what the HOST did to a timed region of bench.py, and the `--diag` legs.

Not part of the measurement: bench.py's default run takes per-step wall clock and CPU time only and prints a line of
under 4 KB.  With `--diag` it samples the cgroup / pressure / per-thread scheduler files between steps (HostProbe),
reads the library's call log phase by phase (region_phases) and runs the self-diagnosing `step_diag` leg (pool sizes
interleaved, a 0.5 ms watcher of the host's runnable tasks: scripts/host_watch.c).  Everything here goes to
bench_detail.json / stderr, never to the one line the driver parses.  History: profiles/r05_stall_diagnosis.md."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _plain_children:
    """Helper processes (workload generation, the CPU baselines, the host watcher) never touch the GPU - and must not be started with a
    profiler's preloaded tool library: under `rocprofv3 --pmc` every process it is loaded into initialises the GPU, a pool of eight of
    them beside the bench hung a profiling run (and exceeds what a box lets one command put on its card).  The variables are taken out
    of the environment while a pool starts its workers, and put back."""

    NAMES = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "HSA_TOOLS_LIB", "ROCP_TOOL_ATTACH")

    def __enter__(self):
        self.saved = {k: os.environ.pop(k) for k in list(os.environ) if k in self.NAMES or k.startswith("ROCPROF")}
        return self

    def __exit__(self, *exc):
        os.environ.update(self.saved)
        return False

def _cgroup_dirs():
    """directories of this process's cgroup and its ancestors as far as they are visible: [(label, dir)] - cgroup v2 (unified)
    and the v1 cpu controller"""
    rel2 = rel1 = None
    try:
        for ln in open("/proc/self/cgroup"):
            a = ln.rstrip("\n").split(":", 2)
            if len(a) == 3:
                if a[1] == "":
                    rel2 = a[2]
                elif "cpu" in a[1].split(","):
                    rel1 = a[2]
    except OSError:
        pass
    out = []
    for tag, roots, rel in (("v2", ("/sys/fs/cgroup", "/sys/fs/cgroup/unified"), rel2), ("v1", ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"), rel1)):
        if rel is None:
            continue
        for root in roots:
            r, seen = rel.rstrip("/"), 0
            while True:
                d = root + r
                if os.path.exists(os.path.join(d, "cpu.stat")):
                    out.append((f"{tag}:{r or '/'}", d))
                    seen += 1
                if not r:
                    break
                r = r[: r.rfind("/")]
            if seen:
                break
    return out


class HostProbe:
    """What the HOST did to this process during a timed region, step by step: raw reads between steps (a clock, the process's CPU
    time, a few pread()s of files kept open - parsed after the region), so that a slow step can be told apart: the cgroup's CPU
    controller throttled it (cpu.stat of every visible level), its threads waited for a CPU (pressure files, per-thread run delay),
    it spun (CPU time up), or it waited for the device / the bus with its CPUs idle (none of those moved).
    full=False: cpu.stat + cpu.pressure of the cgroup levels (two or three pread()s, ~10 us per step: the headline's region);
    full=True: + memory / io pressure, the host-wide /proc/pressure/*, /proc/stat, /proc/loadavg and every thread's schedstat
    (a few hundred pread()s, ~0.5 ms per step: the diagnostic leg only)."""

    def __init__(self, full: bool = False):
        self.full, self.fds, self.samples = full, [], []
        names = ("cpu.stat", "cpu.pressure") + (("memory.pressure", "io.pressure") if full else ())
        for label, d in _cgroup_dirs():
            for nm in names:
                self._open(f"{label}:{nm}", os.path.join(d, nm))
        if full:
            for nm in ("cpu", "memory", "io"):
                self._open(f"host:pressure.{nm}", f"/proc/pressure/{nm}")
            self._open("host:stat", "/proc/stat")
            self.comm = {}
            self._open("host:loadavg", "/proc/loadavg")
            self._open("host:vmstat", "/proc/vmstat")
            self._open("self:stat", "/proc/self/stat")
            try:
                for tid in os.listdir("/proc/self/task"):
                    self._open(f"task:{tid}", f"/proc/self/task/{tid}/schedstat")
                    try:
                        self.comm[tid] = open(f"/proc/self/task/{tid}/comm").read().strip()
                    except OSError:
                        pass
            except OSError:
                pass

    def _open(self, key, path):
        try:
            fd = os.open(path, os.O_RDONLY)
            os.pread(fd, 64, 0)
            self.fds.append((key, fd))
        except OSError:
            pass

    def sample(self):
        raw = []
        for k, fd in self.fds:
            try:
                raw.append(os.pread(fd, 16384 if k == "host:vmstat" else 4096, 0))
            except OSError:  # (a thread that has exited)
                raw.append(b"")
        self.samples.append((time.perf_counter(), time.process_time(), raw))

    def close(self):
        for k, fd in self.fds:
            try:
                if fd >= 0:
                    os.close(fd)
            except OSError:
                pass
        self.fds = [(k, -1) for k, _ in self.fds]  # (the keys stay: per_step() parses after the region)

    @staticmethod
    def _parse(key, b):
        """-> {counter: number} (monotone counters only, except loadavg's runnable count)"""
        t = b.decode(errors="replace")
        out = {}
        if key.endswith("cpu.stat"):
            for ln in t.splitlines():
                a = ln.split()
                if len(a) == 2 and a[0] in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "nr_bursts", "burst_usec"):
                    out["throttled_usec" if a[0] == "throttled_time" else a[0]] = int(a[1]) // (1000 if a[0] == "throttled_time" else 1)
        elif "pressure" in key:
            for ln in t.splitlines():
                a = ln.split()
                if a and a[0] in ("some", "full"):
                    for f in a[1:]:
                        if f.startswith("total="):
                            out[a[0] + "_us"] = int(f[6:])
        elif key == "host:stat":
            a = t.split("\n", 1)[0].split()
            if a and a[0] == "cpu":
                v = [int(x) for x in a[1:]]
                out["busy_jiffies"] = sum(v[:3]) + sum(v[5:8])  # user nice system + irq softirq steal
                out["idle_jiffies"] = v[3] + v[4]
        elif key == "host:vmstat":  # automatic NUMA balancing at work (host-wide counters): PTEs made inaccessible, hinting faults, pages moved
            for ln in t.splitlines():
                a = ln.split()
                if len(a) == 2 and a[0] in ("numa_pte_updates", "numa_hint_faults", "numa_hint_faults_local", "numa_pages_migrated", "pgmigrate_success",
                                            "pgfault", "thp_split_pmd", "nr_tlb_remote_flush", "nr_tlb_remote_flush_received"):
                    out[a[0]] = int(a[1])
        elif key == "self:stat":  # this process's own page faults (minor, major) and context: are the host's fault bursts ours?
            a = t.rsplit(")", 1)[-1].split()
            if len(a) > 10:
                out["minflt"], out["majflt"] = int(a[7]), int(a[9])
        elif key == "host:loadavg":
            a = t.split()
            if len(a) >= 4 and "/" in a[3]:
                out["runnable_now"] = int(a[3].split("/")[0])
        elif key.startswith("task:"):
            a = t.split()
            if len(a) >= 2:
                out["exec_ns"], out["run_delay_ns"] = int(a[0]), int(a[1])
        return out

    def per_step(self):
        """-> list (one per interval between consecutive samples) of {"ms", "cpu_ms", "<file>.<counter>": delta ...}; thread files
        are summed into "threads.exec_ms" / "threads.run_delay_ms" """
        parsed = [(t, c, [self._parse(k, b) for (k, _), b in zip(self.fds, raw)]) for t, c, raw in self.samples]
        steps = []
        for (t0, c0, p0), (t1, c1, p1) in zip(parsed, parsed[1:]):
            d = {"ms": (t1 - t0) * 1e3, "cpu_ms": (c1 - c0) * 1e3}
            ex = rd = 0
            for (k, _), a, b in zip(self.fds, p0, p1):
                for name in b:
                    if name not in a:
                        continue
                    if k.startswith("task:"):
                        ex, rd = ex + (b[name] - a[name] if name == "exec_ns" else 0), rd + (b[name] - a[name] if name == "run_delay_ns" else 0)
                    elif name == "runnable_now":
                        d[f"{k}.{name}"] = b[name]
                    else:
                        d[f"{k}.{name}"] = b[name] - a[name]
            if self.full:
                d["threads.exec_ms"], d["threads.run_delay_ms"] = ex / 1e6, rd / 1e6
                per = [(k[5:], (b.get("run_delay_ns", 0) - a.get("run_delay_ns", 0)) / 1e6, (b.get("exec_ns", 0) - a.get("exec_ns", 0)) / 1e6)
                       for (k, _), a, b in zip(self.fds, p0, p1) if k.startswith("task:") and "run_delay_ns" in a and "run_delay_ns" in b]
                d["_threads"] = sorted(per, key=lambda x: -x[1])[:6]  # (tid, run delay ms, exec ms): who waited for a CPU
            steps.append(d)
        return steps


def probe_stats(probe: "HostProbe"):
    """what the host did to the process during a region sampled with HostProbe: per cgroup level the CPU controller's throttling over
    the region, CPU usage and pressure-stall time (threads runnable but not running) per step"""
    st = probe.per_step()
    if not st:
        return None
    levels = sorted({k.rsplit(":", 1)[0] for k in st[0] if ":cpu.stat." in k or ":cpu.pressure." in k})
    thr = {}
    for lv in levels:
        key, e = lv + ":cpu.stat", {}
        if key + ".nr_throttled" in st[0]:
            e = {"periods": sum(d.get(key + ".nr_periods", 0) for d in st), "nr_throttled": sum(d.get(key + ".nr_throttled", 0) for d in st),
                 "throttled_ms": round(sum(d.get(key + ".throttled_usec", 0) for d in st) / 1e3, 1)}
        if key + ".usage_usec" in st[0]:
            e["usage_ms"] = [round(d.get(key + ".usage_usec", 0) / 1e3, 1) for d in st]
        psi = lv + ":cpu.pressure.some_us"
        if psi in st[0]:
            e["cpu_pressure_some_ms"] = [round(d.get(psi, 0) / 1e3, 2) for d in st]
        thr[lv] = e
    return {"cpu_throttled": thr, "nr_throttled": sum(v.get("nr_throttled", 0) for v in thr.values())}


class PerfCounters:
    """A perf-stat-style reading of THIS process, every thread it creates after this object included (perf_event_open with inherit:
    open it before the library starts its workers): user-space cycles, instructions, last-level-cache references and misses.
    `read()` -> dict of running totals; differences around a region are the region's.  Unavailable (a kernel that forbids it, a VM
    without a PMU): `ok` is False and read() returns {}."""

    EVENTS = {"cycles": 0, "instructions": 1, "cache_references": 2, "cache_misses": 3}

    def __init__(self):
        import ctypes
        import struct

        self.fds, self.ok = {}, False
        try:
            libc = ctypes.CDLL(None, use_errno=True)
            for name, config in self.EVENTS.items():
                attr = bytearray(128)
                struct.pack_into("IIQQQQ", attr, 0, 0, 128, config, 0, 0, 0)  # PERF_TYPE_HARDWARE, size, config, period, sample_type, read_format
                struct.pack_into("Q", attr, 40, (1 << 1) | (1 << 5) | (1 << 6))  # inherit | exclude_kernel | exclude_hv (enabled at once)
                fd = libc.syscall(298, (ctypes.c_char * 128).from_buffer(attr), 0, -1, -1, 0)
                if fd >= 0:
                    self.fds[name] = fd
            self.ok = len(self.fds) == len(self.EVENTS)
        except Exception:
            pass

    def read(self):
        import struct

        out = {}
        for name, fd in self.fds.items():
            try:
                out[name] = struct.unpack("Q", os.read(fd, 8))[0]
            except OSError:
                pass
        return out

    @staticmethod
    def per_symbol(before, after, symbols, seconds=None):
        if not before or not after or symbols <= 0:
            return None
        d = {k: after[k] - before[k] for k in after if k in before}
        out = {k + "_per_symbol": round(v / symbols, 3) for k, v in d.items()}
        if d.get("cycles"):
            out["ipc"] = round(d.get("instructions", 0) / d["cycles"], 3)
            if seconds:
                out["cpus_busy"] = round(d["cycles"] / seconds / 1e9, 2)  # (GHz-seconds: divide by the clock for CPUs)
        return out


_THROTTLE_FILES = None


def throttle_count():
    """nr_throttled of the cgroup CPU controller, every visible level of this process's cgroup summed (None: no such file) - read
    before and after a timed region: a non-zero difference means `value` measured the quota, not the path"""
    global _THROTTLE_FILES
    if _THROTTLE_FILES is None:
        _THROTTLE_FILES = [os.path.join(d, "cpu.stat") for _, d in _cgroup_dirs()]
    total, seen = 0, False
    for f in _THROTTLE_FILES:
        try:
            for ln in open(f):
                a = ln.split()
                if len(a) == 2 and a[0] == "nr_throttled":
                    total, seen = total + int(a[1]), True
        except OSError:
            pass
    return total if seen else None


def _interpreter_work():
    """a fixed piece of interpreter work (list / dict / ctypes traffic like the wrappers'): ~30 us on an undisturbed Zen 5 core"""
    import ctypes as C_

    a_ = [i * 3 for i in range(600)]
    d_ = {i: str(i) for i in range(300)}
    arr = (C_.c_uint64 * 96)(*range(96))
    s_ = 0
    for i in range(96):
        s_ += arr[i] + len(d_[i]) + a_[i]
    return s_


def settle_calling_thread(max_probe: int = 24):
    """Moves the calling thread to a CPU whose SMT sibling is idle, before a timed region (FGMM_BENCH_SETTLE=0: off).

    Why: the Python between the native calls (`phases_ms.between_calls`) takes 0.46 ms per step in one process and 1.1 ms in the next on
    the same box while every native phase is the same (profiles/r05_stall_diagnosis.md 7).  On these shared hosts a few CPUs at any
    moment run interpreter work at HALF speed - their sibling hyperthread is busy with another tenant (scripts/py_speed_probe.py:
    0.030 ms on most CPUs, 0.048 - 0.058 on some, different ones a minute later) - and the calling thread, which polls and never
    sleeps, is never re-placed by the scheduler once it sits on one.  So: time a fixed piece of interpreter work here and on a sample
    of the allowed CPUs, go to the fastest, and open the affinity mask again (the thread stays where it is until the scheduler has a
    reason).  ~5 ms, outside the timed region; the worker threads are the scheduler's business as before."""
    import ctypes as C_

    if os.environ.get("FGMM_BENCH_SETTLE", "1") == "0":
        return None
    try:
        libc = C_.CDLL(None)
        mask = os.sched_getaffinity(0)

        def speed():
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                _interpreter_work()
                best = min(best, time.perf_counter() - t0)
            return best * 1e3

        speed()
        here = int(libc.sched_getcpu())
        t_here = speed()
        cpus = sorted(mask - {here})
        stride = max(1, len(cpus) // max_probe)
        off = (os.getpid() + here) % stride  # (not the same sample every time)
        best_c, best_t, probed = here, t_here, 0
        try:
            for c in cpus[off::stride][:max_probe]:
                os.sched_setaffinity(0, {c})
                speed()
                t = speed()
                probed += 1
                if t < 0.93 * best_t:
                    best_c, best_t = c, t
            os.sched_setaffinity(0, {best_c})
            t_now = speed()
        finally:
            os.sched_setaffinity(0, mask)
        return {"cpu_before": here, "cpu": best_c, "work_ms_before": round(t_here, 4), "work_ms": round(t_now, 4), "cpus_probed": probed,
                "note": "a fixed piece of interpreter work timed on a sample of the allowed CPUs; the calling thread moved to the fastest (an idle SMT sibling)"}
    except (OSError, AttributeError) as e:  # pragma: no cover
        return {"error": str(e)}


def call_phases(calls):
    """the library's phase marks of one step's native calls (fgmm_ctx_call_log) as named durations: a call's head (until its first
    table copy is queued), bus phase (first copy queued -> last piece seen landed), host tail (-> last coder done), end"""
    ph = {}
    for j, c in enumerate(calls):
        m, nm = c["ms"], f"call{j}_{c['kind']}"
        ph[nm + ".head"] = m[1]
        if c["kind"] == "decode" and c.get("head_ms") and c["head_ms"][2] > 0:  # the head in detail: the calling thread's own work until
            h = c["head_ms"]                                                      # the workers are started | waiting for the first launch's size
            ph[nm + ".head_host"] = h[1]
            ph[nm + ".head_first_launch"] = max(h[2] - h[1], 0.0)
        ph[nm + ".bus"] = max(m[3] - m[1], 0.0)
        ph[nm + ".host_tail"] = max(m[4] - max(m[3], m[1]), 0.0)
        ph[nm + ".end"] = max(m[5] - m[4], 0.0)
        ph[nm + ".worker_busy"] = c["worker_busy_ms"]
        ph[nm + ".worker_wait"] = c["worker_wait_ms"]
    return ph


def region_phases(device: int, calls_per_step: int, step_ms):
    """Where a timed region's steps spent their time, from the library's own call log (a ring of 64 calls the library keeps anyway:
    read AFTER the region, it costs the steps nothing): medians over the region's last steps of every call's phases and of the time
    between the calls (the calling thread's Python).  None when the log does not hold whole steps of the expected shape."""
    from flashgmm_amd import _lib

    m = min(len(step_ms), 64 // max(calls_per_step, 1))
    log = _lib.call_log(device, m * calls_per_step)
    if m < 1 or len(log) != m * calls_per_step:
        return None
    steps = [log[i * calls_per_step:(i + 1) * calls_per_step] for i in range(m)]
    if any(st[0]["kind"] != "encode" or any(c["kind"] == "encode" for c in st[1:]) for st in steps):
        return None
    ph = [call_phases(st) for st in steps]
    for i, st in enumerate(steps):
        ph[i]["between_calls"] = step_ms[len(step_ms) - m + i] - sum(c["ms"][5] for c in st)
        # ... and where: the gap after every call of the step (the interpreter's work until the next native call begins; after the
        # last call: until the next step's first call, not known for the region's last step)
        for j, c in enumerate(st):
            nxt = st[j + 1] if j + 1 < len(st) else (steps[i + 1][0] if i + 1 < len(steps) else None)
            if nxt is not None:
                ph[i][f"gap_after_call{j}"] = nxt["t_begin_ms"] - (c["t_begin_ms"] + c["ms"][5])
    if any(set(p_) - set(ph[0]) for p_ in ph):
        return None
    med = {k: float(np.median([p_[k] for p_ in ph if k in p_])) for k in sorted(ph[0])}
    out = {"steps": m, **{k: round(v, 3) for k, v in med.items()}}
    # a step slower than 1.3x the region's median: which of its phases stretched (more than 0.3 ms over that phase's median)
    last = step_ms[len(step_ms) - m:]
    step_med = float(np.median(last))
    slow = []
    for i, p_ in enumerate(ph):
        if last[i] > 1.3 * step_med:
            moved = {k: [round(p_[k], 2), round(med[k], 2)] for k in med if k in p_ and not k.endswith((".worker_busy", ".worker_wait")) and p_[k] > med[k] + 0.3}
            slow.append({"step": len(step_ms) - m + i, "ms": round(float(last[i]), 2), "moved_[this,median]": moved})
    if slow:
        out["slow_steps"] = slow
    return out


def step_diag(leg, schedule: str, steps: int, configs, block: int = 20):
    """The self-diagnosing leg.  `configs` = ["48", "16", "48+pieces=4" ...]: host workers, optionally
    "+option=value" settings of the library (an A/B inside one run); the configurations take turns in blocks of `block` steps until each has run `steps` steps (boxes differ and drift: only interleaved blocks compare).
    Every step is sampled with the full HostProbe and the library's call log (phase marks of the step's native calls); a helper
    process (scripts/bin/host_watch) samples the host's count of runnable tasks and its own wake-up lateness every 0.5 ms.  For every
    step slower than 1.3x its configuration's median the output says WHICH part stretched - the calling thread's glue, a call's
    head (until its first table copy is queued), its bus phase (first copy queued -> last piece seen landed), its host tail (last
    piece landed -> last coder done) - next to what the host did meanwhile: throttling at every cgroup level, pressure-stall time,
    the run delay of this process's threads (runnable, not running) and which threads, CPU time, the host's runnable tasks."""
    import subprocess

    import torch

    from flashgmm_amd import _lib

    lr = leg.env.local_rank
    before = _lib.lib().fgmm_ctx_threads(_lib.ctx(lr))
    calls_per_step = 1 + (leg.spi if schedule == "codec" else 1)

    phases = call_phases

    rounds = max(1, (steps + block - 1) // block)
    watch = None
    exe = os.path.join(ROOT, "scripts", "bin", "host_watch")
    if os.path.exists(exe):
        try:
            with _plain_children():
                watch = subprocess.Popen([exe, str(60.0)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        except OSError:
            watch = None
    per = {c: {"st": [], "ph": [], "t": [], "comm": {}} for c in configs}
    saved_opts = {}
    wide = None
    if leg.env.l3_cpus:  # as in timed(): the calling thread on its own L3 domain (helper processes have been started by now)
        try:
            wide = os.sched_getaffinity(0)
            os.sched_setaffinity(0, leg.env.l3_cpus)
        except OSError:
            wide = None
    try:
        for _ in range(rounds):
            for cfg in configs:
                pool_, *opts_ = str(cfg).split("+")
                _lib.set_threads(lr, int(pool_))
                for kv in opts_:
                    saved_opts.setdefault(kv.split("=")[0], _lib.get_option(lr, kv.split("=")[0]))
                    _lib.set_option(lr, kv.split("=")[0], int(kv.split("=")[1]))
                for _ in range(2):
                    leg.step(schedule)
                torch.cuda.synchronize()
                probe = HostProbe(full=True)
                logs = []
                probe.sample()
                for _ in range(block):
                    leg.step(schedule)
                    probe.sample()
                    logs.append(_lib.call_log(lr, calls_per_step))
                torch.cuda.synchronize()
                probe.close()
                st = probe.per_step()
                ph = [phases(lg) for lg in logs]
                for i in range(block):
                    ph[i]["python_glue"] = st[i]["ms"] - sum(c["ms"][5] for c in logs[i])
                e = per[cfg]
                e["st"] += st
                e["ph"] += ph
                e["t"] += [(a[0] * 1e3, b[0] * 1e3) for a, b in zip(probe.samples, probe.samples[1:])]  # CLOCK_MONOTONIC ms, as host_watch's
                e["comm"].update(probe.comm)
                for k_, v_ in saved_opts.items():
                    _lib.set_option(lr, k_, v_)
    finally:
        if wide is not None:
            os.sched_setaffinity(0, wide)
        _lib.set_threads(lr, before)
        for k_, v_ in saved_opts.items():
            _lib.set_option(lr, k_, v_)
        wt = wr = wl = None
        if watch is not None:
            watch.terminate()
            try:
                raw = watch.communicate(timeout=10)[0].decode(errors="replace").split()
                k = len(raw) // 3 * 3
                wt, wr, wl = (np.asarray(raw[j:k:3], dtype=np.float64) for j in range(3))
            except Exception:
                wt = None
    out = {"host_threads_tried": list(configs), "block": block, "host_watch": None}
    if wt is not None and len(wt):
        # the host's runnable tasks: how often do they surge (other tenants' threads, all runnable at once), and at which cadence?
        r_med = float(np.median(wr))
        thr = max(3.0 * r_med, r_med + 64.0)
        up = np.flatnonzero((wr > thr) & (np.concatenate(([0.0], wr[:-1])) <= thr))  # rising edges
        at = wt[up] - wt[0]
        gaps = np.diff(at)
        out["host_watch"] = {
            "samples": int(len(wt)), "runnable_median": r_med, "runnable_p99": float(np.percentile(wr, 99)), "runnable_max": float(wr.max()),
            "surge_threshold": thr, "surges": int(len(up)), "surges_per_s": round(len(up) / max((wt[-1] - wt[0]) / 1e3, 1e-9), 1),
            "surge_gap_ms_median": round(float(np.median(gaps)), 1) if len(gaps) else None,
            "surge_gaps_ms_first_20": [round(float(g), 1) for g in gaps[:20]],
            "watcher_late_over_1ms": int((wl > 1.0).sum()), "watcher_late_max_ms": round(float(wl.max()), 2),
            "note": "scripts/bin/host_watch: /proc/loadavg's runnable tasks of the whole host every 0.5 ms + the lateness of its own wake-ups"}
    for cfg in configs:
        e = per[cfg]
        st, ph, n = e["st"], e["ph"], len(e["st"])
        ms = np.asarray([d["ms"] for d in st])
        med = float(np.median(ms))
        slow = [i for i in range(n) if ms[i] > 1.3 * med]
        normal = [i for i in range(n) if ms[i] <= 1.15 * med]
        keys = sorted((set(st[0]) | set(ph[0])) - {"_threads"})
        keys = [k for k in keys if not k.endswith((".nr_periods", ".usage_usec"))]

        def val(i, k):
            return ph[i][k] if k in ph[i] else st[i].get(k, 0)

        def watched(i):
            """the host's runnable tasks and the watcher's lateness while step i ran"""
            if wt is None or not len(wt):
                return None
            m = (wt >= e["t"][i][0]) & (wt <= e["t"][i][1])
            return (float(wr[m].max()), float(wl[m].max())) if m.any() else None

        norm = {k: float(np.median([val(i, k) for i in normal])) for k in keys} if normal else {}
        rows = []
        for i in slow[:10]:
            row = {"step": i, "ms": round(float(ms[i]), 2)}
            moved = {}
            for k in keys:
                v, n0 = val(i, k), norm.get(k, 0.0)
                if k.endswith(("_us", "_usec")):
                    if v - n0 > 300:
                        moved[k] = [round(v / 1e3, 2), round(n0 / 1e3, 2), "ms"]
                elif k.endswith((".nr_throttled", ".nr_bursts", ".majflt")):
                    if v > 0:
                        moved[k] = [v, n0]
                elif k.endswith(("jiffies", "runnable_now")) or k.startswith("host:vmstat"):
                    continue
                elif k.endswith(".minflt"):
                    if v - n0 > 1000:
                        moved[k] = [v, n0]
                elif k != "ms" and v - n0 > 0.3:
                    moved[k] = [round(v, 2), round(n0, 2)]
            row["moved_[slow,normal]"] = moved
            w = watched(i)
            if w:
                row["host_runnable_max"], row["watcher_late_max_ms"] = w[0], round(w[1], 2)
            row["host_pgfault"] = st[i].get("host:vmstat.pgfault")
            row["threads_that_waited_[name,run_delay_ms,exec_ms]"] = [[e["comm"].get(t, t), round(rd, 2), round(ex, 2)] for t, rd, ex in st[i].get("_threads", []) if rd > 0.2]
            rows.append(row)
        wn = [w for w in map(watched, normal) if w]
        thr_total = sum(int(d.get(k, 0)) for d in st for k in d if k.endswith(".nr_throttled"))
        rd_slow = float(np.median([st[i]["threads.run_delay_ms"] for i in slow])) if slow else None
        ws = [w for w in map(watched, slow) if w]
        finding = (f"{len(slow)} of {n} steps slower than 1.3x the median ({med:.2f} ms)" +
                   (f": in those the process's threads stood runnable-but-not-running for {rd_slow:.1f} ms per step "
                    f"({norm.get('threads.run_delay_ms', 0.0):.2f} in a normal step)" if slow else "") +
                   (f" while the HOST's runnable tasks peaked at {np.median([w[0] for w in ws]):.0f} "
                    f"({np.median([w[0] for w in wn]):.0f} in a normal step)" if ws and wn else "") +
                   f"; the cgroup's CPU controller throttled the process {thr_total} times (every visible level summed)")
        out[str(cfg)] = {
            "finding": finding,
            "host_threads": int(str(cfg).split("+")[0]), "options": str(cfg).split("+")[1:], "steps": n, "median_ms": round(med, 3), "p90_ms": round(float(np.percentile(ms, 90)), 3),
            "max_ms": round(float(ms.max()), 3), "mean_ms": round(float(ms.mean()), 3), "p90_over_median": round(float(np.percentile(ms, 90)) / med, 3),
            "slow_steps": len(slow), "all_ms": [round(float(v), 2) for v in ms],
            "run_delay_ms_per_step": {"normal_median": round(norm.get("threads.run_delay_ms", 0.0), 2),
                                      "slow_median": round(float(np.median([st[i]["threads.run_delay_ms"] for i in slow])), 2) if slow else None},
            "host_runnable_max_per_step": {"normal_median": float(np.median([w[0] for w in wn])) if wn else None,
                                           "slow_median": float(np.median([w[0] for w in map(watched, slow) if w])) if slow and wt is not None and len(wt) else None},
            "normal_medians": {k: round(v / 1e3, 3) if k.endswith(("_us", "_usec")) else round(v, 3) for k, v in norm.items()
                               if not k.endswith(("jiffies",)) and not k.startswith("host:vmstat") and (abs(v) > 1e-9 or k.endswith("nr_throttled"))},
            "slow": rows,
        }
    return out


