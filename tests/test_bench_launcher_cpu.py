"""CPU: `python bench.py --gpus N` launches its own ranks (no torch.distributed.run around it) and prints ONE well-formed
JSON line from rank 0.  FGMM_BENCH_DRYRUN=1 replaces the GPU work of a step by a stand-in; the launcher, the process
group (gloo), the per-step all-gather, the barriers and the max-over-ranks timing are the real ones."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = dict(os.environ, FGMM_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_prints_one_line(n):
    r = _run(["--gpus", str(n), "--steps", "4", "--warmup", "1", "--images", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 4 and d["warmup"] == 1 and d["data"] == "dryrun-no-gpu"
    assert len(d["ranks"]["ms_per_step"]) == n and d["ms_per_step"] == max(d["ranks"]["ms_per_step"])
    # what the first hardware run with N > 1 will report per rank: the collective's share, the rank's step time, its share
    # of the host's CPU budget (affinity mask and cgroup quota, divided by the ranks on the node; at least one worker)
    assert d["ranks"]["backend"] == "gloo" and d["ranks"]["rccl_ranks"] == 0 and d["ranks"]["allgather_ms"] > 0
    threads = d["ranks"]["host_threads_per_gpu"]
    budget = d["ranks"]["host_cpu_budget"]
    from helpers import expected_threads
    assert len(threads) == n and all(t == expected_threads(budget, n) for t in threads)
    for key in ("metric", "value", "unit", "higher_is_better", "scaling", "vs_baseline", "dtype", "config"):
        assert key in d


def test_under_an_external_launcher_nothing_is_spawned():
    """With WORLD_SIZE set (torch.distributed.run, the driver's form) a rank runs as itself."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FGMM_BENCH_DRYRUN": "1"}
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), **base)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--warmup", "1", "--images", "2"], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert json.loads(outs[0].strip())["n_gpus"] == 2 and outs[1].strip() == ""


def test_a_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE_MISMATCH": "1", "FGMM_BENCH_DRYRUN": "",
                                                                  "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    # without the dry-run switch the children need a GPU; here there is none, so they fail and so must the launcher
    assert r.returncode != 0


def test_one_rank_dying_at_startup_ends_the_launch_within_seconds():
    """Rank 1 exits before the rendezvous; rank 0 would sit in init_process_group until the store's timeout (minutes).
    The launcher polls every child: it stops rank 0 and fails."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--images", "2"], {"FGMM_BENCH_DRYRUN_FAIL_RANK": "1"})
    dt = time.monotonic() - t0
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""
    assert dt < 60, f"the launcher took {dt:.0f} s to notice a dead rank"


def test_launch_timeout_stops_every_rank():
    """A rank that never finishes (here: rank 0 waiting for a rank that is held back) is stopped by --launch-timeout."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--images", "2", "--launch-timeout", "8"],
             {"FGMM_BENCH_DRYRUN_HANG_RANK": "1"})
    dt = time.monotonic() - t0
    assert r.returncode != 0 and "--launch-timeout" in r.stderr, r.stderr[-2000:]
    assert dt < 60


def test_sigterm_to_the_launcher_stops_the_ranks():
    import signal
    import time

    env = dict(os.environ, FGMM_BENCH_DRYRUN="1", FGMM_BENCH_DRYRUN_HANG_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(6.0)  # both children are up (rank 1 sleeping, rank 0 in the rendezvous)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    p.send_signal(signal.SIGTERM)
    _, err = p.communicate(timeout=60)
    assert p.returncode != 0 and "interrupted by signal" in err, err[-2000:]
    time.sleep(0.5)
    for k in kids:  # no rank survives its launcher
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z"


def test_rehearsal_switch_and_checkpoint_stride_are_parsed_in_the_dry_run():
    """FGMM_BENCH_ONE_DEVICE (the N > 1 rehearsal on a 1-GPU box: tests/test_gpu_parity.py runs it for real) and
    --checkpoint-stride reach the line; a stride the segment decoders do not take is refused at the command line."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--images", "2", "--checkpoint-stride", "512"], {"FGMM_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["config"]["one_device_rehearsal"] is True and d["config"]["checkpoint_stride"] == 512 and d["n_gpus"] == 2
    for bad in ("300", "128", "x"):
        r = _run(["--gpus", "1", "--steps", "1", "--checkpoint-stride", bad])
        assert r.returncode == 2 and "checkpoint-stride" in r.stderr


def test_a_launcher_killed_outright_takes_its_ranks_with_it():
    """SIGKILL leaves the launcher no chance to stop anything: the ranks are told by the kernel (PR_SET_PDEATHSIG)."""
    import signal
    import time

    env = dict(os.environ, FGMM_BENCH_DRYRUN="1", FGMM_BENCH_DRYRUN_HANG_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(6.0)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 2
    p.send_signal(signal.SIGKILL)
    p.wait(timeout=30)
    deadline = time.monotonic() + 30
    while time.monotonic() < deadline and any(os.path.exists(f"/proc/{k}") and open(f"/proc/{k}/stat").read().split()[2] != "Z" for k in kids):
        time.sleep(0.2)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z"


def test_helper_processes_start_without_a_profilers_preload(monkeypatch):
    """bench.py's pools (workload generation, CPU baselines) never touch the GPU and must not inherit a profiler's tool library:
    the variables are out of the environment while a pool starts its workers, and back afterwards"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    monkeypatch.setenv("HSA_TOOLS_LIB", "libx.so")
    monkeypatch.setenv("KEEP_ME", "1")
    with bench._plain_children():
        assert "LD_PRELOAD" not in os.environ and "ROCPROF_OUTPUT_PATH" not in os.environ and "HSA_TOOLS_LIB" not in os.environ
        assert os.environ["KEEP_ME"] == "1"
        seen = subprocess.run([sys.executable, "-c", "import os; print(int('LD_PRELOAD' in os.environ))"], capture_output=True, text=True).stdout.strip()
        assert seen == "0"
    assert os.environ["LD_PRELOAD"].endswith("tool.so") and os.environ["ROCPROF_OUTPUT_PATH"] == "/tmp/x" and os.environ["HSA_TOOLS_LIB"] == "libx.so"


def test_l3_plan_partitions_the_cpus(monkeypatch, tmp_path):
    """parallel.plan_l3 (through bench.plan_l3): every rank of a NUMA node takes one L3 domain (not domain 0) for its calling thread, FGMM_WORKER_CPUS gives the
    workers of every rank the CPUs outside all reserved domains - on a made-up host of 2 NUMA nodes x 4 L3 domains x 16 CPUs, this
    process bound to node 0 (64 CPUs), one rank and four ranks per node; too small a host: nothing is done."""
    import builtins
    import io

    sys.path.insert(0, ROOT)
    import bench
    from flashgmm_amd import parallel as P

    l3 = {c: f"{c // 8 * 8}-{c // 8 * 8 + 7},{128 + c // 8 * 8}-{128 + c // 8 * 8 + 7}" for c in range(32)}  # domain d: CPUs 8d..8d+7 and their siblings 128+
    l3.update({128 + c: l3[c] for c in range(32)})
    mask = set(l3)
    real_open, real_listdir = builtins.open, os.listdir

    def fake_open(path, *a, **k):
        p = str(path)
        if p.startswith("/sys/devices/system/cpu/cpu") and p.endswith("/cache/index3/shared_cpu_list"):
            return io.StringIO(l3[int(p.split("/cpu")[2].split("/")[0])] + "\n")
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(mask))
    monkeypatch.setattr(os, "listdir", lambda p: ["node0", "node1", "possible"] if p == "/sys/devices/system/node" else real_listdir(p))
    monkeypatch.delenv("FGMM_WORKER_CPUS", raising=False)
    monkeypatch.delenv("FGMM_BENCH_L3", raising=False)
    mine, note = bench.plan_l3(0, 1)
    assert mine == set(range(8, 16)) | set(range(136, 144)), note
    assert P.cpulist_to_set(os.environ["FGMM_WORKER_CPUS"]) == mask - mine
    taken = []
    for r in range(8):  # eight ranks on the node, four per NUMA node: ranks 0-3 (and 4-7 on the other node's CPUs) take domains 1-3 ... too few here
        monkeypatch.delenv("FGMM_WORKER_CPUS", raising=False)
        mine, note = bench.plan_l3(r, 8)
        taken.append(mine)
    assert all(t is None for t in taken), "four reserved domains of four would leave the workers nothing: not done"
    for r in range(4):  # four ranks on the node = two per NUMA node: domains 1 and 2 reserved, the workers on domains 0 and 3
        monkeypatch.delenv("FGMM_WORKER_CPUS", raising=False)
        mine, note = bench.plan_l3(r, 4)
        d = 1 + r % 2
        assert mine == set(range(8 * d, 8 * d + 8)) | set(range(128 + 8 * d, 128 + 8 * d + 8)), (r, note)
        assert P.cpulist_to_set(os.environ["FGMM_WORKER_CPUS"]) == set(range(0, 8)) | set(range(24, 32)) | set(range(128, 136)) | set(range(152, 160))
    # GPUs of a NUMA node that are NOT a contiguous block of local ranks (even GPUs on node 0, odd ones on node 1): a rank's domain is
    # its place among the ranks of ITS node - ranks 0 and 2 (both on this node's CPUs) must not collide on one domain
    monkeypatch.setattr(P, "gpu_numa_node", lambda r: r % 2)
    got = {}
    for r in (0, 2):
        monkeypatch.delenv("FGMM_WORKER_CPUS", raising=False)
        got[r], note = bench.plan_l3(r, 4)
    assert got[0] == set(range(8, 16)) | set(range(136, 144)) and got[2] == set(range(16, 24)) | set(range(144, 152)), note
    monkeypatch.setattr(P, "gpu_numa_node", lambda r: None)
    monkeypatch.setenv("FGMM_BENCH_L3", "0")
    monkeypatch.delenv("FGMM_WORKER_CPUS", raising=False)
    assert bench.plan_l3(0, 1)[0] is None and os.environ["FGMM_WORKER_CPUS"] == "inherit"


def test_region_phases_names_what_stretched_in_a_slow_step(monkeypatch):
    """scripts/bench_diag.region_phases: the headline region's time by phase from the library's call log, and for a step slower than 1.3x the median
    the phases that stretched - here a made-up log of three steps whose last one has 5 ms in the head of its first decode call"""
    sys.path.insert(0, ROOT)
    import bench
    from flashgmm_amd import _lib

    def call(kind, ms):
        return {"kind": kind, "count": 24, "t_begin_ms": 0.0, "ms": ms, "worker_busy_ms": 1.0, "worker_wait_ms": 1.0, "head_ms": [0.05, 0.1, 0.13]}

    log = []
    for st in range(3):
        late = 5.0 if st == 2 else 0.0
        log += [call("encode", [0.1, 0.14, 0.2, 0.8, 0.93, 0.94]), call("decode", [0.05, 0.15 + late, 3.0, 3.6 + late, 3.75 + late, 3.8 + late]),
                call("decode", [0.05, 0.15, 3.0, 3.6, 3.75, 3.8])]
    monkeypatch.setattr(_lib, "call_log", lambda dev, last=64: log[-last:])
    ph = bench.D.region_phases(0, 3, [9.1, 9.0, 14.2])
    assert ph["steps"] == 3 and abs(ph["call1_decode.bus"] - 3.45) < 1e-9 and abs(ph["between_calls"] - 0.56) < 0.01
    assert ph["slow_steps"] == [{"step": 2, "ms": 14.2, "moved_[this,median]": {"call1_decode.head": [5.15, 0.15]}}]
    assert bench.D.region_phases(0, 3, [9.1, 9.0]) is not None and bench.D.region_phases(0, 4, [9.1, 9.0, 9.2]) is None  # (a log of another shape: nothing)


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("recorded", ["r05_bench_last.json", "r05_bench_elic4k.json", "r05_n2_one_device.json", "r04_bench_last.json"])
def test_the_stdout_line_fits_the_drivers_window(recorded, tmp_path, capfd, monkeypatch):
    """BENCH_r05.json was `parsed: null`: the line had grown to 22 KB and the driver keeps ~8 KB of stdout.  The line is now a
    summary of the full result (which goes to bench_detail.json and stderr), under 4 KB, asserted in emit(): built here from
    recorded full results of earlier rounds - the largest the bench has ever produced."""
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", recorded)))
    assert len(json.dumps(full)) > 4096  # (the recorded result itself would not fit)
    # the worst case the driver can produce: eight ranks' per-rank entries
    full.setdefault("ranks", {})["ms_per_step"] = [9.446] * 8
    full["ranks"]["host_threads_per_gpu"] = [48] * 8
    detail = tmp_path / "detail.json"
    monkeypatch.setenv("FGMM_BENCH_DETAIL", str(detail))
    bench.emit(full)
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT <= 4096
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d and d[key] == full[key] or key == "config"
    assert d["config"]["workload"] == full["config"]["workload"] and "model" not in d["config"]
    assert set(d["step_ms"]) >= {"min", "median", "p90", "max"}
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    if "cpu_baseline" in full:
        cb = d["cpu_baseline"]
        assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["unit"] == "Mpixels/s" and "sample" in cb and cb["value"] > 0
        assert cb["reference_bytes_equal"] is None or cb["reference_bytes_equal"]["equal"] == cb["reference_bytes_equal"]["streams"]
    # nothing was lost: the whole result is beside the line and on stderr
    assert json.load(open(detail)) == full and d["detail"] == str(detail)
    assert "[bench detail] " in err


def test_a_line_that_would_not_fit_is_refused(monkeypatch, tmp_path):
    bench = _load_bench()
    monkeypatch.setenv("FGMM_BENCH_DETAIL", str(tmp_path / "d.json"))
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_last.json")))
    full["cpu_baseline"]["sample"] = "x" * 5000
    with pytest.raises(AssertionError, match="driver's window"):
        bench.emit(full)
