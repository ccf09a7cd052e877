"""The product's host pipeline on the FAKE device (tests/fake/fake_device.cpp: device memory = host memory, every stream an in-order
queue on its own thread, the kernels restated from the oracle): randomized batches through the C ABI, every bitstream against the
oracle's encoder, every decode against round(y), truncated bitstreams refused - the same driver scripts/tsan_host.sh runs under
ThreadSanitizer and AddressSanitizer + UBSan.  Here: an un-instrumented build, a few seconds, so that the CPU suite covers the
concurrent pipeline (planner, staging, task queue, event waits, overflow re-runs, GPU-segment hand-back) without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["flashgmm_amd/csrc/fgmm_capi.cpp", "flashgmm_amd/csrc/fgmm_encode.cpp", "flashgmm_amd/csrc/fgmm_decode.cpp", "flashgmm_amd/csrc/fgmm_decode_gpu.cpp",
       "flashgmm_amd/csrc/fgmm_rans.cpp", "tests/fake/fake_device.cpp", "tests/fake/stress_main.cpp"]


@pytest.fixture(scope="module")
def stress_binary():
    out = os.path.join(ROOT, "scripts", "bin", "stress_plain")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    newest = max(os.path.getmtime(os.path.join(ROOT, f)) for f in SRC + ["oracle/fgmm_oracle.c", "flashgmm_amd/csrc/fgmm_ctx.h", "flashgmm_amd/csrc/fgmm_internal.h"])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math"] + SRC +
                              ["-x", "c", "oracle/fgmm_oracle.c", "-x", "none", "-lpthread", "-lm", "-o", out], cwd=ROOT)
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_host_pipeline_on_the_fake_device(stress_binary, seed):
    r = subprocess.run([stress_binary, "4", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "bitstreams == oracle" in r.stdout and " 0 batches" not in r.stdout, r.stdout


def test_host_sources_do_not_include_the_gpu_runtime():
    """the host side reaches the device through fgmm_device.h only: that is what lets it build against the fake"""
    for f in SRC[:5] + ["flashgmm_amd/csrc/fgmm_ctx.h", "flashgmm_amd/csrc/fgmm_internal.h", "flashgmm_amd/csrc/fgmm_device.h"]:
        text = open(os.path.join(ROOT, f)).read()
        assert "hip/hip_runtime" not in text and "#include <hsa" not in text, f


def test_worker_cpus_follow_the_environment(stress_binary):
    """FGMM_WORKER_CPUS: a cpulist = exactly these CPUs for every host worker (a resized pool included), "inherit" = the creating
    thread's mask; unset on a host too small to set an L3 aside (this one: < 32 CPUs would remain) = the creating thread's mask too.
    (What it is for: the workers stream the decode tables through their L3 - profiles/r05_l3_ab.txt.)"""
    cpus = sorted(os.sched_getaffinity(0))
    one = str(cpus[-1])
    for env, want in (({"FGMM_WORKER_CPUS": one}, one), ({"FGMM_WORKER_CPUS": "inherit"}, "")) + ((({}, ""),) if len(cpus) < 48 else ()):
        e = {k: v for k, v in os.environ.items() if k != "FGMM_WORKER_CPUS"}
        e.update(env)
        e["FGMM_STRESS_EXPECT_WORKER_CPUS"] = want
        r = subprocess.run([stress_binary, "1", "3"], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, (env, r.stderr[-2000:])
        assert f"worker CPUs: '{want}'" in r.stdout and "bitstreams == oracle" in r.stdout, r.stdout


@pytest.mark.parametrize("bad", ["abc", "5-", "3-1", "9999999"])
def test_an_unusable_worker_cpulist_fails_the_context(stress_binary, bad):
    """FGMM_WORKER_CPUS that is not a cpulist, or names no CPU the process may run on, is an ERROR at fgmm_ctx_create (through
    fgmm_last_error) - never a silent fall back to the creating thread's mask, which would put the workers' table stream on the very
    L3 the variable was set to protect."""
    e = dict(os.environ, FGMM_WORKER_CPUS=bad)
    r = subprocess.run([stress_binary, "1", "3"], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode != 0 and "FGMM_WORKER_CPUS" in r.stderr, r.stderr[-1000:]


def test_the_reported_worker_cpus_are_the_effective_ones(stress_binary):
    """a list wider than the process's mask: the workers get (and fgmm_ctx_worker_cpus reports) its intersection with the mask"""
    cpus = sorted(os.sched_getaffinity(0))
    e = dict(os.environ, FGMM_WORKER_CPUS=f"{cpus[0]},4000-4003", FGMM_STRESS_EXPECT_WORKER_CPUS=str(cpus[0]))
    r = subprocess.run([stress_binary, "1", "3"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0 and f"worker CPUs: '{cpus[0]}'" in r.stdout, r.stderr[-1000:]


def test_worker_cpulist_is_granted_by_the_kernel_not_by_the_callers_mask(stress_binary):
    """a caller that narrows ITS OWN affinity to the CPUs it keeps for itself before the first context exists (plan_l3's recipe, in
    that order) must still get the workers it named with FGMM_WORKER_CPUS: the process's cpuset decides what a thread may ask for, not
    the creating thread's current mask (round 6: the loud check had intersected with that mask and refused a valid list)"""
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip("one CPU")
    e = dict(os.environ, FGMM_WORKER_CPUS=str(cpus[-1]), FGMM_STRESS_EXPECT_WORKER_CPUS=str(cpus[-1]), FGMM_STRESS_NARROW_CALLER="1")
    r = subprocess.run([stress_binary, "1", "3"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0 and f"worker CPUs: '{cpus[-1]}'" in r.stdout, r.stderr[-1000:]
